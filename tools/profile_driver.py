"""Small driver for rocprofv3 runs: renders one config for a few passes and prints Mrays/s.
usage: python3 tools/profile_driver.py <config.yml> <spp> <repeats> [paths_per_pass]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracing_amd as rt  # noqa: E402

if os.environ.get("RT_VARIANT_LIB"):   # A/B runs: a library built with other -D flags (tools/build_variant.sh)
    rt.LIB_PATH = os.path.join(ROOT, os.environ["RT_VARIANT_LIB"])
name, spp, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ppp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
abl = int(sys.argv[5]) if len(sys.argv) > 5 else 0
b = rt.Builder()
if name.startswith("Synthetic:"):      # the sphere-field stress scene at 4K (no config file: it is not a scene of the reference)
    import types
    cfg = types.SimpleNamespace(image_width=3840, image_height=2160, max_depth=50)
    scene = rt.Scene(b, b.canned_scene(name, 0x5EED0001), 0)
    cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
    background = (0.7, 0.8, 1.0)
else:
    path = os.path.join(ROOT, "configs", name)
    cfg = rt.load_config(path)
    scene = rt.Scene(b, b.scene_from_config(path, 0x5EED0001), 0)
    cam = rt.camera_from_config(cfg)
    background = (0.0, 0.0, 0.0)
import torch  # noqa: E402  (device buffers only)
acc = torch.zeros((cfg.image_height, cfg.image_width, 3), dtype=torch.float32, device="cuda:0")
for i in range(reps):
    p = rt.render_params(cfg.image_width, cfg.image_height, spp, cfg.max_depth, 0x5EED0002, sample_offset=i * spp, paths_per_pass=ppp, flags=abl, background=background)
    st = scene.render_device(cam, p, acc.data_ptr(), None, None)
    print("flags %d %s spp %d: %.2f ms, %.1f Mrays/s, bounce loop %.2f ms (trace %.2f ms), passes %d" % (abl, name, spp, st.seconds * 1e3, st.rays / st.seconds / 1e6, st.kernel_ms_extend, st.kernel_ms_trace, st.passes), flush=True)
if abl & 2:
    print("work per ray:", " ".join("%s=%.2f" % (n, st.work[i] / st.rays) for i, n in ((0, "nodes"), (15, "wide_nodes"), (1, "spheres"), (7, "boxes"), (3, "rects"), (4, "instances"))), flush=True)
print("library:", [l.split()[-1] for l in open("/proc/self/maps") if "ray_tracing_amd" in l and l.rstrip().endswith(".so")][:1], flush=True)
