"""A/B of environment switches that the library reads at scene creation: renders a bench workload under each setting in turn
(same process, same GPU) and prints step time and Mrays/s.
usage: ab_env.py <final_scene.yml|cornell.yml|synthetic1m|...> <spp per step> VAR=a,b,c [VAR2=x,y ...] [--steps N] [--check]
--check: the images of all settings must be bit-identical to the first one's."""
import itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_amd as rt

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 4
if "--steps" in sys.argv:
    args.remove(sys.argv[sys.argv.index("--steps") + 1])
check = "--check" in sys.argv
name, spp = args[0], int(args[1])
sweeps = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args[2:]]
b = rt.Builder()
if name == "synthetic1m":
    w = b.canned_scene("Synthetic:1000000", 0x5EED0001)
    cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
    W, H, D, bg = 3840, 2160, 50, (0.7, 0.8, 1.0)
else:
    path = os.path.join(ROOT, "configs", name)
    cfg = rt.load_config(path); w = b.scene_from_config(path, 0x5EED0001); cam = rt.camera_from_config(cfg)
    W, H, D, bg = cfg.image_width, cfg.image_height, cfg.max_depth, tuple(cfg.background)
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
first = None
for combo in itertools.product(*[vals for _, vals in sweeps]):
    for (var, _), val in zip(sweeps, combo):
        if val == "-":
            os.environ.pop(var, None)
        else:
            os.environ[var] = val
    scene = rt.Scene(b, w, 0)
    p = lambda off: rt.render_params(W, H, spp, D, 0x5EED0002, sample_offset=off, background=bg, flags=rt.RT_FLAG_TIME_TRACE)
    scene.render_device(cam, p(0), acc.data_ptr(), None, None)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter(); rays = 0; tr = 0.0; ext = 0.0
    for k in range(steps):
        st = scene.render_device(cam, p(spp * (k + 1)), acc.data_ptr(), None, None)
        rays += st.rays; tr += st.kernel_ms_trace; ext += st.kernel_ms_extend
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    note = ""
    if check:
        acc.zero_()
        scene.render_device(cam, rt.render_params(W, H, min(spp, 4), D, 0x5EED0002, background=bg), acc.data_ptr(), None, None)
        img = acc.cpu().numpy().copy()
        if first is None:
            first = img
        note = "  image %s" % ("identical" if np.array_equal(img, first) else "DIFFERS (%.2e of pixels)" % np.mean(np.any(img != first, axis=2)))
    print("%-40s %8.2f ms/step  %9.1f Mrays/s   k_trace %7.2f ms/step  bounce loop %7.2f ms/step%s" %
          (" ".join("%s=%s" % (v, c) for (v, _), c in zip(sweeps, combo)), dt / steps * 1e3, rays / dt / 1e6, tr / steps, ext / steps, note), flush=True)
    scene.close()
