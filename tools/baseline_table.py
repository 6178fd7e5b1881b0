"""One row per BASELINE config: Mrays/s of the CPU oracle (f64 restatement of the reference, on this box's host cores) and of
one MI355X, on the same scene.  usage: python3 tools/baseline_table.py [threads]   (prints markdown rows and a JSON line)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import helpers  # noqa: E402
import ray_tracing_amd as rt  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(16, os.cpu_count() or 1)
SKY = (0.7, 0.8, 1.0)
rows = []
for name, gpu_spp, cpu_spp, bg in (("rectangle.yml", 400, 16, (0, 0, 0)), ("orig_img.yml", 400, 16, SKY), ("cornell.yml", 400, 16, (0, 0, 0)),
                                   ("final_scene.yml", 400, 16, (0, 0, 0)), ("Synthetic:1000000", 32, 1, SKY)):
    if name.startswith("Synthetic"):
        b, ob = rt.Builder(), helpers.OracleBuilder()
        world = b.canned_scene(name, helpers.SCENE_SEED)
        ob.set_world(ob.canned_scene(name, helpers.SCENE_SEED))
        cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
        W, H, D = 3840, 2160, 50
        cW, cH = 960, 540                                   # the oracle renders a quarter-size image of the same view
    else:
        cfg, cam, b, world, ob = helpers.scene_pair(name)
        W, H, D = cfg.image_width, cfg.image_height, cfg.max_depth
        cW, cH = W, H
    scene = rt.Scene(b, world, 0)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    best = 0.0
    for rep in range(3):
        st = scene.render_device(cam, rt.render_params(W, H, gpu_spp, D, helpers.RENDER_SEED, sample_offset=rep * gpu_spp, background=bg), acc.data_ptr(), None, None)
        best = max(best, st.rays / st.seconds / 1e6)
    scene.close()
    t0 = time.perf_counter()
    _, _, ost = helpers.oracle_render(ob, helpers.OracleCamera(cam), cW, cH, cpu_spp, D, helpers.RENDER_SEED, background=bg, n_threads=threads)
    dt = time.perf_counter() - t0
    rows.append(dict(config=name, image="%dx%d" % (W, H), gpu_spp=gpu_spp, gpu_mrays=best, cpu_image="%dx%d" % (cW, cH), cpu_spp=cpu_spp,
                     cpu_mrays=ost["rays"] / dt / 1e6, cpu_seconds=dt, threads=threads))
    r = rows[-1]
    print("| %s | %s | %d / %d | %.1f (%d threads) | %.0f | %.0fx |" % (name, r["image"], gpu_spp, cpu_spp, r["cpu_mrays"], threads, best, best / r["cpu_mrays"]), flush=True)
print(json.dumps(rows))
