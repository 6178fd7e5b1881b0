"""Stack depth in use by k_trace (instrumented build, RT_DEBUG_WORK): how deep the traversal stack really gets on the bench workloads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RT_DEBUG_WORK"] = "1"
import ray_tracing_amd as rt
SKY = (0.7, 0.8, 1.0)
for name in sys.argv[1:] or ["final_scene.yml", "cornell_box.yml", "random.yml", "synthetic1m"]:
    b = rt.Builder()
    if name == "synthetic1m":
        w = b.canned_scene("Synthetic:1000000", 0x5EED0001)
        cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
        W, H, D, bg = 1920, 1080, 50, SKY
    else:
        path = os.path.join(ROOT, "configs", name)
        cfg = rt.load_config(path); w = b.scene_from_config(path, 0x5EED0001); cam = rt.camera_from_config(cfg)
        W, H, D, bg = cfg.image_width, cfg.image_height, cfg.max_depth, SKY if name == "random.yml" else (0, 0, 0)
    scene = rt.Scene(b, w, 0)
    info = scene.info()
    print("== %s: max_stack %d, nodes %d" % (name, info.max_stack, info.n_nodes), flush=True)
    sys.stderr.flush()
    _, _, st = scene.render(cam, rt.render_params(W, H, 4, D, 0x5EED0002, background=bg, flags=rt.RT_FLAG_COUNTERS))
    print("   rays %d" % st.rays, flush=True)
    scene.close()
