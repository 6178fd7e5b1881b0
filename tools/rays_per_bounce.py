import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ray_tracing_amd as rt, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
path = os.path.join(ROOT, "configs", "final_scene.yml")
cfg = rt.load_config(path); b = rt.Builder()
scene = rt.Scene(b, b.scene_from_config(path, 0x5EED0001), 0); cam = rt.camera_from_config(cfg)
acc = torch.zeros((cfg.image_height, cfg.image_width, 3), dtype=torch.float32, device="cuda:0")
p = rt.render_params(cfg.image_width, cfg.image_height, 400, cfg.max_depth, 0x5EED0002)
st = scene.render_device(cam, p, acc.data_ptr(), None, None)
print(" ".join("%.2f" % (x / 1e6) for x in list(st.rays_per_bounce)[:50]))
