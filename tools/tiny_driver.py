"""Renders final_scene.yml at 64x64, 4 spp, three times: per-launch floor of the bounce kernels under rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracing_amd as rt
import torch
path = os.path.join(ROOT, "configs", "final_scene.yml")
cfg = rt.load_config(path); b = rt.Builder()
scene = rt.Scene(b, b.scene_from_config(path, 0x5EED0001), 0)
cam = rt.camera_from_config(cfg)
acc = torch.zeros((64, 64, 3), dtype=torch.float32, device="cuda:0")
for i in range(3):
    st = scene.render_device(cam, rt.render_params(64, 64, 4, cfg.max_depth, 0x5EED0002, sample_offset=4 * i), acc.data_ptr(), None, None)
    print("%.3f ms, %d rays" % (st.seconds * 1e3, st.rays))
