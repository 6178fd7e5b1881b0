"""Build time and traversal rate of the two top-level builders on the sphere-field stress scene.
usage: python3 tools/lbvh_probe.py [n_spheres] [spp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import ray_tracing_amd as rt  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b = rt.Builder()
w = b.canned_scene("Synthetic:%d" % n, 0x5EED0001)
cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
p = rt.render_params(3840, 2160, spp, 50, 0x5EED0002, background=(0.7, 0.8, 1.0))
acc = torch.zeros((2160, 3840, 3), dtype=torch.float32, device="cuda:0")
for label, flags in (("binned SAH on the host", 0), ("linear BVH on the GPU", rt.RT_BVH_LINEAR)):
    t0 = time.perf_counter()
    scene = rt.Scene(b, w, 0, flags)
    t1 = time.perf_counter()
    info = scene.info()
    scene.render_device(cam, p, acc.data_ptr(), None, None)
    st = scene.render_device(cam, p, acc.data_ptr(), None, None)
    print("%s: scene create %.3f s (top-level build on the device %.4f s), %d nodes, depth %d; %d spp at 4K: %.1f Mrays/s"
          % (label, t1 - t0, info.bvh_build_seconds, info.n_nodes, info.tlas_depth, spp, st.rays / st.seconds / 1e6), flush=True)
    scene.close()
