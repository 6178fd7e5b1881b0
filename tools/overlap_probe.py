"""Probe: do two half-size renders on two streams (two scene handles, two host threads) finish sooner than one full-size render?
usage: python3 tools/overlap_probe.py <config.yml> <spp>"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ray_tracing_amd as rt  # noqa: E402

name, spp = sys.argv[1], int(sys.argv[2])
nthr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
path = os.path.join(ROOT, "configs", name)
cfg = rt.load_config(path)
cam = rt.camera_from_config(cfg)
scenes, accs, streams = [], [], []
for k in range(nthr):
    b = rt.Builder()
    scenes.append(rt.Scene(b, b.scene_from_config(path, 0x5EED0001), 0))
    accs.append(torch.zeros((cfg.image_height, cfg.image_width, 3), dtype=torch.float32, device="cuda:0"))
    streams.append(torch.cuda.Stream())


def run(k, n, off):
    p = rt.render_params(cfg.image_width, cfg.image_height, n, cfg.max_depth, 0x5EED0002, sample_offset=off)
    return scenes[k].render_device(cam, p, accs[k].data_ptr(), None, streams[k].cuda_stream)


for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); st = run(0, spp, 0); torch.cuda.synchronize(); t1 = time.perf_counter()
    res = [None] * nthr
    def work(k):
        res[k] = run(k, spp // nthr, k * (spp // nthr))
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
    t2 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    rays2 = sum(int(r.rays) for r in res)
    print("one stream: %.2f ms (%.0f Mrays/s)   %d streams: %.2f ms (%.0f Mrays/s)" % ((t1 - t0) * 1e3, st.rays / (t1 - t0) / 1e6, nthr, (t3 - t2) * 1e3, rays2 / (t3 - t2) / 1e6), flush=True)
