"""First-light GPU script (not collected by pytest): all configs, primary hits + small renders vs the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import ray_tracing_amd as rt
from helpers import *
print("devices", rt.device_count(), flush=True)
for name, W, spp in [("orig_img.yml",96,32), ("rectangle.yml",96,32), ("cfg.yml",96,16), ("two_spheres.yml",96,16), ("random.yml",96,16), ("cornell_box.yml",64,32), ("cornell.yml",64,32), ("final_scene.yml",64,32)]:
    cfg, cam, b, world, ob = scene_pair(name)
    H = int(round(W / cfg.aspect_ratio))
    scene = rt.Scene(b, world, 0)
    bg = (0.7, 0.8, 1.0) if name in ("orig_img.yml","cfg.yml","two_spheres.yml","random.yml") else (0,0,0)
    p = rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED, background=bg)
    oc = OracleCamera(cam)
    g = scene.trace_primary(cam, p, 0, True, True)
    o = oracle_trace_primary(ob, oc, W, H, 0, RENDER_SEED, True)
    agree, rel = primary_agreement(g, o)
    oa, osq, ost = oracle_render(ob, oc, W, H, spp, cfg.max_depth, RENDER_SEED, background=bg, want_sumsq=True)
    ga, gsq, st = scene.render(cam, p, want_sumsq=True)
    om, gm = oa/spp, ga/spp
    var = np.maximum(osq/spp - om**2, 0); noise = np.sqrt(var.mean()/spp); rmse = np.sqrt(((om-gm)**2).mean())
    exact = np.mean(np.all(np.abs(om-gm) <= 1e-4*(1+np.abs(om)), axis=2))
    print(f"{name:18s} {W}x{H}x{spp} prim agree={agree:.6f} relt={rel:.2e} | mean orc={om.mean():.5f} gpu={gm.mean():.5f} rmse/noise={rmse/max(noise,1e-12):.4f} pix_same={exact:.4f} rays orc={ost['rays']} gpu={st.rays} | {st.seconds*1e3:.1f} ms", flush=True)
    scene.close()
# throughput probe at full size
for name, spp in [("cornell.yml", 32), ("final_scene.yml", 16)]:
    cfg, cam = config_camera(name)
    b = rt.Builder(); world = b.scene_from_config(os.path.join(CONFIGS, name), SCENE_SEED)
    scene = rt.Scene(b, world, 0)
    p = rt.render_params(cfg.image_width, cfg.image_height, spp, cfg.max_depth, RENDER_SEED)
    for it in range(3):
        acc, _, st = scene.render(cam, p)
        print(f"{name} {cfg.image_width}x{cfg.image_height}x{spp}: {st.seconds*1e3:.1f} ms, rays {st.rays}, {st.rays/st.seconds/1e6:.1f} Mrays/s, bounce-loop {st.kernel_ms_extend:.1f} ms, passes {st.passes}, mean {acc.mean()/spp:.4f}", flush=True)
    print("  rays/bounce", list(st.rays_per_bounce[:12]), flush=True)
    scene.close()
