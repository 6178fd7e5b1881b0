"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/rt_oracle.cpp).

The reference (a Rust crate) cannot be built or run in this environment and ships no fixtures of its
own, so these vectors come from the oracle, not from the reference: they guard the oracle against
regressions and give the GPU tests fixed expected outputs.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

CASES = [("orig_img.yml", 48, 8, (0.7, 0.8, 1.0)), ("rectangle.yml", 48, 8, (0, 0, 0)), ("cfg.yml", 48, 4, (0.7, 0.8, 1.0)),
         ("two_spheres.yml", 48, 4, (0.7, 0.8, 1.0)), ("random.yml", 48, 4, (0.7, 0.8, 1.0)), ("cornell_box.yml", 40, 8, (0, 0, 0)),
         ("cornell.yml", 40, 8, (0, 0, 0)), ("final_scene.yml", 40, 8, (0, 0, 0)), ("showcase.yml", 48, 8, (0.02, 0.03, 0.06))]


def make(name, width, spp, background):
    cfg, cam, b, world, ob = helpers.scene_pair(name)
    height = int(round(width / cfg.aspect_ratio))
    oc = helpers.OracleCamera(cam)
    prim = helpers.oracle_trace_primary(ob, oc, width, height, 0, helpers.RENDER_SEED, True)
    accum, _, st = helpers.oracle_render(ob, oc, width, height, spp, cfg.max_depth, helpers.RENDER_SEED, background=background)
    return dict(width=width, height=height, spp=spp, max_depth=cfg.max_depth, background=np.array(background, np.float64),
                scene_seed=helpers.SCENE_SEED, render_seed=helpers.RENDER_SEED, prim_id=prim["id"], prim_t=prim["t"],
                prim_normal=prim["normal"].astype(np.float32), accum=accum.astype(np.float32), rays=st["rays"])


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, width, spp, bg in CASES:
        if only and name not in only:
            continue
        out = os.path.join(HERE, name.replace(".yml", ".npz"))
        np.savez_compressed(out, **make(name, width, spp, bg))
        print(out, os.path.getsize(out))
