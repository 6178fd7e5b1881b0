#!/usr/bin/env python
"""Benchmark of the render path: whole-job Mrays/s on final_scene.yml (default) or cornell.yml.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch: `--spp-per-step` samples of every pixel of the config's image on
each GPU (weak scaling: per-GPU work is fixed), followed by the reduce(SUM) of the per-GPU accumulation buffers to
rank 0 over RCCL.  Scene and buffers are resident in HBM before the timed region.  A ray = one top-level closest-hit
query (one execution of src/lib.rs:63), counted on the device.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import ray_tracing_amd as rt  # noqa: E402
from ray_tracing_amd.distributed import reduce_accumulators  # noqa: E402

SCENE_SEED, RENDER_SEED = 0x5EED0001, 0x5EED0002
HBM_PEAK_GBS = 8000.0        # HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (packed FMA)
LDS_PEAK_GBS = 150000.0      # aggregate ds_read_b64/b128 rate with every CU streaming (256 B/clk/CU)

# Algorithmic work per unit: FLOPs (FMA = 2) and compulsory bytes, charged per counted event.
FLOPS = dict(node=2 * 24, sphere=26, msphere=35, rect=12, box=30, instance=35, medium=20 + 2 * 26, raygen=45,
             finalize=20, lambertian=30, metal=40, dielectric=60, light=0, isotropic=20, checker=10, noise=7 * 190 + 10)
# k_trace reads a ray (origin+time, direction+source primitive: 2 x 16 B) and writes a hit record (16 B); every
# visited node is one 64 B record, every primitive test one record of the size below.
TRACE_BYTES = dict(ray=32, hit=16, node=64, sphere=16, msphere=48, rect=32, box=32, instance=32)
# the rest of a bounce (k_shade): ray 48 B + hit 16 B in, scattered ray 48 B out, contribution slot 16 B
SHADE_BYTES = dict(ray_in=64, ray_out=48, contrib=16, material=32)


def per_ray_work(st):
    """Average algorithmic work per ray from the instrumented kernels' counters: (k_trace FLOPs, k_trace bytes,
    whole-bounce FLOPs, whole-bounce bytes, event counts)."""
    w = list(st.work)
    rays, paths = float(st.rays), float(st.paths)
    shades = dict(lambertian=w[8], metal=w[9], dielectric=w[10], light=w[11], isotropic=w[12])
    n_shade = sum(shades.values())
    tests = dict(node=w[0], sphere=w[1], msphere=w[2], rect=w[3], instance=w[4], box=w[7])
    trace_flops = sum(tests[k] * FLOPS[k] for k in tests)
    trace_bytes = rays * (TRACE_BYTES["ray"] + TRACE_BYTES["hit"]) + sum(tests[k] * TRACE_BYTES[k] for k in tests)
    flops = (trace_flops + w[5] * FLOPS["medium"] + paths * FLOPS["raygen"] + n_shade * FLOPS["finalize"] +
             sum(shades[k] * FLOPS[k] for k in shades) + w[13] * FLOPS["noise"] + w[14] * FLOPS["checker"])
    byts = (trace_bytes + rays * SHADE_BYTES["ray_in"] + n_shade * (SHADE_BYTES["ray_out"] + SHADE_BYTES["material"]) +
            paths * (48 + SHADE_BYTES["contrib"] * 2))
    counts = dict(nodes=w[0] / rays, sphere_tests=w[1] / rays, msphere_tests=w[2] / rays, rect_tests=w[3] / rays, box_tests=w[7] / rays,
                  instance_entries=w[4] / rays, medium_tests=w[5] / rays, miss_fraction=w[6] / rays, noise_shades=w[13] / rays,
                  rays_per_path=rays / paths)
    return trace_flops / rays, trace_bytes / rays, flops / rays, byts / rays, counts


class StressConfig:
    """BASELINE config 5: 1 000 000 random spheres at 3840x2160, depth 50 (the deep-BVH traversal stress scene; sky background,
    since nothing in it emits)."""
    image_width, image_height, max_depth = 3840, 2160, 50
    background = (0.7, 0.8, 1.0)
    scene = "Synthetic:1000000"
    camera = dict(look_from=(0, 260, 1500), look_at=(0, 60, 0), v_up=(0, 1, 0), vfov=40.0, aspect=16 / 9, aperture=0.1, focus_dist=10.0)


def load_workload(name, device):
    b = rt.Builder()
    if name == "synthetic1m":
        cfg = StressConfig()
        return cfg, rt.make_camera(**cfg.camera), rt.Scene(b, b.canned_scene(cfg.scene, SCENE_SEED), device)
    path = os.path.join(ROOT, "configs", name)
    cfg = rt.load_config(path)
    scene = rt.Scene(b, b.scene_from_config(path, SCENE_SEED), device)
    return cfg, rt.camera_from_config(cfg), scene


def run_steps(scene, cam, cfg, spp, steps, first_step, rank, world, accum, stream_ptr):
    rays = kernel_ms = launches = trace_ms = trace_launches = 0
    for k in range(steps):
        accum.zero_()
        offset = ((first_step + k) * world + rank) * spp          # disjoint sample ranges across ranks and steps
        p = rt.render_params(cfg.image_width, cfg.image_height, spp, cfg.max_depth, RENDER_SEED, sample_offset=offset,
                             background=tuple(getattr(cfg, "background", (0.0, 0.0, 0.0))), flags=rt.RT_FLAG_TIME_TRACE)   # HIP event pairs around every k_trace launch, on the render stream
        st = scene.render_device(cam, p, accum.data_ptr(), None, stream_ptr)
        reduce_accumulators(accum, 0)
        rays += int(st.rays); kernel_ms += st.kernel_ms_extend; launches += int(st.extend_launches)
        trace_ms += st.kernel_ms_trace; trace_launches += int(st.trace_launches)
    return rays, kernel_ms, launches, trace_ms, trace_launches


def cpu_baseline(config_name, cfg, cam, spp, threads):
    """The f64 CPU oracle (a restatement of the reference, not the Rust binary) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    _, _, _, _, ob = helpers.scene_pair(config_name)
    oc = helpers.OracleCamera(cam)
    t0 = time.perf_counter()
    _, _, st = helpers.oracle_render(ob, oc, cfg.image_width, cfg.image_height, spp, cfg.max_depth, RENDER_SEED, n_threads=threads)
    dt = time.perf_counter() - t0
    return dict(value=st["rays"] / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample="%s %dx%d, %d spp, depth %d: %d rays in %.2f s on %d threads (f64 oracle, recursive pointer BVH)"
                       % (config_name, cfg.image_width, cfg.image_height, spp, cfg.max_depth, st["rays"], dt, threads))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="final_scene.yml", choices=["final_scene.yml", "cornell.yml", "synthetic1m"])
    ap.add_argument("--spp-per-step", type=int, default=400)
    ap.add_argument("--cpu-spp", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configs of the metric (cornell.yml, the 1M-sphere scene)")
    ap.add_argument("--stress-spp", type=int, default=32, help="samples per step of the 1M-sphere 3840x2160 run reported under `also`")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks share GPU 0 and talk over gloo: runs the N>1 code path on a one-GPU box; the line is tagged and is not a scaling result")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if not torch.cuda.is_available() or rt.device_count() <= local_rank:
        raise SystemExit("bench.py needs a GPU per rank; the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(config_name, steps, warmup, spp=None):
        spp = spp or args.spp_per_step
        cfg, cam, scene = load_workload(config_name, local_rank)
        accum = torch.zeros((cfg.image_height, cfg.image_width, 3), dtype=torch.float32, device=device)
        stream_ptr = torch.cuda.current_stream().cuda_stream
        run_steps(scene, cam, cfg, spp, warmup, 0, rank, world, accum, stream_ptr)
        barrier()
        t0 = time.perf_counter()
        rays, kernel_ms, launches, trace_ms, trace_launches = run_steps(scene, cam, cfg, spp, steps, warmup, rank, world, accum, stream_ptr)
        barrier()
        dt = time.perf_counter() - t0
        tot = torch.tensor([dt, float(rays)], dtype=torch.float64, device=device)
        if world > 1:
            mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            dt, total_rays = float(mx[0]), float(sm[1])
        else:
            total_rays = float(rays)
        mean = float(accum.mean()) / (spp * world)                  # rank 0: image mean of the last step (sanity only)
        return dict(cfg=cfg, cam=cam, scene=scene, dt=dt, total_rays=total_rays, rays0=rays, kernel_ms=kernel_ms, launches=launches, trace_ms=trace_ms,
                    trace_launches=trace_launches, mean=mean)

    main_run = timed(args.workload, args.steps, args.warmup)
    cfg, cam, scene = main_run["cfg"], main_run["cam"], main_run["scene"]
    out = None
    if rank == 0:
        # work per ray of this workload from the instrumented kernel (outside the timed region)
        acc_np, _, cst = scene.render(cam, rt.render_params(cfg.image_width, cfg.image_height, 2, cfg.max_depth, RENDER_SEED, flags=rt.RT_FLAG_COUNTERS))
        tr_flops, tr_bytes, flops_per_ray, bytes_per_ray, counts = per_ray_work(cst)
        rays0 = main_run["rays0"]
        t_s = main_run["trace_ms"] * 1e-3                           # sum of the k_trace launches of the timed steps
        b_s = main_run["kernel_ms"] * 1e-3                          # raygen-to-last-shade span of the timed steps
        fused = t_s < 0.05 * b_s                                    # tiny scene: the closest hit is found inside the shading kernel
        if fused:
            t_s = b_s                                               # ... which is then the dominant (only) bounce kernel
        n_tr = max(main_run["trace_launches"], 1)
        hbm_bytes = TRACE_BYTES["ray"] + TRACE_BYTES["hit"]        # per ray: the queue record in, the hit record out -- what crosses HBM
        if fused:
            hbm_bytes = SHADE_BYTES["ray_in"] - 16 + SHADE_BYTES["ray_out"] + SHADE_BYTES["contrib"]   # fused kernel: ray in (no hit record), ray out, contribution
        gbs_hbm = hbm_bytes * rays0 / t_s / 1e9
        gbs_lds = tr_bytes * rays0 / t_s / 1e9                      # with the node / primitive records charged although they are served from LDS
        tfl = tr_flops * rays0 / t_s / 1e12
        wl = args.workload.replace(".yml", "")
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % wl)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), "profiles/traffic_%s.json (%s)" % (wl, tj.get("source", "rocprofv3 FETCH_SIZE / WRITE_SIZE passes, committed"))
        issue = None   # VALU issue figures of the same workload from the committed PMC passes (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, ...)
        vpath = os.path.join(ROOT, "profiles", "valu_%s.json" % wl)
        if os.path.exists(vpath):
            vj = json.load(open(vpath))
            issue = vj.get("kernels", {}).get("k_trace")
            if issue is not None:
                issue = dict(issue, measured_in_this_run=False, source="profiles/valu_%s.json <- %s" % (wl, vj.get("source", "committed rocprofv3 --pmc passes")))
        paths = cfg.image_width * cfg.image_height * args.spp_per_step * world
        kname = ("k_shade_lds<fused> (closest hit + shading in one kernel; k_trace is not run for this scene)" if fused else
                 "k_trace (closest hit of every queued ray: BVH2 traversal + primitive tests)")
        out = {
            "metric": "Mrays/sec (whole node)", "value": main_run["total_rays"] / main_run["dt"] / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_run["dt"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, max_depth %d, %d spp per step per GPU (scene seed 0x%X, Philox render seed 0x%X)"
                                   % (args.workload, cfg.image_width, cfg.image_height, cfg.max_depth, args.spp_per_step, SCENE_SEED, RENDER_SEED),
                       "paths_per_step": paths, "rays_per_path": main_run["total_rays"] / args.steps / paths,
                       "mpaths_per_s": paths * args.steps / main_run["dt"] / 1e6,
                       "parallelism": "sample-sharded x%d, reduce(SUM) to rank 0" % world, "image_mean": main_run["mean"]},
            # HBM line of the dominant kernel: the bytes that cross HBM (ray record in, hit record out), measured launch time
            "roofline": {"kernel": kname, "bound": "hbm", "achieved": gbs_hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_hbm / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_ray": hbm_bytes, "rays_per_launch": rays0 / n_tr, "bytes_per_launch": hbm_bytes * rays0 / n_tr,
                         "avg_launch_ms": t_s * 1e3 / n_tr, "launches": n_tr,
                         "share_of_bounce_time": t_s / b_s, "mrays_per_s_in_kernel": rays0 / t_s / 1e6,
                         "note": "HBM is not what binds this kernel: the scene (<1 MB) is served from the block's LDS copy and the kernel is bound by "
                                 "VALU instruction issue of divergent traversal at partial lane occupancy -- see roofline_valu.  lds_charged "
                                 "prices the node and primitive records too (64 B per visited node + the record of every primitive tested); those "
                                 "bytes come from LDS, so that figure is compared with the LDS peak, never with HBM"},
            "roofline_valu": {"kernel": "k_trace", "bound": "valu_fp32", "achieved": tfl, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": tfl / VALU_PEAK_TFLOPS, "flops_per_ray": tr_flops,
                              "note": "algorithmic FLOPs of live lanes only (FMA = 2); peak = 256 CUs x 128 lanes x 2 x 2.4 GHz, reachable only with packed fp32",
                              "issue": issue},
            "lds_charged": {"bytes_per_ray": tr_bytes, "achieved": gbs_lds, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": gbs_lds / LDS_PEAK_GBS,
                            "note": "ray + hit records + every node and primitive record read, against the aggregate LDS read rate (ds_read_b128, all CUs)"},
            "bounce_total": {"kernels": "k_trace + k_shade", "ms": main_run["kernel_ms"], "launch_pairs": main_run["launches"],
                             "flops_per_ray": flops_per_ray, "bytes_per_ray": bytes_per_ray,
                             "achieved_gbs": bytes_per_ray * rays0 / b_s / 1e9, "achieved_tflops": flops_per_ray * rays0 / b_s / 1e12},
            "work_per_ray": counts,
        }
        if args.rehearse_on_one_gpu:
            out["rehearsal"] = "%d ranks share one GPU and reduce over gloo: a run of the multi-rank code path, not a scaling measurement" % world
    if not args.no_secondary:
        main_run["scene"].close()
        also = {}
        for other in ("cornell.yml", "final_scene.yml", "synthetic1m"):
            if other == args.workload:
                continue
            stress = other == "synthetic1m"
            k = 2 if stress else max(1, args.steps // 2)
            sec = timed(other, k, 1, args.stress_spp if stress else None)
            if rank == 0:
                c = sec["cfg"]
                also[other] = {"value": sec["total_rays"] / sec["dt"] / 1e6, "unit": "Mrays/s", "steps": k, "ms_per_step": sec["dt"] / k * 1e3,
                               "workload": "%s: %dx%d, max_depth %d, %d spp per step per GPU" % (getattr(c, "scene", other), c.image_width, c.image_height,
                                                                                                  c.max_depth, args.stress_spp if stress else args.spp_per_step),
                               "image_mean": sec["mean"]}
                if stress:
                    n_t = max(sec["trace_launches"], 1)
                    also[other]["k_trace"] = {"avg_launch_ms": sec["trace_ms"] / n_t, "launches": sec["trace_launches"],
                                              "share_of_bounce_time": sec["trace_ms"] / max(sec["kernel_ms"], 1e-9),
                                              "mrays_per_s_in_kernel": sec["rays0"] / max(sec["trace_ms"] * 1e-3, 1e-12) / 1e6}
                    spath = os.path.join(ROOT, "profiles", "memory_synthetic1m.json")   # L2 / fabric counters of this workload (committed PMC passes)
                    if os.path.exists(spath):
                        also[other]["roofline"] = dict(json.load(open(spath)), measured_in_this_run=False, source="profiles/memory_synthetic1m.json")
            sec["scene"].close()
        if rank == 0:
            out["also"] = also
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(args.workload, cfg, cam, args.cpu_spp, threads)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
