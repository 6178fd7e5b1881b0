#!/usr/bin/env python
"""Benchmark of the render path: whole-job Mrays/s on final_scene.yml (default), cornell.yml or the 1M-sphere scene.

    python bench.py --gpus N --steps K --warmup W
        N = 1: this process renders on GPU 0.
        N > 1: this process starts N rank processes (one per GPU, torch.distributed over RCCL, 127.0.0.1 rendezvous) before it
               touches any GPU, relays rank 0's JSON line and exits with the ranks' status.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
        the same ranks under an external launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).
    python bench.py --gpus N --mode inprocess
        one process, one scene handle per GPU, every step one rt_render_multi call (host threads + one RCCL reduce inside the
        C ABI): the path a single-process caller such as the reference's main() would use.

A step = one pass of the hot path over one batch: `--spp-per-step` samples of every pixel of the config's image on each GPU
(weak scaling: per-GPU work is fixed), followed by the reduce(SUM) of the per-GPU accumulation buffers to rank 0 / GPU 0.
Scene and buffers are resident in HBM before the timed region.  A ray = one top-level closest-hit query (one execution of
src/lib.rs:63), counted on the device.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE_SEED, RENDER_SEED = 0x5EED0001, 0x5EED0002
HBM_PEAK_GBS = 8000.0        # HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (packed FMA)
LDS_PEAK_GBS = 150000.0      # aggregate ds_read_b64/b128 rate with every CU streaming (256 B/clk/CU)

# Algorithmic work per unit: FLOPs (FMA = 2) and compulsory bytes, charged per counted event.
FLOPS = dict(node=2 * 24, wide_node=4 * 24, sphere=26, msphere=35, rect=12, box=30, instance=35, medium=20 + 2 * 26, raygen=45,
             finalize=20, lambertian=30, metal=40, dielectric=60, light=0, isotropic=20, checker=10, noise=7 * 190 + 10)
# k_trace reads a ray (origin+time, direction+source primitive: 2 x 16 B) and writes a hit record (16 B); every
# visited node is one 64 B record (binary or 4-wide), every primitive test one record of the size below.
TRACE_BYTES = dict(ray=32, hit=16, node=64, wide_node=64, sphere=16, msphere=48, rect=32, box=32, instance=32)
# the rest of a bounce (k_shade): ray 48 B + hit 16 B in, scattered ray 48 B out, contribution slot 16 B
SHADE_BYTES = dict(ray_in=64, ray_out=48, contrib=16, material=32)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="ranks", choices=["ranks", "inprocess"],
                    help="N > 1: one process per GPU over torch.distributed (ranks), or one process calling rt_render_multi (inprocess)")
    ap.add_argument("--workload", default="final_scene.yml", choices=["final_scene.yml", "cornell.yml", "synthetic1m"])
    ap.add_argument("--spp-per-step", type=int, default=400)
    ap.add_argument("--cpu-spp", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-first", action="store_true", help="time the CPU oracle before the GPU legs instead of beside the secondary ones")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configs of the metric (cornell.yml, the 1M-sphere scene)")
    ap.add_argument("--stress-spp", type=int, default=32, help="samples per step of the 1M-sphere 3840x2160 run reported under `also`")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks / handles share GPU 0 (ranks talk over gloo): runs the N>1 code path on a one-GPU box; the line is tagged and is not a scaling result")
    return ap.parse_args()


def _env_int(name):
    v = os.environ.get(name, "").strip()
    return int(v) if v else None


def spawn_ranks(args):
    """Starts one rank process per GPU and relays rank 0's output.  Nothing in this process has touched a GPU (no torch import, no
    library call), and nothing is exec'ed: the ranks are children and this process exits with their status."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill(); rcs.append(-9)
    sys.stdout.write(out or ""); sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit("bench.py: rank processes failed: %s" % bad)


def per_ray_work(st):
    """Average algorithmic work per ray from the instrumented kernels' counters: (k_trace FLOPs, k_trace bytes,
    whole-bounce FLOPs, whole-bounce bytes, event counts)."""
    w = list(st.work)
    rays, paths = float(st.rays), float(st.paths)
    shades = dict(lambertian=w[8], metal=w[9], dielectric=w[10], light=w[11], isotropic=w[12])
    n_shade = sum(shades.values())
    tests = dict(node=w[0], wide_node=w[15], sphere=w[1], msphere=w[2], rect=w[3], instance=w[4], box=w[7])
    trace_flops = sum(tests[k] * FLOPS[k] for k in tests)
    traced = rays - float(st.rays_settled)     # k_trace reads and writes records only for the rays it searches for
    trace_bytes = traced * (TRACE_BYTES["ray"] + TRACE_BYTES["hit"]) + sum(tests[k] * TRACE_BYTES[k] for k in tests)
    flops = (trace_flops + w[5] * FLOPS["medium"] + paths * FLOPS["raygen"] + n_shade * FLOPS["finalize"] +
             sum(shades[k] * FLOPS[k] for k in shades) + w[13] * FLOPS["noise"] + w[14] * FLOPS["checker"])
    byts = (trace_bytes + rays * SHADE_BYTES["ray_in"] + n_shade * (SHADE_BYTES["ray_out"] + SHADE_BYTES["material"]) +
            paths * (48 + SHADE_BYTES["contrib"] * 2))
    counts = dict(nodes=w[0] / rays, wide_nodes=w[15] / rays, sphere_tests=w[1] / rays, msphere_tests=w[2] / rays, rect_tests=w[3] / rays,
                  box_tests=w[7] / rays, instance_entries=w[4] / rays, medium_tests=w[5] / rays, miss_fraction=w[6] / rays, noise_shades=w[13] / rays,
                  rays_per_path=rays / paths, settled_without_the_tree=float(st.rays_settled) / rays)
    return trace_flops / rays, trace_bytes / rays, flops / rays, byts / rays, counts


class StressConfig:
    """BASELINE config 5: 1 000 000 random spheres at 3840x2160, depth 50 (the deep-BVH traversal stress scene; sky background,
    since nothing in it emits)."""
    image_width, image_height, max_depth = 3840, 2160, 50
    background = (0.7, 0.8, 1.0)
    scene = "Synthetic:1000000"
    camera = dict(look_from=(0, 260, 1500), look_at=(0, 60, 0), v_up=(0, 1, 0), vfov=40.0, aspect=16 / 9, aperture=0.1, focus_dist=10.0)


def main():
    args = parse_args()
    world, rank, local_rank = _env_int("WORLD_SIZE"), _env_int("RANK") or 0, _env_int("LOCAL_RANK") or 0
    inprocess = args.mode == "inprocess" and args.gpus > 1
    if world is None and args.gpus > 1 and not inprocess:
        return spawn_ranks(args)
    world = world or 1
    if inprocess and world != 1:
        raise SystemExit("--mode inprocess is one process: do not start it under a launcher")
    if not inprocess and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rt
    from ray_tracing_amd.distributed import reduce_accumulators

    if args.rehearse_on_one_gpu:
        local_rank = 0
    n_handles = args.gpus if inprocess else 1
    handle_devices = [0 if args.rehearse_on_one_gpu else i for i in range(n_handles)] if inprocess else [local_rank]
    if not torch.cuda.is_available() or rt.device_count() <= max(handle_devices):
        raise SystemExit("bench.py needs a GPU per rank; the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if inprocess and args.rehearse_on_one_gpu:
        os.environ["RT_MULTI_FORCE_RCCL"] = "1"     # one GPU: still go through the (one-rank) communicator
    n_units = args.gpus                              # GPUs (or rehearsal stand-ins) the job's samples are sharded over

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def load_workload(name):
        """(cfg, camera, [one scene handle per GPU this process drives])"""
        b = rt.Builder()
        if name == "synthetic1m":
            cfg = StressConfig()
            w = b.canned_scene(cfg.scene, SCENE_SEED)
            return cfg, rt.make_camera(**cfg.camera), [rt.Scene(b, w, d) for d in handle_devices]
        path = os.path.join(ROOT, "configs", name)
        cfg = rt.load_config(path)
        w = b.scene_from_config(path, SCENE_SEED)
        return cfg, rt.camera_from_config(cfg), [rt.Scene(b, w, d) for d in handle_devices]

    def params(cfg, spp, offset, flags=0):
        return rt.render_params(cfg.image_width, cfg.image_height, spp, cfg.max_depth, RENDER_SEED, sample_offset=offset,
                                background=tuple(getattr(cfg, "background", (0.0, 0.0, 0.0))), flags=flags)

    def run_steps(scenes, cam, cfg, spp, steps, first_step, accum, stream_ptr):
        rays = kernel_ms = launches = trace_ms = trace_launches = settled = first = paths = 0
        seen = (0, 0)
        for k in range(steps):
            if inprocess:          # one rt_render_multi call: the samples of all GPUs of this step, sharded inside the library
                accum.fill(0.0)
                _, _, st = rt.render_multi(scenes, cam, params(cfg, spp * n_units, (first_step + k) * n_units * spp, rt.RT_FLAG_TIME_TRACE), accum=accum)
                seen = (int(st.multi_gpus), int(st.multi_comm_ranks))
            else:
                accum.zero_()
                offset = ((first_step + k) * world + rank) * spp      # disjoint sample ranges across ranks and steps
                # RT_FLAG_TIME_TRACE: HIP event pairs around every k_trace launch, on the render stream
                st = scenes[0].render_device(cam, params(cfg, spp, offset, rt.RT_FLAG_TIME_TRACE), accum.data_ptr(), None, stream_ptr)
                reduce_accumulators(accum, 0)
            rays += int(st.rays); kernel_ms += st.kernel_ms_extend; launches += int(st.extend_launches)
            trace_ms += st.kernel_ms_trace; trace_launches += int(st.trace_launches); settled += int(st.rays_settled)
            first += int(st.rays_per_bounce[0]); paths += int(st.paths)
        return rays, kernel_ms, launches, trace_ms, trace_launches, seen, settled, first, paths

    def timed(config_name, steps, warmup, spp=None):
        spp = spp or args.spp_per_step
        cfg, cam, scenes = load_workload(config_name)
        if inprocess:
            accum = np.zeros((cfg.image_height, cfg.image_width, 3), dtype=np.float32)
        else:
            accum = torch.zeros((cfg.image_height, cfg.image_width, 3), dtype=torch.float32, device=device)
        stream_ptr = torch.cuda.current_stream().cuda_stream
        run_steps(scenes, cam, cfg, spp, warmup, 0, accum, stream_ptr)
        barrier()
        t0 = time.perf_counter()
        rays, kernel_ms, launches, trace_ms, trace_launches, seen, settled, first, paths = run_steps(scenes, cam, cfg, spp, steps, warmup, accum, stream_ptr)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tot = torch.tensor([dt, float(rays)], dtype=torch.float64, device=device)
            mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            dt, total_rays = float(mx[0]), float(sm[1])
        else:
            total_rays = float(rays)
        mean = float(accum.mean()) / (spp * n_units)               # rank 0: image mean of the last step (sanity only)
        # rays0 / kernel times are those of the GPU(s) this process drove; per_gpu divides them for the per-launch figures
        return dict(cfg=cfg, cam=cam, scenes=scenes, dt=dt, total_rays=total_rays, rays0=rays, kernel_ms=kernel_ms, launches=launches, trace_ms=trace_ms,
                    trace_launches=trace_launches, mean=mean, spp=spp, seen=seen, name=config_name, settled=settled, first=first, paths=paths)

    def close(run):
        for s in run["scenes"]:
            s.close()

    def multi_check(run):
        """N > 1 only, outside the timed region: who really took part, and does the reduced image hold every rank's light?"""
        cfg, cam, scenes = run["cfg"], run["cam"], run["scenes"]
        spp = 2
        if inprocess:
            acc = np.zeros((cfg.image_height, cfg.image_width, 3), dtype=np.float32)
            _, _, st = rt.render_multi(scenes, cam, params(cfg, spp * n_units, 0), accum=acc)
            one, _, st1 = scenes[0].render(cam, params(cfg, spp * n_units, 0))
            return dict(ranks_seen=int(st.multi_comm_ranks), gpus_seen=int(st.multi_gpus), handles=len(scenes),
                        reduced_sum=float(acc.astype(np.float64).sum()), sum_of_rank_sums=float(one.astype(np.float64).sum()),
                        rays_equal=int(st.rays) == int(st1.rays), what="rt_render_multi image vs one handle rendering the same samples alone")
        acc = torch.zeros((cfg.image_height, cfg.image_width, 3), dtype=torch.float32, device=device)
        scenes[0].render_device(cam, params(cfg, spp, rank * spp), acc.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        mine = acc.double().sum().reshape(1)
        ones = torch.ones(1, dtype=torch.float64, device=device)
        props = torch.cuda.get_device_properties(local_rank)
        ident = "%s:%s" % (socket.gethostname(), getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or local_rank)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if dist.get_backend() == "gloo":
            mine, ones = mine.cpu(), ones.cpu()
        dist.all_reduce(mine); dist.all_reduce(ones)
        reduce_accumulators(acc, 0)
        return dict(ranks_seen=int(round(float(ones[0]))), world_size=dist.get_world_size(), backend=dist.get_backend(), gpus_seen=len(set(idents)),
                    reduced_sum=float(acc.double().sum()), sum_of_rank_sums=float(mine[0]),
                    what="rank 0's reduced image vs the all-reduced sum of every rank's own image, one extra step of %d spp per rank" % spp)

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

    def kernel_roofline(run, tr_flops=None, tr_bytes=None):
        """HBM line of the dominant bounce kernel of a timed run: algorithmic bytes that cross HBM per launch / measured launch time."""
        rays0 = run["rays0"]
        t_s = run["trace_ms"] * 1e-3                              # sum of the k_trace launches of the timed steps (HIP events)
        b_s = run["kernel_ms"] * 1e-3                             # raygen-to-last-shade span of the timed steps (HIP events)
        fused = t_s < 0.05 * b_s                                  # tiny scene: the closest hit is found inside the shading kernel
        if fused:
            t_s = b_s                                             # ... which is then the dominant (only) bounce kernel
        # in-process runs sum rays and launches over the handles and report the slowest GPU's kernel time: per-GPU figures below
        n_tr = max(run["launches"] if fused else run["trace_launches"], 1) / n_handles
        # the rays the kernel handled: k_trace does not see the rays whose closest hit the shading kernel settled (RtStats.rays_settled)
        rays_gpu = (rays0 if fused else rays0 - run["settled"]) / n_handles
        if fused:
            # What the fused kernel moves, from the run's own counts: a camera ray is read as two records, every later ray as three;
            # every ray but a camera ray was written as three records by the launch before; a path's contribution slot is written once.
            later = rays0 - run["first"]
            hbm_bytes = (32.0 * run["first"] + (SHADE_BYTES["ray_in"] - 16) * later + SHADE_BYTES["ray_out"] * later +
                         SHADE_BYTES["contrib"] * run["paths"]) / max(rays0, 1)
        else:
            hbm_bytes = TRACE_BYTES["ray"] + TRACE_BYTES["hit"]
        gbs = hbm_bytes * rays_gpu / t_s / 1e9
        kname = ("k_shade_lds<fused> (closest hit + shading in one kernel; k_trace is not run for this scene)" if fused else
                 "k_trace (closest hit of every queued ray: BVH traversal + primitive tests)")
        r = {"kernel": kname, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
             "bytes_per_ray": hbm_bytes, "rays_per_launch": rays_gpu / n_tr, "bytes_per_launch": hbm_bytes * rays_gpu / n_tr,
             "avg_launch_ms": t_s * 1e3 / n_tr, "launches": n_tr, "timing": "HIP event pairs on the render stream, inside this run",
             "share_of_bounce_time": t_s / b_s, "mrays_per_s_in_kernel": rays_gpu / t_s / 1e6,
             "rays_settled_by_the_shading_kernel_frac": run["settled"] / max(rays0, 1)}
        return r, fused, t_s, b_s

    # ---- the CPU baseline runs on host threads beside the first secondary GPU leg (the oracle call releases the GIL), so that the
    # GPU is busy for more of the run; it is joined before the legs of the metric's own configs, which therefore run on a quiet host
    want_cpu = rank == 0 and args.gpus == 1 and not args.no_cpu_baseline
    cpu_results, cpu_thread = {}, None
    if want_cpu:
        threads = min(16, os.cpu_count() or 1)
        main_path = os.path.join(ROOT, "configs", args.workload)
        jobs = []
        if args.workload != "synthetic1m":
            c = rt.load_config(main_path)
            jobs.append((args.workload, c, rt.camera_from_config(c), args.cpu_spp))
        if not args.no_secondary and args.workload != "cornell.yml":
            c = rt.load_config(os.path.join(ROOT, "configs", "cornell.yml"))
            jobs.append(("cornell.yml", c, rt.camera_from_config(c), max(8, args.cpu_spp // 4)))

        def cpu_leg():
            for name, c, cam_, spp_ in jobs:
                cpu_results[name] = cpu_baseline(name, c, cam_, spp_, threads)
        if args.cpu_baseline_first:
            cpu_leg()
        else:
            cpu_thread = threading.Thread(target=cpu_leg); cpu_thread.start()

    also = {}
    if not args.no_secondary:
        # the 1M-sphere leg (seconds of scene building and long steps) is the one that runs beside the CPU baseline; the legs of the
        # metric's two configs run after the join, on a quiet host: their steps are tens of milliseconds and a busy host delays
        # the wake-up after every step's synchronisation (measured: cornell.yml -6 % beside 16 oracle threads)
        for other in ("synthetic1m", "cornell.yml", "final_scene.yml"):
            if other == args.workload:
                continue
            if other != "synthetic1m" and cpu_thread is not None:
                cpu_thread.join(); cpu_thread = None
            stress = other == "synthetic1m"
            k = 2 if stress else max(1, args.steps // 2)
            sec = timed(other, k, 1, args.stress_spp if stress else None)
            if rank == 0:
                c = sec["cfg"]
                roof, fused, t_s, b_s = kernel_roofline(sec)
                also[other] = {"value": sec["total_rays"] / sec["dt"] / 1e6, "unit": "Mrays/s", "steps": k, "ms_per_step": sec["dt"] / k * 1e3,
                               "workload": "%s: %dx%d, max_depth %d, %d spp per step per GPU" % (getattr(c, "scene", other), c.image_width, c.image_height,
                                                                                                  c.max_depth, sec["spp"]),
                               "image_mean": sec["mean"], "roofline": roof}
                if stress:
                    spath = os.path.join(ROOT, "profiles", "memory_synthetic1m.json")   # L2 / fabric counters of this workload (committed PMC passes)
                    if os.path.exists(spath):
                        also[other]["memory_counters"] = dict(json.load(open(spath)), measured_in_this_run=False, source="profiles/memory_synthetic1m.json")
            close(sec)
    if cpu_thread is not None:
        cpu_thread.join()

    main_run = timed(args.workload, args.steps, args.warmup)
    cfg, cam, scenes = main_run["cfg"], main_run["cam"], main_run["scenes"]
    check = multi_check(main_run) if args.gpus > 1 else None
    out = None
    if rank == 0:
        # work per ray of this workload from the instrumented kernel (outside the timed region)
        _, _, cst = scenes[0].render(cam, params(cfg, 2, 0, rt.RT_FLAG_COUNTERS))
        tr_flops, tr_bytes, flops_per_ray, bytes_per_ray, counts = per_ray_work(cst)
        rays0 = main_run["rays0"]
        roof, fused, t_s, b_s = kernel_roofline(main_run)
        gbs_lds = tr_bytes * rays0 / t_s / 1e9 / n_handles             # with the node / primitive records charged although they are served from LDS
        tfl = tr_flops * rays0 / t_s / 1e12 / n_handles
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
        paths = cfg.image_width * cfg.image_height * args.spp_per_step * n_units
        roof.update({"traffic": traffic, "traffic_source": traffic_src,
                     "note": "HBM is not what binds this kernel: the scene (<1 MB) is served from the block's LDS copy and the kernel is bound by "
                             "VALU instruction issue of divergent traversal at partial lane occupancy -- see roofline_valu.  lds_charged "
                             "prices the node and primitive records too (64 B per visited node + the record of every primitive tested); those "
                             "bytes come from LDS, so that figure is compared with the LDS peak, never with HBM"})
        if inprocess:
            par = "rt_render_multi: %d scene handles in one process, samples sharded inside the library, one RCCL reduce to the first GPU" % n_handles
        else:
            par = "sample-sharded x%d, one process per GPU, reduce(SUM) to rank 0" % world
        out = {
            "metric": "Mrays/sec (whole node)", "value": main_run["total_rays"] / main_run["dt"] / 1e6, "unit": "Mrays/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_run["dt"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, max_depth %d, %d spp per step per GPU (scene seed 0x%X, Philox render seed 0x%X)"
                                   % (args.workload, cfg.image_width, cfg.image_height, cfg.max_depth, args.spp_per_step, SCENE_SEED, RENDER_SEED),
                       "paths_per_step": paths, "rays_per_path": main_run["total_rays"] / args.steps / paths,
                       "mpaths_per_s": paths * args.steps / main_run["dt"] / 1e6,
                       "parallelism": par, "mode": args.mode if args.gpus > 1 else "single", "image_mean": main_run["mean"]},
            # HBM line of the dominant kernel: the bytes that cross HBM (ray record in, hit record out), measured launch time
            "roofline": roof,
            "roofline_valu": {"kernel": "k_trace", "bound": "valu_fp32", "achieved": tfl, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": tfl / VALU_PEAK_TFLOPS, "flops_per_ray": tr_flops,
                              "note": "algorithmic FLOPs of live lanes only (FMA = 2); peak = 256 CUs x 128 lanes x 2 x 2.4 GHz, reachable only with packed fp32",
                              "issue": issue},
            "lds_charged": {"bytes_per_ray": tr_bytes, "achieved": gbs_lds, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": gbs_lds / LDS_PEAK_GBS,
                            "note": "ray + hit records + every node and primitive record read, against the aggregate LDS read rate (ds_read_b128, all CUs)"},
            "bounce_total": {"kernels": "k_trace + k_shade", "ms": main_run["kernel_ms"], "launch_pairs": main_run["launches"],
                             "flops_per_ray": flops_per_ray, "bytes_per_ray_lds_and_hbm": bytes_per_ray,
                             "charged_gbs_lds_and_hbm": bytes_per_ray * rays0 / b_s / 1e9 / n_handles,
                             "charged_note": "every record a ray touches, whether served from LDS or HBM: compare with lds_charged.peak, not with the HBM peak",
                             "achieved_tflops": flops_per_ray * rays0 / b_s / 1e12 / n_handles},
            "work_per_ray": counts,
        }
        if check is not None:
            out["ranks_seen"] = check["ranks_seen"]
            rs, ss = check["reduced_sum"], check["sum_of_rank_sums"]
            check["rel_err"] = abs(rs - ss) / max(abs(ss), 1e-30)
            check["ok"] = bool(check["rel_err"] < 1e-5 and check.get("rays_equal", True))
            out["reduce_check"] = check
        if args.rehearse_on_one_gpu:
            out["rehearsal"] = "%d ranks / handles share one GPU (ranks reduce over gloo): a run of the multi-GPU code path, not a scaling measurement" % args.gpus
        if also:
            for name, res in cpu_results.items():
                if name in also:
                    also[name]["cpu_baseline"] = res
                    also[name]["gpu_over_cpu"] = also[name]["value"] / res["value"]
            out["also"] = also
        if args.workload in cpu_results:
            out["cpu_baseline"] = cpu_results[args.workload]
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["ran"] = "before the GPU legs" if args.cpu_baseline_first else "on host threads beside the 1M-sphere GPU leg, joined before the legs of cornell.yml and final_scene.yml"
        print(json.dumps(out), flush=True)
    close(main_run)
    if inprocess:
        rt.multi_shutdown()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if check is not None and rank == 0 and not check["ok"]:
        raise SystemExit("bench.py: the reduced image does not hold every rank's contribution: %s" % check)


if __name__ == "__main__":
    main()
