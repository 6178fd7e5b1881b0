"""GPU tests of the stress configuration at its stated size (1 000 000 spheres, 3840x2160), of bias under independent
seeds at high sample counts, of the queue-overflow report, and of the multi-GPU entry points at the sizes one GPU allows.
Everything goes through the C ABI (librt_hip.so); the f64 oracle is the checker."""
import os

import numpy as np
import pytest

import ray_tracing_amd as rt
from helpers import CONFIGS, RENDER_SEED, SCENE_SEED, OracleBuilder, OracleCamera, config_camera, oracle_render, oracle_trace_primary, primary_agreement, scene_pair
from parity_checks import ID_AGREEMENT, T_REL_TOL, noise_bound

pytestmark = pytest.mark.gpu

STRESS = "Synthetic:1000000"
STRESS_CAM = dict(look_from=(0, 260, 1500), look_at=(0, 60, 0), v_up=(0, 1, 0), vfov=40.0, aspect=16 / 9, aperture=0.1, focus_dist=10.0)
SKY = (0.7, 0.8, 1.0)


@pytest.fixture(scope="module")
def stress_world():
    """The 1M-sphere field on the product builder and on the oracle (its own builder and its own random-axis median BVH)."""
    b = rt.Builder()
    w = b.canned_scene(STRESS, SCENE_SEED)
    ob = OracleBuilder()
    ob.set_world(ob.canned_scene(STRESS, SCENE_SEED))
    return b, w, ob


@pytest.mark.parametrize("wavefront", [False, True], ids=["per_ray_hook", "render_kernels_wide4"])
@pytest.mark.parametrize("flags", [0, rt.RT_BVH_LINEAR], ids=["sah_tree", "device_built_linear_tree"])
def test_one_million_spheres_primary_hits_match_oracle(gpu_required, stress_world, flags, wavefront):
    """wavefront: the hits come from k_trace over the 4-wide quantised tree (rt_node_step4), the only traversal a render of this
    scene uses, instead of the per-ray walk of the binary tree."""
    b, w, ob = stress_world
    scene = rt.Scene(b, w, 0, flags)
    info = scene.info()
    assert info.n_spheres == 1000001 and info.max_stack <= 32
    cam = rt.make_camera(**STRESS_CAM)
    W, H = 480, 270
    oc = OracleCamera(cam)
    for sample in (0, 5):
        g = scene.trace_primary(cam, rt.render_params(W, H, 1, 50, RENDER_SEED), sample, wavefront=wavefront)
        o = oracle_trace_primary(ob, oc, W, H, sample, RENDER_SEED)
        agree, rel = primary_agreement(g, o)
        assert agree >= ID_AGREEMENT and rel <= T_REL_TOL, (agree, rel)
        assert (o["id"] >= 0).mean() > 0.5 and len(np.unique(o["id"])) > 20000       # the view really is a field of spheres
    scene.close()


def test_one_million_spheres_render_matches_oracle(gpu_required, stress_world):
    """A small render of the stress scene on a shared Philox stream.  This scene is chaotic -- a path bounces between spheres of
    radius 0.2 to 2 that are hundreds of units apart, so the 1e-4 difference between an fp32 and an f64 hit point is amplified
    about a hundredfold per bounce -- and only the paths that leave for the sky within a bounce or two stay identical (86 % of
    them in the CPU build of the same fp32 code).  The checks are therefore statistical: image inside the RMSE noise bound,
    equal image means, ray counts within 1 %, and a floor on the identical single-sample pixels."""
    b, w, ob = stress_world
    scene = rt.Scene(b, w, 0)
    cam = rt.make_camera(**STRESS_CAM)
    oc = OracleCamera(cam)
    W, H, spp = 96, 54, 16
    p = rt.render_params(W, H, spp, 50, RENDER_SEED, background=SKY)
    ga, gsq, st = scene.render(cam, p, want_sumsq=True)
    oa, osq, ost = oracle_render(ob, oc, W, H, spp, 50, RENDER_SEED, background=SKY, want_sumsq=True)
    gm, om = ga.astype(np.float64) / spp, oa / spp
    rmse = np.sqrt(((gm - om) ** 2).mean(axis=(0, 1)))
    assert np.all(rmse <= noise_bound(gm, gsq.astype(np.float64), spp, om, osq, spp)), rmse
    assert np.allclose(gm.mean(axis=(0, 1)), om.mean(axis=(0, 1)), rtol=5e-3)
    assert abs(int(st.rays) - ost["rays"]) <= 1e-2 * ost["rays"], (st.rays, ost["rays"])
    g1, _, _ = scene.render(cam, rt.render_params(W, H, 1, 50, RENDER_SEED, background=SKY))
    o1, _, _ = oracle_render(ob, oc, W, H, 1, 50, RENDER_SEED, background=SKY)
    same = np.mean(np.all(np.abs(g1 - o1) <= 1e-3 * (1 + np.abs(o1)), axis=2))
    assert same >= 0.80, same
    scene.close()


def test_one_million_spheres_at_3840x2160(gpu_required, stress_world):
    """Size-independent properties at the full image size of the stress configuration: determinism, additivity over sample
    ranges, independence from the pass size, ray conservation; and the two trees give the same image."""
    b, w, _ = stress_world
    scene = rt.Scene(b, w, 0)
    cam = rt.make_camera(**STRESS_CAM)
    W, H, D, spp = 3840, 2160, 50, 2
    full, _, st = scene.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED, background=SKY))
    again, _, st2 = scene.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED, background=SKY))
    assert np.array_equal(full, again) and st.rays == st2.rays
    rpb = np.array(list(st.rays_per_bounce[:D]), dtype=np.int64)
    assert rpb[0] == W * H * spp == st.paths and np.all(np.diff(rpb) <= 0) and rpb.sum() == st.rays <= st.paths * D
    assert np.isfinite(full).all() and full.min() >= 0 and 0.05 < full.mean() / spp < 1.0 + 1e-6     # nothing emits but the sky
    parts = np.zeros_like(full)
    _, _, sa = scene.render(cam, rt.render_params(W, H, 1, D, RENDER_SEED, sample_offset=0, background=SKY), accum=parts)
    _, _, sb = scene.render(cam, rt.render_params(W, H, 1, D, RENDER_SEED, sample_offset=1, background=SKY), accum=parts)
    assert sa.rays + sb.rays == st.rays and np.allclose(parts, full, rtol=2e-6, atol=1e-5)
    small, _, ss = scene.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED, background=SKY, paths_per_pass=W * H))
    assert ss.rays == st.rays and ss.passes == 2 and np.allclose(small, full, rtol=2e-6, atol=1e-5)
    scene.close()
    lin = rt.Scene(b, w, 0, rt.RT_BVH_LINEAR)
    other, _, sl = lin.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED, background=SKY))
    # Closest hits do not depend on the tree except on exact ties: two overlapping spheres hit at the same fp32 t, where the
    # first one found wins (the reference's outcome depends on its random tree there too, src/bvh.rs:159-187 with
    # src/objects.rs:218-220).  12 % of this scene's spheres overlap another; a handful of rays in 63 M land on such a tie.
    assert abs(int(sl.rays) - int(st.rays)) <= 1e-6 * st.rays
    assert np.mean(np.any(other != full, axis=2)) <= 1e-5
    lin.close()


@pytest.mark.parametrize("name,gpu_spp,cpu_spp", [("cornell.yml", 4096, 1024), ("final_scene.yml", 4096, 1024)])
def test_converged_images_agree_under_independent_seeds(gpu_required, name, gpu_spp, cpu_spp):
    """The GPU at 4096 spp with one seed against the oracle at 1024 spp with another: nothing is shared between the two
    estimates, so a bias of the fp32 path (self-hit exclusion, widened medium re-query, NaN rectangles) cannot hide behind
    path-by-path agreement.  Per channel: RMSE <= 1.5 sqrt(var_gpu / N_gpu + var_cpu / N_cpu), and the image-wide mean
    difference within 4 standard errors."""
    cfg, cam, b, world, ob = scene_pair(name)
    scene = rt.Scene(b, world, 0)
    W = H = 64
    ga, gsq, _ = scene.render(cam, rt.render_params(W, H, gpu_spp, cfg.max_depth, 0xA11CE), want_sumsq=True)
    oa, osq, _ = oracle_render(ob, OracleCamera(cam), W, H, cpu_spp, cfg.max_depth, 0xB0B, n_threads=16, want_sumsq=True)
    gm, om = ga.astype(np.float64) / gpu_spp, oa / cpu_spp
    var_g = np.maximum(gsq.astype(np.float64) / gpu_spp - gm ** 2, 0)
    var_o = np.maximum(osq / cpu_spp - om ** 2, 0)
    per_pixel = var_g / gpu_spp + var_o / cpu_spp                       # variance of the difference of the two pixel means
    rmse = np.sqrt(((gm - om) ** 2).mean(axis=(0, 1)))
    bound = 1.5 * np.sqrt(per_pixel.mean(axis=(0, 1)))
    assert np.all(rmse <= bound), (rmse, bound)
    mean_diff = (gm - om).mean(axis=(0, 1))
    stderr = np.sqrt(per_pixel.sum(axis=(0, 1))) / (W * H)
    assert np.all(np.abs(mean_diff) <= 4 * stderr), (mean_diff, stderr)
    print("%s: rmse %s <= %s; mean difference %s (standard error %s)" % (name, rmse, bound, mean_diff, stderr))
    scene.close()


def test_queue_overflow_is_reported_not_rendered(gpu_required):
    """A region capacity too small for the pass (forced through the test hook) must end in RT_ERR_INTERNAL, never in an image
    with rays silently dropped."""
    cfg, cam, b, world, _ = scene_pair("cornell.yml")
    p = rt.render_params(128, 128, 4, cfg.max_depth, RENDER_SEED)
    os.environ["RT_DEBUG_QUEUE_CAP"] = "1024"                       # 128 * 128 * 4 / 16 regions = 4096 rays per region
    try:
        scene = rt.Scene(b, world, 0)
        with pytest.raises(rt.RtError) as e:
            scene.render(cam, p)
        assert e.value.code == -9 and "overflow" in str(e.value)
        scene.close()
    finally:
        os.environ.pop("RT_DEBUG_QUEUE_CAP", None)
    scene = rt.Scene(b, world, 0)
    img, _, st = scene.render(cam, p)                               # the same call with the real capacity
    assert st.rays > 128 * 128 * 4 and img.mean() > 0
    scene.close()


def test_render_multi_on_one_gpu_equals_render(gpu_required):
    """rt_render_multi with one scene: host thread, private accumulator and the host add must give rt_render's image bit for bit,
    without RCCL (one GPU has nothing to reduce) and, forced through it, with a one-rank communicator; rt_multi_shutdown destroys
    the communicator and the next call makes a new one; the same handle twice is refused."""
    cfg, cam, b, world, _ = scene_pair("final_scene.yml")
    scene = rt.Scene(b, world, 0)
    p = rt.render_params(160, 120, 6, cfg.max_depth, RENDER_SEED, sample_offset=3)
    want, want_sq, st = scene.render(cam, p, want_sumsq=True)
    got, got_sq, st2 = rt.render_multi([scene], cam, p, want_sumsq=True)
    assert st2.rays == st.rays and st2.paths == st.paths
    assert np.array_equal(got, want) and np.array_equal(got_sq, want_sq)
    assert (st2.multi_gpus, st2.multi_comm_ranks, st2.multi_tiles) == (1, 0, 0)
    os.environ["RT_MULTI_FORCE_RCCL"] = "1"
    try:
        for _ in range(2):                                            # the second round after a shutdown: a fresh communicator
            got, got_sq, st3 = rt.render_multi([scene], cam, p, want_sumsq=True)
            assert np.array_equal(got, want) and np.array_equal(got_sq, want_sq) and (st3.multi_gpus, st3.multi_comm_ranks) == (1, 1)
            rt.multi_shutdown()
    finally:
        os.environ.pop("RT_MULTI_FORCE_RCCL", None)
    rt.multi_shutdown()                                               # nothing left to destroy: still fine
    again, _, _ = rt.render_multi([scene], cam, p, accum=got.copy())  # adds into the caller's buffer; the communicator is reused
    assert np.array_equal(again, 2 * want)
    with pytest.raises(rt.RtError) as e:
        rt.render_multi([scene, scene], cam, p)                       # the same handle twice
    assert e.value.code == -1
    scene.close()


def test_render_multi_with_three_handles_on_one_gpu(gpu_required):
    """Everything of rt_render_multi that one GPU can exercise with more than one handle: a host thread per handle, contiguous
    sample ranges with absolute Philox sample indices, private accumulators, the on-device add of handles that share a GPU and
    the RCCL reduce of what is left.  7 samples over 3 handles = 3 + 2 + 2; the image equals the one-call render up to the order
    of the three partial sums, the ray counts exactly."""
    cfg, cam, b, world, _ = scene_pair("cornell.yml")
    scenes = [rt.Scene(b, world, 0) for _ in range(3)]
    p = rt.render_params(120, 96, 7, cfg.max_depth, RENDER_SEED, sample_offset=11)
    want, want_sq, st = scenes[0].render(cam, p, want_sumsq=True)
    got, got_sq, st2 = rt.render_multi(scenes, cam, p, want_sumsq=True)
    assert st2.rays == st.rays and st2.paths == st.paths and list(st2.rays_per_bounce) == list(st.rays_per_bounce)
    assert np.allclose(got, want, rtol=2e-6, atol=1e-6) and np.allclose(got_sq, want_sq, rtol=2e-6, atol=1e-6)
    parts = [scenes[0].render(cam, rt.render_params(120, 96, c, cfg.max_depth, RENDER_SEED, sample_offset=11 + f))[0] for f, c in ((0, 3), (3, 2), (5, 2))]
    assert np.array_equal(got, (parts[0] + parts[1]) + parts[2])     # the shards, added in handle order
    assert (st2.multi_gpus, st2.multi_tiles) == (1, 0)
    # fewer samples than handles: the pixels are sharded instead (every handle traces both samples of its row bands), nothing is
    # summed across handles, and the image is the one-call render bit for bit; the same when pixel sharding is asked for
    for spp, flags in ((2, 0), (7, rt.RT_FLAG_TILES)):
        q = rt.render_params(120, 96, spp, cfg.max_depth, RENDER_SEED, sample_offset=11, flags=flags)
        want, want_sq, st = scenes[0].render(cam, rt.render_params(120, 96, spp, cfg.max_depth, RENDER_SEED, sample_offset=11), want_sumsq=True)
        got, got_sq, st2 = rt.render_multi(scenes, cam, q, want_sumsq=True)
        assert st2.multi_tiles == 1 and st2.multi_comm_ranks == 0
        assert st2.rays == st.rays and st2.paths == st.paths and np.array_equal(got, want) and np.array_equal(got_sq, want_sq)
    # a shard without samples (its accumulator holds only the memset): 2 samples over 3 handles, forced to shard by samples
    # cannot be asked for, so the zero-sample case is spp = 0
    got, _, st2 = rt.render_multi(scenes, cam, rt.render_params(120, 96, 0, cfg.max_depth, RENDER_SEED))
    assert st2.rays == 0 and np.all(got == 0)
    for s in scenes:
        s.close()


def test_pixel_shards_tile_the_image_bit_for_bit(gpu_required):
    """RtRenderParams.tile / .stripe: a rectangle, the row bands of rt_shard_tiles, and both together render exactly the pixels
    they name with the values a whole-image render gives them, and leave the others alone."""
    cfg, cam, b, world, _ = scene_pair("final_scene.yml")
    scene = rt.Scene(b, world, 0)
    W, H, spp = 150, 101, 3                                            # 101 rows: 13 bands of 8, the last one short
    base = rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED, sample_offset=4)
    whole, whole_sq, st = scene.render(cam, base, want_sumsq=True)
    for n in (3, 5, 16):                                               # 16 > 13 bands: three shards are empty
        total, rays, pixels = np.zeros_like(whole), 0, 0
        for i in range(n):
            q, count = rt.shard_tiles(base, i, n)
            part, _, sp = scene.render(cam, q)
            assert sp.paths == count * spp and np.count_nonzero(part.sum(axis=2)) <= count
            assert not np.any((total != 0) & (part != 0))             # disjoint
            total += part; rays += int(sp.rays); pixels += count
        assert pixels == W * H and rays == st.rays and np.array_equal(total, whole)
    x0, y0, w, h = 17, 9, 64, 50
    rect, rect_sq, sr = scene.render(cam, rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED, sample_offset=4, tile=(x0, y0, w, h)), want_sumsq=True)
    mask = np.zeros((H, W, 1), bool); mask[y0:y0 + h, x0:x0 + w] = True
    assert sr.paths == w * h * spp and np.array_equal(rect, np.where(mask, whole, 0)) and np.array_equal(rect_sq, np.where(mask, whole_sq, 0))
    both = np.zeros_like(whole)
    for i in range(2):                                                 # bands of 5 rows of the rectangle, dealt to two shards
        scene.render(cam, rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED, sample_offset=4, tile=(x0, y0, w, h), stripe=(5, 2, i)), accum=both)
    assert np.array_equal(both, rect)
    # the no-allocation entry point takes the same selection (its limit is on the pixels rendered)
    import torch
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    scene.render_device_noalloc(cam, rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED, sample_offset=4, tile=(x0, y0, w, h)), acc.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), rect)
    for bad in (dict(tile=(0, 0, W + 1, 4)), dict(tile=(-1, 0, 4, 4)), dict(tile=(0, 0, 0, 4)), dict(stripe=(0, 2, 1)), dict(stripe=(8, 2, 2)), dict(stripe=(8, 0, 0))):
        with pytest.raises(rt.RtError) as e:
            scene.render(cam, rt.render_params(W, H, 1, 5, 1, **bad))
        assert e.value.code == -1
    scene.close()


def test_render_sharded_with_the_hip_renderer(gpu_required):
    """The per-process sharding layer (one process per GPU under torch.distributed) driving rt_render_device, at world size 1."""
    import torch
    from ray_tracing_amd.distributed import render_sharded
    cfg, cam, b, world, _ = scene_pair("cornell.yml")
    scene = rt.Scene(b, world, 0)
    W = H = 120
    spp = 5

    def render_into(accum, first, count):
        scene.render_device(cam, rt.render_params(W, H, count, cfg.max_depth, RENDER_SEED, sample_offset=first), accum.data_ptr(), None,
                            torch.cuda.current_stream().cuda_stream)
    got = render_sharded(render_into, H, W, spp, torch.device("cuda", 0)).cpu().numpy()
    want, _, _ = scene.render(cam, rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED))
    assert np.array_equal(got, want)
    scene.close()


@pytest.mark.parametrize("flags", [0, rt.RT_BVH_LINEAR], ids=["sah_tree", "device_built_linear_tree"])
def test_refit_on_the_gpu_gives_the_hits_of_a_rebuild(gpu_required, flags):
    """rt_scene_set_spheres / rt_scene_set_moving_spheres / rt_scene_refit: the boxes of a resident scene follow its moved
    primitives (free spheres, spheres of an instanced subtree, a moving sphere); closest hits and the rendered image must be
    those of a scene created from the moved primitives, through the binary tree and through the 4-wide tree."""
    from test_flatten import moving_scene
    rng = np.random.default_rng(5)
    n, m = 3000, 200
    pos0 = rng.uniform([-15, -1, -30], [15, 8, -8], (n, 3)); rad = rng.uniform(0.1, 0.5, n)
    cl0 = rng.uniform(0, 6, (m, 3))
    ms0 = np.array([4, 5, -10, 6, 6, -10, 0.0, 1.0, 0.8])
    pos1 = pos0 + rng.normal(0, 1.0, (n, 3)); rad1 = rad * rng.uniform(0.7, 1.3, n)
    cl1 = cl0 + rng.normal(0, 0.5, (m, 3))
    ms1 = np.array([-6, 4, -12, -3, 7, -11, 0.0, 1.0, 1.1])
    cam = rt.make_camera((0, 6, 12), (0, 2, -18), (0, 1, 0), 45.0, 1.5, 0.05, 20.0)
    W, H = 300, 200
    p = rt.render_params(W, H, 4, 50, RENDER_SEED, background=SKY)
    oracle_hits = None
    for wide in ("0", "1"):
        os.environ["RT_TRACE_BVH4"] = wide
        try:
            b = rt.Builder()
            world, ids, cids, mid = moving_scene(b, pos0, rad, cl0, ms0)
            scene = rt.Scene(b, world, 0, flags)
            before, _, _ = scene.render(cam, p)
            scene.set_spheres(ids, np.concatenate([pos1, rad1[:, None]], axis=1))
            scene.set_spheres(cids, np.concatenate([cl1, np.full((m, 1), 0.4)], axis=1))
            scene.set_moving_spheres([mid], ms1[None, :])
            scene.refit()
            assert scene.info().bvh_refit_seconds > 0
            b2 = rt.Builder()
            w2, _, _, _ = moving_scene(b2, pos1, rad1, cl1, ms1)
            fresh = rt.Scene(b2, w2, 0, flags)
            got, want = scene.trace_primary(cam, p, 0), fresh.trace_primary(cam, p, 0)
            assert np.array_equal(got["id"], want["id"]) and np.array_equal(got["t"], want["t"])
            # ... and those of the oracle built at the moved positions, through the kernels a render runs on the refitted trees
            via = scene.trace_primary(cam, p, 0, wavefront=True)
            assert np.array_equal(via["id"], got["id"]) and np.array_equal(via["t"], got["t"])
            if oracle_hits is None:
                ob = OracleBuilder()
                ow, _, _, _ = moving_scene(ob, pos1, rad1, cl1, ms1)
                ob.set_world(ow)
                oracle_hits = oracle_trace_primary(ob, OracleCamera(cam), W, H, 0, RENDER_SEED)
                assert (oracle_hits["id"] >= 0).mean() > 0.5
            agree, rel = primary_agreement(via, oracle_hits)
            assert agree >= ID_AGREEMENT and rel <= T_REL_TOL, (wide, agree, rel)
            img, _, st = scene.render(cam, p)
            ref, _, st2 = fresh.render(cam, p)
            assert st.rays == st2.rays and np.array_equal(img, ref) and not np.array_equal(img, before)
            scene.close(); fresh.close()
        finally:
            os.environ.pop("RT_TRACE_BVH4", None)


def test_refit_of_the_stress_scene(gpu_required, stress_world):
    """All 1 000 000 spheres jittered and refitted on the device-built tree: primary hits against the oracle built at the new
    positions would need its 8-second build, so the check is against a second scene created from the moved spheres."""
    b, w, _ = stress_world
    scene = rt.Scene(b, w, 0, rt.RT_BVH_LINEAR)
    # the spheres of Synthetic:<n> are objects 1 .. n (0 is the ground); read their records back through a host flatten
    from helpers import FlatScene
    from test_flatten import arrays
    flat = FlatScene(b, w)
    a = arrays(flat)
    ids = a["sphere_meta"][:, 1]
    keep = ids > 0
    rng = np.random.default_rng(9)
    xyzr = a["spheres"][keep].astype(np.float64)
    xyzr[:, :3] += rng.normal(0, 0.5, (xyzr.shape[0], 3))
    scene.set_spheres(ids[keep], xyzr)
    scene.refit()
    assert 0 < scene.info().bvh_refit_seconds < 0.1
    b2 = rt.Builder()
    mat = b2.lambertian((0.5, 0.5, 0.5))
    order = np.argsort(ids[keep])
    kids = [b2.sphere((0, -100000, 0), 100000, mat)] + [b2.sphere(tuple(r[:3]), r[3], mat) for r in xyzr[order]]
    fresh = rt.Scene(b2, b2.list(kids), 0)
    cam = rt.make_camera(**STRESS_CAM)
    p = rt.render_params(480, 270, 1, 50, RENDER_SEED)
    got, want = scene.trace_primary(cam, p, 0), fresh.trace_primary(cam, p, 0)
    same = got["id"] == want["id"]
    assert same.mean() >= ID_AGREEMENT and np.array_equal(got["t"][same], want["t"][same])       # ties between overlapping spheres aside
    print("refit of 1M spheres: %.2f ms on the device" % (scene.info().bvh_refit_seconds * 1e3))
    scene.close(); fresh.close()


def test_scenes_on_one_gpu_share_one_workspace(gpu_required):
    """The queues of a pass belong to the GPU: a second scene handle must not allocate them again, two host threads rendering two
    scenes on one GPU take turns and each gets its own image, and rt_release_workspace gives the memory back."""
    import threading
    import torch
    cfg_a, cam_a, ba, wa, _ = scene_pair("final_scene.yml")
    cfg_b, cam_b, bb, wb, _ = scene_pair("cornell.yml")
    a, b = rt.Scene(ba, wa, 0), rt.Scene(bb, wb, 0)
    pa = rt.render_params(400, 400, 8, cfg_a.max_depth, RENDER_SEED)
    pb = rt.render_params(400, 400, 8, cfg_b.max_depth, RENDER_SEED)
    want_a, _, _ = a.render(cam_a, pa)
    free_after_first = torch.cuda.mem_get_info(0)[0]
    want_b, _, _ = b.render(cam_b, pb)
    assert free_after_first - torch.cuda.mem_get_info(0)[0] < 64 << 20          # same pass size: nothing new but small buffers
    got = {}

    def run(key, scene, cam, p):
        for _ in range(3):
            got[key], _, _ = scene.render(cam, p)
    ts = [threading.Thread(target=run, args=("a", a, cam_a, pa)), threading.Thread(target=run, args=("b", b, cam_b, pb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert np.array_equal(got["a"], want_a) and np.array_equal(got["b"], want_b)
    before = torch.cuda.mem_get_info(0)[0]
    rt.release_workspace(0)
    assert torch.cuda.mem_get_info(0)[0] - before > 100 << 20                    # 400 * 400 * 8 paths * 128 B and the region slack
    again, _, _ = a.render(cam_a, pa)                                            # allocated again on demand
    assert np.array_equal(again, want_a)
    # the last handle on the GPU takes the workspace with it: a caller that closes its scenes gets the memory back
    a.close()
    held = torch.cuda.mem_get_info(0)[0]
    b.close()
    assert torch.cuda.mem_get_info(0)[0] - held > 100 << 20


def _hip_rank(rank, world_size, port, out_path):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from ray_tracing_amd.distributed import render_sharded
        cfg, cam, b, world, _ = scene_pair("cornell_box.yml")
        scene = rt.Scene(b, world, 0)

        def render_into(accum, first, count):
            scene.render_device(cam, rt.render_params(96, 96, count, cfg.max_depth, RENDER_SEED, sample_offset=first), accum.data_ptr(), None,
                                torch.cuda.current_stream().cuda_stream)
        accum = render_sharded(render_into, 96, 96, 7, torch.device("cuda", 0))
        if rank == 0:
            np.save(out_path, accum.cpu().numpy())
        scene.close()
    finally:
        dist.destroy_process_group()


def test_three_ranks_shard_the_hip_renderer(gpu_required, tmp_path):
    """One process per rank, as bench.py runs under torch.distributed.run, with the HIP renderer in every rank.  The box has one
    GPU, so the three ranks share it and reduce over gloo (through the host); sharding, absolute sample indices and the reduce to
    rank 0 are the code the N-GPU run uses.  7 samples = 3 + 2 + 2."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "accum.npy")
    mp.spawn(_hip_rank, args=(3, port, out), nprocs=3, join=True)
    got = np.load(out)
    cfg, cam, b, world, _ = scene_pair("cornell_box.yml")
    scene = rt.Scene(b, world, 0)
    want, _, _ = scene.render(cam, rt.render_params(96, 96, 7, cfg.max_depth, RENDER_SEED))
    assert want.sum() > 0 and np.allclose(got, want, rtol=2e-6, atol=1e-6)
    scene.close()


def _walk_scene(b, extra_inside=True, poke=True, move=None):
    """A glass ball with a dense medium inside it (as src/main.rs:392-405 builds one), a small metal sphere wholly inside the
    ball, a box that pokes into it, a light and a floor: rays that scatter inside the ball can only meet the ball's own surface,
    the small sphere or the box before they leave it.  Returns (world, id of a free sphere outside the ball)."""
    glass = b.dielectric(1.5)
    kids = [b.sphere((0, 0, -10), 3.0, glass)]
    kids.append(b.constant_medium_from_color(b.sphere((0, 0, -10), 3.0, glass), 0.9, (0.3, 0.5, 0.9)))
    if extra_inside:
        kids.append(b.sphere((0.6, 0.4, -10.2), 0.7, b.metal((0.9, 0.8, 0.7), 0.1)))
    if poke:
        kids.append(b.box((-1.0, -6.0, -11.0), (1.0, -2.2, -9.0), b.lambertian((0.7, 0.3, 0.3))))
    free_id = b.object_count()
    kids.append(b.sphere(move if move is not None else (7.0, 0.0, -10.0), 1.0, b.lambertian((0.2, 0.7, 0.3))))
    kids.append(b.rectangle(b.diffuse_light_from_color((6, 6, 6)), -4, 4, -14, -6, 8.0, rt.RECT_XZ))
    kids.append(b.rectangle(b.lambertian((0.6, 0.6, 0.6)), -30, 30, -40, 10, -6.0, rt.RECT_XZ))
    return b.list(kids), free_id


def test_walks_inside_a_medium_skip_the_tree_search_exactly(gpu_required):
    """Rays that leave a collision inside the ball are settled by the shading kernel against the surfaces listed in the medium
    record (three here: the ball, the sphere inside it, the box that pokes in) and never reach the closest-hit kernel.  The image
    and the ray counts must be those of the render that sends every ray through the tree, and of the oracle; and a sphere moved
    into the ball empties the list (the scene then equals one built from the moved spheres)."""
    cam = rt.make_camera((0, 2, 6), (0, 0, -10), (0, 1, 0), 35.0, 1.5, 0.05, 16.0)
    W, H, spp = 240, 160, 8
    p = rt.render_params(W, H, spp, 50, RENDER_SEED, background=(0.05, 0.06, 0.08))

    def render(local, **kw):
        b = rt.Builder()
        world, free_id = _walk_scene(b, **kw)
        os.environ["RT_LOCAL_MEDIA"] = local
        os.environ["RT_DEBUG_WORK"] = "1"
        try:
            scene = rt.Scene(b, world, 0)
            img, _, st = scene.render(cam, p)
            _, _, counted = scene.render(cam, rt.render_params(W, H, 1, 50, RENDER_SEED, flags=rt.RT_FLAG_COUNTERS))     # the instrumented kernels take the same route
        finally:
            os.environ.pop("RT_LOCAL_MEDIA", None); os.environ.pop("RT_DEBUG_WORK", None)
        return scene, img, st, free_id
    for kw in (dict(), dict(extra_inside=False), dict(poke=False), dict(extra_inside=False, poke=False)):
        s1, on, st_on, _ = render("1", **kw)
        s0, off, st_off, _ = render("0", **kw)
        assert st_on.rays == st_off.rays and list(st_on.rays_per_bounce) == list(st_off.rays_per_bounce) and np.array_equal(on, off), kw
        assert st_on.rays > 3 * W * H * spp                              # the walks are there: paths are long
        os.environ["RT_LOCAL_SURFACES"] = "1"                            # also settle rays that start on a surface wholly inside the ball
        try:
            s2, both, st_both, _ = render("1", **kw)
        finally:
            os.environ.pop("RT_LOCAL_SURFACES", None)
        assert st_both.rays == st_off.rays and np.array_equal(both, off), kw
        s0.close(); s1.close(); s2.close()
    ob = OracleBuilder()
    ow, _ = _walk_scene(ob)
    ob.set_world(ow)
    oa, _, ost = oracle_render(ob, OracleCamera(cam), W, H, spp, 50, RENDER_SEED, background=(0.05, 0.06, 0.08))
    scene, img, st, free_id = render("1")
    same = np.mean(np.all(np.abs(img - oa) <= 1e-3 * (spp + np.abs(oa)), axis=2))
    assert same >= 0.97 and abs(int(st.rays) - ost["rays"]) <= 3e-3 * ost["rays"], (same, st.rays, ost["rays"])
    # a fourth surface inside the ball: the list no longer describes it, and a moved primitive must empty it
    scene.set_spheres([free_id], [[-0.8, -0.5, -9.5, 1.0]])
    scene.refit()
    moved, _, st_m = scene.render(cam, p)
    b2 = rt.Builder()
    w2, _ = _walk_scene(b2, move=(-0.8, -0.5, -9.5))
    fresh = rt.Scene(b2, w2, 0)
    want, _, st_w = fresh.render(cam, p)
    got_fix = fresh.trace_primary(cam, p, 0)
    assert st_m.rays == st_w.rays and np.array_equal(moved, want) and not np.array_equal(moved, img)
    assert (got_fix["id"] >= 0).any()
    scene.close(); fresh.close()


@pytest.mark.parametrize("name", ["final_scene.yml", "cornell.yml"])
def test_settled_rays_under_every_call_shape(gpu_required, name):
    """The two-ended queue regions (rays the shading kernel settles go to the back part) against the one-ended form, RT_LOCAL_MEDIA=0, over
    the shapes a call can take: depth 1 and 2 (no settled ray ever gets shaded / exactly one generation does), several passes per call,
    passes of a few hundred paths (regions of a handful of slots, back parts included), a pixel rectangle, row bands of one shard, the
    sum-of-squares output.  Same image, same counts, every time.  (cornell.yml's media have instanced boundaries and list no surfaces:
    nothing is settled there, and the switch must change nothing.)"""
    cfg, cam = config_camera(name)
    b = rt.Builder()
    scene = rt.Scene(b, b.scene_from_config(os.path.join(CONFIGS, name), SCENE_SEED), 0)
    W, H = 96, 80
    shapes = [dict(spp=3, depth=1), dict(spp=3, depth=2), dict(spp=2, depth=50), dict(spp=5, depth=50, ppp=W * H * 2), dict(spp=2, depth=50, ppp=700),
              dict(spp=3, depth=50, tile=(8, 5, 61, 50)), dict(spp=3, depth=17, stripe=(4, 3, 1)), dict(spp=2, depth=50, tile=(3, 3, 40, 40), stripe=(8, 2, 0), ppp=3000),
              dict(spp=2, depth=50, sumsq=True)]
    settled_any = 0
    for sh in shapes:
        p = rt.render_params(W, H, sh["spp"], sh["depth"], RENDER_SEED, sample_offset=5, paths_per_pass=sh.get("ppp", 0), tile=sh.get("tile"), stripe=sh.get("stripe"))
        res = {}
        for local in ("1", "0"):
            os.environ["RT_LOCAL_MEDIA"] = local
            try:
                img, sq, st = scene.render(cam, p, want_sumsq=bool(sh.get("sumsq")))
            finally:
                os.environ.pop("RT_LOCAL_MEDIA", None)
            res[local] = (img, sq, st)
        (i1, q1, s1), (i0, q0, s0) = res["1"], res["0"]
        assert np.array_equal(i1, i0) and s1.rays == s0.rays and list(s1.rays_per_bounce) == list(s0.rays_per_bounce) and s1.passes == s0.passes, sh
        assert s0.rays_settled == 0 and i1.sum() > 0, sh
        if sh.get("sumsq"):
            assert np.array_equal(q1, q0)
        settled_any += int(s1.rays_settled)
    assert (settled_any > 0) == (name == "final_scene.yml")
    scene.close()
