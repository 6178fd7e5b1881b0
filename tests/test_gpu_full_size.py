"""Size-independent properties at the image sizes of the headline configs (cornell.yml 600x600, final_scene.yml
800x800, a 200k-sphere synthetic scene): determinism, sample-range additivity (what multi-GPU sharding relies on),
independence from the pass size, ray accounting, and the device-buffer entry point."""
import os

import numpy as np
import pytest

import ray_tracing_amd as rt
from helpers import CONFIGS, RENDER_SEED, SCENE_SEED, OracleBuilder, OracleCamera, config_camera, oracle_trace_primary, primary_agreement

pytestmark = pytest.mark.gpu


def load(name):
    cfg, cam = config_camera(name)
    b = rt.Builder()
    return cfg, cam, rt.Scene(b, b.scene_from_config(os.path.join(CONFIGS, name), SCENE_SEED), 0)


@pytest.mark.parametrize("name,spp", [("cornell.yml", 8), ("final_scene.yml", 6)])
def test_full_size_properties(gpu_required, name, spp):
    cfg, cam, scene = load(name)
    W, H, D = cfg.image_width, cfg.image_height, cfg.max_depth
    full, sq, st = scene.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED), want_sumsq=True)
    again, _, st2 = scene.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED))
    assert np.array_equal(full, again) and st.rays == st2.rays                        # deterministic, bit for bit
    # ray accounting: every path casts a camera ray; bounces only lose rays; at most max_depth segments per path (lib.rs:58-60)
    rpb = np.array(list(st.rays_per_bounce[:D]), dtype=np.int64)
    assert rpb[0] == W * H * spp == st.paths and np.all(np.diff(rpb) <= 0) and rpb.sum() == st.rays <= st.paths * D
    assert np.isfinite(full).all() and full.min() >= 0 and 0.05 < full.mean() / spp < 5.0
    assert np.all(sq >= 0) and np.all(sq * spp >= full * full * (1 - 1e-4) - 1e-3)     # Cauchy-Schwarz per pixel
    # sample ranges add up: [0, k) + [k, spp) is the same set of paths as [0, spp)  -> multi-GPU sample sharding
    k = spp // 2
    parts = np.zeros_like(full)
    _, _, sa = scene.render(cam, rt.render_params(W, H, k, D, RENDER_SEED, sample_offset=0), accum=parts)
    _, _, sb = scene.render(cam, rt.render_params(W, H, spp - k, D, RENDER_SEED, sample_offset=k), accum=parts)
    assert sa.rays + sb.rays == st.rays
    assert np.allclose(parts, full, rtol=2e-6, atol=1e-5)
    # pass size only changes how the work is cut, not which paths are traced
    small, _, ss = scene.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED, paths_per_pass=W * H * 2))
    assert ss.rays == st.rays and ss.passes > st.passes and np.allclose(small, full, rtol=2e-6, atol=1e-5)
    # a different seed is a different image with the same statistics
    other, _, _ = scene.render(cam, rt.render_params(W, H, spp, D, RENDER_SEED + 1))
    assert not np.array_equal(other, full) and abs(other.mean() - full.mean()) < 0.05 * full.mean()
    scene.close()


def test_device_buffer_entry_point_matches_host_one(gpu_required):
    import torch
    cfg, cam, scene = load("cornell.yml")
    W, H = 200, 200
    p = rt.render_params(W, H, 4, cfg.max_depth, RENDER_SEED, flags=rt.RT_FLAG_SUMSQ)
    host, host_sq, st = scene.render(cam, rt.render_params(W, H, 4, cfg.max_depth, RENDER_SEED), want_sumsq=True)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    sq = torch.zeros_like(acc)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        st2 = scene.render_device(cam, p, acc.data_ptr(), sq.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    assert st2.rays == st.rays and np.array_equal(acc.cpu().numpy(), host) and np.array_equal(sq.cpu().numpy(), host_sq)
    st3 = scene.render_device(cam, p, acc.data_ptr(), sq.data_ptr(), None)           # adds into the same buffers
    assert np.allclose(acc.cpu().numpy(), 2 * host, rtol=1e-6)
    scene.close()


def test_deep_bvh_synthetic_spheres(gpu_required):
    """200k random spheres (the shape of the 1M-sphere stress config, scaled to build quickly from Python): primary hits against the oracle."""
    rng = np.random.default_rng(4)
    n = 200000
    cx, cz, cy, rr = rng.uniform(-300, 300, n), rng.uniform(-300, 300, n), rng.uniform(0, 60, n), rng.uniform(0.2, 2.0, n)
    b, ob = rt.Builder(), OracleBuilder()
    worlds = []
    for bb in (b, ob):
        mats = [bb.lambertian((0.5, 0.5, 0.5)), bb.metal((0.8, 0.8, 0.8), 0.1), bb.dielectric(1.5)]
        kids = [bb.sphere((0, -1e5, 0), 1e5, mats[0])] + [bb.sphere((cx[i], cy[i], cz[i]), rr[i], mats[0 if i % 20 < 16 else (1 if i % 20 < 19 else 2)]) for i in range(n)]
        worlds.append(bb.list(kids))
    ob.set_world(worlds[1])
    scene = rt.Scene(b, worlds[0], 0)
    info = scene.info()
    assert info.n_spheres == n + 1 and info.max_stack <= 32
    cam = rt.make_camera((0, 120, 420), (0, 20, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
    W, H = 480, 270
    p = rt.render_params(W, H, 1, 50, RENDER_SEED)
    g = scene.trace_primary(cam, p, 0)
    o = oracle_trace_primary(ob, OracleCamera(cam), W, H, 0, RENDER_SEED)
    agree, rel = primary_agreement(g, o)
    assert agree >= 0.9999 and rel <= 1e-4, (agree, rel)
    accum, _, st = scene.render(cam, rt.render_params(W, H, 4, 50, RENDER_SEED, background=(0.7, 0.8, 1.0)))
    assert np.isfinite(accum).all() and st.rays > W * H * 4
    scene.close()


def _synthetic_pair(n):
    b, ob = rt.Builder(), OracleBuilder()
    w = b.canned_scene("Synthetic:%d" % n, SCENE_SEED)
    ob.set_world(ob.canned_scene("Synthetic:%d" % n, SCENE_SEED))
    return b, w, ob


def test_synthetic_sphere_field_matches_oracle(gpu_required):
    """The deep-BVH stress config at 60k spheres: built by the library's and by the oracle's own builder from the same seed."""
    b, w, ob = _synthetic_pair(60000)
    scene = rt.Scene(b, w, 0)
    assert scene.info().n_spheres == 60001
    cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
    W, H = 480, 270
    g = scene.trace_primary(cam, rt.render_params(W, H, 1, 50, RENDER_SEED), 0)
    o = oracle_trace_primary(ob, OracleCamera(cam), W, H, 0, RENDER_SEED)
    agree, rel = primary_agreement(g, o)
    assert agree >= 0.9999 and rel <= 1e-4, (agree, rel)
    scene.close()


@pytest.mark.parametrize("env", [{"RT_TRACE_WIDE_STACK": "1"}, {"RT_TRACE_BLOCK": "256"}, {"RT_SHADE_LDS": "0"}, {"RT_SHADE_FUSED": "0"},
                                 {"RT_TRACE_BVH4": "1", "RT_SHADE_FUSED": "0"}, {"RT_SHADE_FLAT": "0"}, {"RT_LOCAL_MEDIA": "0"}, {"RT_LOCAL_SURFACES": "1"},
                                 {"RT_TRACE_STACK_LDS": "6"}, {"RT_TRACE_STACK_LDS": "3", "RT_TRACE_BVH4": "1", "RT_SHADE_FUSED": "0"}])
@pytest.mark.parametrize("name", ["cornell.yml", "final_scene.yml", "showcase.yml", "rectangle.yml", "orig_img.yml", "cornell_box.yml", "two_spheres.yml"])
def test_kernel_variants_agree(gpu_required, name, env):
    """k_trace picks its block shape, stack form, node format (binary or 4-wide quantised) and what lives in LDS per scene, and
    shading runs either from tables in LDS with staged ray records (k_shade_lds) or from global memory (k_shade); every choice
    must give the very same image."""
    cfg, cam = config_camera(name)
    p = rt.render_params(200, 200, 4, cfg.max_depth, RENDER_SEED)

    def render():
        b = rt.Builder()
        scene = rt.Scene(b, b.scene_from_config(os.path.join(CONFIGS, name), SCENE_SEED), 0)
        img, _, st = scene.render(cam, p)
        scene.close()
        return img, (st.rays, tuple(st.rays_per_bounce))
    want, rays = render()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        got, rays2 = render()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert rays2 == rays and np.array_equal(got, want)


@pytest.mark.parametrize("size", [(37, 29), (131, 7), (2, 2), (260, 251)])
def test_shading_kernels_agree_on_ragged_sizes(gpu_required, size):
    """k_shade_lds stages 64 queue entries per wave and clamps its last batch to the queue end; k_shade reads entry by entry.
    Queue lengths that are a multiple of nothing, single-ray queues and passes of a few hundred paths must give the same image."""
    W, H = size
    cfg, cam = config_camera("final_scene.yml")
    cam = rt.make_camera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 10.0)
    b = rt.Builder()
    world = b.scene_from_config(os.path.join(CONFIGS, "final_scene.yml"), SCENE_SEED)
    images = []
    for lds in ("1", "0"):
        old = os.environ.get("RT_SHADE_LDS")
        os.environ["RT_SHADE_LDS"] = lds
        try:
            scene = rt.Scene(b, world, 0)
            one, _, st1 = scene.render(cam, rt.render_params(W, H, 5, cfg.max_depth, RENDER_SEED))
            many, _, st2 = scene.render(cam, rt.render_params(W, H, 5, cfg.max_depth, RENDER_SEED, paths_per_pass=max(1, W * H * 2)))
            scene.close()
        finally:
            if old is None:
                os.environ.pop("RT_SHADE_LDS", None)
            else:
                os.environ["RT_SHADE_LDS"] = old
        assert st2.passes == 3 and st1.rays == st2.rays
        np.testing.assert_allclose(many, one, rtol=1e-5, atol=1e-6)   # pass boundaries only change the fp32 summation order
        images.append((one, many, st1.rays))
    assert images[0][2] == images[1][2]
    assert np.array_equal(images[0][0], images[1][0]) and np.array_equal(images[0][1], images[1][1])


def test_scenes_with_different_lds_plans_coexist(gpu_required):
    """The kernels' dynamic-LDS limit is set once per kernel, not per scene: a scene with a large LDS plan must still render
    after a scene with a small one has been created (and the other way round)."""
    def make(name):
        cfg, cam = config_camera(name)
        b = rt.Builder()
        return rt.Scene(b, b.scene_from_config(os.path.join(CONFIGS, name), SCENE_SEED), 0), cam, rt.render_params(96, 96, 2, cfg.max_depth, RENDER_SEED)
    big, cam_b, p_b = make("final_scene.yml")
    want_b, _, _ = big.render(cam_b, p_b)
    small, cam_s, p_s = make("rectangle.yml")
    want_s, _, _ = small.render(cam_s, p_s)
    got_b, _, _ = big.render(cam_b, p_b)
    got_s, _, _ = small.render(cam_s, p_s)
    assert np.array_equal(got_b, want_b) and np.array_equal(got_s, want_s)
    big.close(); small.close()


def test_cli_checkpoint_and_resume(gpu_required, tmp_path):
    """rt_cli renders in passes with --checkpoint, continues an existing checkpoint, and writes PNG by extension."""
    import subprocess
    import zlib
    cli = os.path.join(os.path.dirname(rt.LIB_PATH), "rt_cli")
    cfg = os.path.join(CONFIGS, "cornell.yml")
    ck, out1, out2 = str(tmp_path / "acc.bin"), str(tmp_path / "a.png"), str(tmp_path / "b.png")

    def run(*args):
        r = subprocess.run([cli, "-c", cfg] + list(args), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        return r.stderr

    run("-o", out1, "--spp", "8", "--checkpoint", ck, "--pass-spp", "3")           # 3 + 3 + 2 samples, saved after each pass
    acc, spp, seed = rt.accum_load(ck, 600, 600)
    assert spp == 8 and seed == 0x5EED0002 and acc.mean() > 0
    log = run("-o", out2, "--spp", "12", "--checkpoint", ck, "--pass-spp", "4")    # continues at sample 8
    assert "resuming" in log and "at 8 of 12" in log
    acc2, spp2, _ = rt.accum_load(ck, 600, 600)
    assert spp2 == 12 and np.all(acc2 >= acc)                                       # samples only add light

    def decode(path):
        data = open(path, "rb").read()
        pos, idat = 8, b""
        while pos < len(data):
            n = int.from_bytes(data[pos:pos + 4], "big")
            if data[pos + 4:pos + 8] == b"IDAT": idat += data[pos + 8:pos + 8 + n]
            pos += 12 + n
        return np.frombuffer(zlib.decompress(idat), np.uint8).reshape(600, 1 + 1800)[:, 1:].reshape(600, 600, 3)
    img = decode(out2)
    _, cam, scene = load("cornell.yml")
    whole, _, _ = scene.render(cam, rt.render_params(600, 600, 12, 50, RENDER_SEED))  # the same 12 samples in one pass
    scene.close()
    want = rt.tonemap_u8(whole, 12)
    assert np.abs(img.astype(int) - want.astype(int)).max() <= 1                    # fp32 summation order only
    # asking for fewer samples than the checkpoint holds renders nothing and divides by what the buffer holds, not by the request
    out3 = str(tmp_path / "c.png")
    run("-o", out3, "--spp", "4", "--checkpoint", ck)
    assert np.array_equal(decode(out3), img)
    # samples of another scene are not mixed into the image
    r = subprocess.run([cli, "-c", cfg, "-o", out3, "--spp", "16", "--checkpoint", ck, "--scene-seed", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "another scene" in r.stderr


@pytest.mark.parametrize("name", ["final_scene.yml", "random.yml", "showcase.yml"])
def test_device_built_linear_bvh(gpu_required, name):
    """RT_BVH_LINEAR: the top level is built on the GPU; it must be the tree the host mirror of the same steps builds, and it
    must give the same closest hits and the same image as the SAH tree (hits do not depend on the tree)."""
    cfg, cam = config_camera(name)
    path = os.path.join(CONFIGS, name)
    b = rt.Builder()
    world = b.scene_from_config(path, SCENE_SEED)
    host = b.flatten_info(world, rt.RT_BVH_LINEAR)
    lin = rt.Scene(b, world, 0, rt.RT_BVH_LINEAR)
    sah = rt.Scene(b, world, 0)
    li, si = lin.info(), sah.info()
    assert (li.n_nodes, li.tlas_depth, li.max_stack) == (host.n_nodes, host.tlas_depth, host.max_stack)
    assert li.bvh_build_seconds > 0 and si.bvh_build_seconds == 0
    W, H = 200, int(round(200 / cfg.aspect_ratio))
    p = rt.render_params(W, H, 4, cfg.max_depth, RENDER_SEED)
    a, c = lin.trace_primary(cam, p, 0), sah.trace_primary(cam, p, 0)
    assert np.array_equal(a["id"], c["id"]) and np.array_equal(a["t"], c["t"])
    img_l, _, st_l = lin.render(cam, p)
    img_s, _, st_s = sah.render(cam, p)
    assert st_l.rays == st_s.rays and np.array_equal(img_l, img_s)
    lin.close(); sah.close()


def test_device_built_linear_bvh_one_million_spheres(gpu_required):
    b = rt.Builder()
    w = b.canned_scene("Synthetic:1000000", SCENE_SEED)
    scene = rt.Scene(b, w, 0, rt.RT_BVH_LINEAR)
    info = scene.info()
    assert info.n_spheres == 1000001 and info.n_nodes == 1000000 + 2048 and info.max_stack <= 32   # + copies of the top levels
    assert 0 < info.bvh_build_seconds < 0.5
    cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
    img, _, st = scene.render(cam, rt.render_params(480, 270, 2, 50, RENDER_SEED, background=(0.7, 0.8, 1.0)))
    assert np.isfinite(img).all() and st.rays > 480 * 270 * 2
    scene.close()
