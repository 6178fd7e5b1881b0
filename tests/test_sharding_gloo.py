"""The N>1 path on CPU: two gloo ranks shard the samples of one image, reduce to rank 0, and the result must be
the single-process image.  The oracle stands in for the renderer here (test infrastructure); on GPUs the same
shard/reduce code drives rt_render_device over RCCL (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ray_tracing_amd.distributed import shard_samples

W, H, SPP, DEPTH = 24, 20, 6, 8


def test_shard_samples_partitions_the_range():
    for total in (0, 1, 5, 16, 1000, 100000):
        for world in (1, 2, 3, 4, 8):
            parts = [shard_samples(total, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == total
            assert parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_samples(10, 2, 2)


def test_c_abi_shards_like_the_python_layer():
    """rt_render_multi (GPUs inside one process) and render_sharded (one process per GPU) must cut the samples the same way."""
    import ray_tracing_amd as rt
    for total in (0, 1, 5, 16, 1000, 100001):
        for world in (1, 2, 3, 4, 7, 8):
            for r in range(world):
                assert rt.shard_samples(total, r, world) == shard_samples(total, r, world)
    for bad in ((10, 2, 2), (10, -1, 2), (-1, 0, 1), (10, 0, 0)):
        with pytest.raises(rt.RtError) as e:
            rt.shard_samples(*bad)
        assert e.value.code == -1


def test_render_multi_rejects_bad_arguments_without_a_gpu():
    import ctypes as C
    import ray_tracing_amd as rt
    L = rt.lib()
    cam = rt.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 20.0, 1.0, 0.0, 1.0)
    p = rt.render_params(8, 8, 1, 5, 1)
    buf = np.zeros((8, 8, 3), np.float32)
    none = (C.c_void_p * 1)(None)
    assert L.rt_render_multi(None, 1, C.byref(cam), C.byref(p), buf.ctypes.data, None, None) == -1
    assert L.rt_render_multi(none, 0, C.byref(cam), C.byref(p), buf.ctypes.data, None, None) == -1
    assert L.rt_render_multi(none, 1, C.byref(cam), C.byref(p), buf.ctypes.data, None, None) == -1      # null scene handle
    assert L.rt_render_multi(none, 1, C.byref(cam), C.byref(p), None, None, None) == -1
    assert L.rt_render_device_noalloc(None, C.byref(cam), C.byref(p), buf.ctypes.data, None, None, None) == -1


def test_pixel_shards_partition_the_image():
    """rt_shard_tiles: row bands of 8 rows dealt round-robin.  Counts add up; and, through the host build of the kernels' own
    pixel numbering (rt_set_pixel), the shards are disjoint, cover the image and render each pixel exactly as a whole-image
    call does -- also for a rectangle, for bands of another height, and for more shards than bands."""
    import ray_tracing_amd as rt
    import helpers
    for (w, h) in ((7, 1), (16, 8), (33, 9), (800, 800), (3840, 2160)):
        for n in (1, 2, 3, 8, 300):
            base = rt.render_params(w, h, 1, 5, 1)
            counts = [rt.shard_tiles(base, i, n)[1] for i in range(n)]
            assert sum(counts) == w * h and max(counts) - min(counts) <= 8 * w
            q, _ = rt.shard_tiles(base, n - 1, n)
            assert tuple(q.stripe) == (rt.RT_TILE_ROWS, n, n - 1) and tuple(q.tile) == (0, 0, 0, 0)
    for shard, n in ((0, 0), (1, 1), (-1, 2), (2, 2)):
        with pytest.raises(rt.RtError) as e:
            rt.shard_tiles(rt.render_params(8, 8, 1, 5, 1), shard, n)
        assert e.value.code == -1
    cfg, cam, b, world, _ = helpers.scene_pair("cornell_box.yml")
    flat = helpers.FlatScene(b, world)
    Wd, Hd, spp = 37, 29, 2
    base = rt.render_params(Wd, Hd, spp, 6, helpers.RENDER_SEED, sample_offset=3)
    whole, _, rays = helpers.emul_render(flat, cam, base, n_threads=4)
    assert whole.sum() > 0
    for n in (2, 3, 5):
        total, total_rays = np.zeros_like(whole), 0
        for i in range(n):
            q, count = rt.shard_tiles(base, i, n)
            part, _, r = helpers.emul_render(flat, cam, q, n_threads=4)
            assert not np.any((total != 0) & (part != 0))
            total += part; total_rays += r
        assert np.array_equal(total, whole) and total_rays == rays
    x0, y0, tw, th = 5, 3, 20, 17
    rect, _, _ = helpers.emul_render(flat, cam, rt.render_params(Wd, Hd, spp, 6, helpers.RENDER_SEED, sample_offset=3, tile=(x0, y0, tw, th)), n_threads=4)
    mask = np.zeros((Hd, Wd, 1), bool); mask[y0:y0 + th, x0:x0 + tw] = True
    assert np.array_equal(rect, np.where(mask, whole, 0))
    both = np.zeros_like(whole)
    for i in range(3):
        part, _, _ = helpers.emul_render(flat, cam, rt.render_params(Wd, Hd, spp, 6, helpers.RENDER_SEED, sample_offset=3, tile=(x0, y0, tw, th), stripe=(4, 3, i)), n_threads=4)
        both += part
    assert np.array_equal(both, rect)


def _emul_render_shard(accum, shard, n_shards):
    """The host build of the kernels' arithmetic as the renderer of a pixel shard (test infrastructure, like the oracle below)."""
    import helpers
    import ray_tracing_amd as rt
    cfg, cam, b, world, _ = helpers.scene_pair("cornell_box.yml")
    q, _ = rt.shard_tiles(rt.render_params(W, H, 1, DEPTH, helpers.RENDER_SEED), shard, n_shards)
    part, _, _ = helpers.emul_render(helpers.FlatScene(b, world), cam, q, n_threads=2)
    accum += torch.from_numpy(part)


def _tile_worker(rank, world_size, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from ray_tracing_amd.distributed import render_tile_sharded
        accum = render_tile_sharded(_emul_render_shard, H, W, "cpu")
        if rank == 0:
            np.save(out_path, accum.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_pixel_sharding_equals_single_process(tmp_path):
    """One sample per pixel over two ranks: nothing to shard by samples, so the ranks take row bands; rank 0's reduced image is
    the single-process image bit for bit (each pixel comes from one rank, the other adds zero)."""
    import helpers
    import ray_tracing_amd as rt
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "tiles.npy")
    mp.spawn(_tile_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    cfg, cam, b, world, _ = helpers.scene_pair("cornell_box.yml")
    want, _, _ = helpers.emul_render(helpers.FlatScene(b, world), cam, rt.render_params(W, H, 1, DEPTH, helpers.RENDER_SEED), n_threads=2)
    assert want.sum() > 0 and np.array_equal(got, want)


def _oracle_render_into(accum, first, count):
    import helpers
    cfg, cam, b, world, ob = helpers.scene_pair("cornell_box.yml")
    acc, _, _ = helpers.oracle_render(ob, helpers.OracleCamera(cam), W, H, count, DEPTH, helpers.RENDER_SEED, sample_offset=first, n_threads=2)
    accum += torch.from_numpy(acc.astype(np.float32))


def _worker(rank, world_size, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from ray_tracing_amd.distributed import render_sharded
        accum = render_sharded(_oracle_render_into, H, W, SPP, "cpu")
        if rank == 0:
            np.save(out_path, accum.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_sample_sharding_equals_single_process(tmp_path):
    import helpers
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "accum.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    cfg, cam, b, world, ob = helpers.scene_pair("cornell_box.yml")
    want, _, _ = helpers.oracle_render(ob, helpers.OracleCamera(cam), W, H, SPP, DEPTH, helpers.RENDER_SEED, n_threads=2)
    assert got.shape == (H, W, 3) and want.sum() > 0
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)   # same paths; only the order of the final additions differs
