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
