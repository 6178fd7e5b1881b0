"""rt_render_device_noalloc: the render entry point that allocates no device memory (its queues, hit records, contribution slots
and counters live in a static device array).  It must give rt_render_device's image bit for bit, on two calls in a row, the
second with a smaller input (the array keeps the first call's contents), and refuse what does not fit the array."""
import numpy as np
import pytest

import ray_tracing_amd as rt
from helpers import RENDER_SEED, scene_pair

pytestmark = pytest.mark.gpu
MAX_PATHS = 1 << 18


def _both(scene, cam, W, H, spp, depth, flags=0, offset=0):
    import torch
    want = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    want_sq = torch.zeros_like(want)
    got = torch.zeros_like(want)
    got_sq = torch.zeros_like(want)
    stream = torch.cuda.current_stream().cuda_stream
    p_ref = rt.render_params(W, H, spp, depth, RENDER_SEED, sample_offset=offset, paths_per_pass=MAX_PATHS, flags=flags)
    p_na = rt.render_params(W, H, spp, depth, RENDER_SEED, sample_offset=offset, flags=flags)
    st = scene.render_device(cam, p_ref, want.data_ptr(), want_sq.data_ptr() if flags else None, stream)
    st2 = scene.render_device_noalloc(cam, p_na, got.data_ptr(), got_sq.data_ptr() if flags else None, stream)
    torch.cuda.synchronize()
    assert st2.rays == st.rays and st2.paths == st.paths and st2.passes == st.passes
    assert list(st2.rays_per_bounce) == list(st.rays_per_bounce)
    assert np.array_equal(got.cpu().numpy(), want.cpu().numpy())
    if flags:
        assert np.array_equal(got_sq.cpu().numpy(), want_sq.cpu().numpy())
    return want.cpu().numpy(), st


@pytest.mark.parametrize("name", ["final_scene.yml", "cornell.yml"])
def test_noalloc_matches_render_device_twice_in_a_row(gpu_required, name):
    import torch
    cfg, cam, b, world, _ = scene_pair(name)
    scene = rt.Scene(b, world, 0)
    big, st_big = _both(scene, cam, 512, 512, 3, cfg.max_depth, flags=rt.RT_FLAG_SUMSQ)       # 3 passes of 2^18 paths: the whole array in use
    assert st_big.passes == 3 and big.mean() > 0
    small, st_small = _both(scene, cam, 97, 61, 7, cfg.max_depth, offset=2)                    # smaller, ragged, one pass, over the leftovers
    assert st_small.passes == 1 and small.mean() > 0
    # without statistics nothing is read back: the work is only enqueued, and the result is the same once the stream has drained
    W, H = 97, 61
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    assert scene.render_device_noalloc(cam, rt.render_params(W, H, 7, cfg.max_depth, RENDER_SEED, sample_offset=2), acc.data_ptr(), None, stream,
                                       want_stats=False) is None
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), small)
    scene.close()


def test_noalloc_refuses_what_does_not_fit(gpu_required):
    import torch
    cfg, cam, b, world, _ = scene_pair("cornell.yml")
    scene = rt.Scene(b, world, 0)
    acc = torch.zeros((600, 600, 3), dtype=torch.float32, device="cuda:0")
    for kw in (dict(width=600, height=600), dict(width=64, height=64, max_depth=127), dict(width=64, height=64, flags=rt.RT_FLAG_COUNTERS),
               dict(width=64, height=64, flags=rt.RT_FLAG_TIME_TRACE)):
        args = dict(width=64, height=64, spp=1, max_depth=50, seed=1); args.update(kw)
        with pytest.raises(rt.RtError) as e:
            scene.render_device_noalloc(cam, rt.render_params(**args), acc.data_ptr(), None, None)
        assert e.value.code == -1
    st = scene.render_device_noalloc(cam, rt.render_params(512, 512, 1, 126, 1), acc.data_ptr(), None, None)   # exactly at both limits
    assert st.rays >= 512 * 512
    scene.close()


def test_noalloc_allocates_nothing_and_runs_under_graph_capture(gpu_required):
    """The property itself: the GPU's free memory is the same before and after ten calls, and a call is legal between
    hipStreamBeginCapture and hipStreamEndCapture (where an allocation or a synchronisation would fail the capture); replays
    of the captured graph give rt_render_device's image."""
    import torch
    cfg, cam, b, world, _ = scene_pair("final_scene.yml")
    scene = rt.Scene(b, world, 0)
    W, H, spp = 97, 61, 7
    p = rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED, sample_offset=2)
    want = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    acc = torch.zeros_like(want)
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_device(cam, rt.render_params(W, H, spp, cfg.max_depth, RENDER_SEED, sample_offset=2, paths_per_pass=MAX_PATHS), want.data_ptr(), None, stream)
    scene.render_device_noalloc(cam, p, acc.data_ptr(), None, stream)            # first call: code objects are loaded by now
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), want.cpu().numpy())
    free_before = torch.cuda.mem_get_info(0)[0]
    for k in range(10):
        scene.render_device_noalloc(cam, p, acc.data_ptr(), None, stream, want_stats=(k % 2 == 0))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] == free_before
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        scene.render_device_noalloc(cam, p, acc.data_ptr(), None, torch.cuda.current_stream().cuda_stream, want_stats=False)
    for _ in range(2):
        acc.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(acc.cpu().numpy(), want.cpu().numpy())
    del graph
    scene.close()
