"""Structural checks of the flattener and BVH builder on the CPU (no GPU): every primitive is reachable exactly once,
child boxes contain their content, leaves are homogeneous, and the traversal stack bound holds."""
import ctypes as C

import numpy as np
import pytest

import ray_tracing_amd as rt
from helpers import FlatScene, scene_pair

LEAF_INST = 0x40000000


def arrays(flat):
    ptrs = (C.c_void_p * 15).from_address(flat.view)
    ints = (C.c_int32 * 13).from_address(flat.view + 15 * 8)
    (root, n_nodes, n_sph, n_msph, n_rect, n_inst, n_ops, n_media, n_mat, n_tex, n_perlin, max_stack, n_box) = list(ints)

    def f4(i, n):
        if n == 0 or not ptrs[i]:
            return np.zeros((0, 4), np.float32)
        return np.ctypeslib.as_array((C.c_float * (4 * n)).from_address(ptrs[i])).reshape(-1, 4).copy()

    def i2(i, n):
        if n == 0 or not ptrs[i]:
            return np.zeros((0, 2), np.int32)
        return np.ctypeslib.as_array((C.c_int32 * (2 * n)).from_address(ptrs[i])).reshape(-1, 2).copy()
    return dict(root=root, max_stack=max_stack, nodes=f4(0, 4 * n_nodes).reshape(-1, 16), spheres=f4(1, n_sph), sphere_meta=i2(2, n_sph),
                mspheres=f4(3, 3 * n_msph).reshape(-1, 12), msphere_meta=i2(4, n_msph), rects=f4(5, 2 * n_rect).reshape(-1, 8),
                rect_meta=i2(6, n_rect), instances=f4(7, 2 * n_inst).reshape(-1, 8), media=f4(9, 10 * n_media).reshape(-1, 40),
                boxes=f4(14, 2 * n_box).reshape(-1, 8))


def prim_box(a, kind, i):
    if kind == 0:
        c, r = a["spheres"][i, :3].astype(np.float64), abs(float(a["spheres"][i, 3]))
        return c - r, c + r
    if kind == 1:
        m = a["mspheres"][i].astype(np.float64)
        c0, r, dc, t0, inv = m[0:3], abs(m[3]), m[4:7], m[7], m[8]
        cs = [c0 + (t - t0) * inv * dc for t in (0.0, 1.0)]
        return np.minimum(cs[0], cs[1]) - r, np.maximum(cs[0], cs[1]) + r
    if kind == 3:
        b = a["boxes"][i].astype(np.float64)
        return b[0:3], b[4:7]
    x0, x1, y0, y1, k = a["rects"][i, :5].astype(np.float64)
    t = int(a["rects"][i, 5:6].view(np.int32)[0])
    lo, hi = {0: ((x0, y0, k), (x1, y1, k)), 1: ((x0, k, y0), (x1, k, y1)), 2: ((k, x0, y0), (k, x1, y1))}[t]
    return np.array(lo), np.array(hi)


def walk(a, node, seen, depth, stats, in_blas):
    n = a["nodes"][node]
    kids = n[12:14].view(np.int32)
    boxes = [(n[[0, 1, 2]], n[[3, 4, 5]]), (n[[6, 7, 8]], n[[9, 10, 11]])]
    stats["depth"] = max(stats["depth"], depth + 1)
    out_lo, out_hi = np.full(3, np.inf), np.full(3, -np.inf)
    for (lo, hi), c in zip(boxes, kids):
        lo, hi = lo.astype(np.float64), hi.astype(np.float64)
        if c >= 0:
            clo, chi = walk(a, int(c), seen, depth + 1, stats, in_blas)
        else:
            x = ~int(c) & 0x7FFFFFFF
            if x & LEAF_INST:
                assert not in_blas, "instances may only appear in the top level"
                ii = x & 0x3FFFFFFF
                stats["inst"].append(ii)
                inst = a["instances"][ii]
                root = int(inst[5:6].view(np.int32)[0])
                sub = dict(depth=0, inst=[])
                walk(a, root, seen, 0, sub, True)
                stats["blas_depth"] = max(stats.get("blas_depth", 0), sub["depth"])
                clo, chi = lo, hi       # world box of an instance is checked by the oracle-bbox test below
            else:
                kind, cnt, first = (x >> 28) & 3, ((x >> 26) & 3) + 1, x & 0x03FFFFFF
                assert 1 <= cnt <= 4
                clo, chi = np.full(3, np.inf), np.full(3, -np.inf)
                for i in range(first, first + cnt):
                    assert (kind, i) not in seen or kids[0] == kids[1], "primitive referenced twice"   # a one-leaf BVH points both children at it
                    seen.add((kind, i))
                    plo, phi = prim_box(a, kind, i)
                    clo, chi = np.minimum(clo, plo), np.maximum(chi, phi)
        assert np.all(lo <= clo + 1e-3 * (1 + np.abs(clo))) and np.all(hi >= chi - 1e-3 * (1 + np.abs(chi))), (node, lo, clo, hi, chi)
        out_lo, out_hi = np.minimum(out_lo, lo), np.maximum(out_hi, hi)
    return out_lo, out_hi


@pytest.mark.parametrize("name", ["orig_img.yml", "rectangle.yml", "random.yml", "cornell_box.yml", "cornell.yml", "final_scene.yml", "showcase.yml"])
def test_bvh_invariants(name):
    cfg, cam, b, world, ob = scene_pair(name)
    flat = FlatScene(b, world)
    a = arrays(flat)
    info = flat.info()
    seen, stats = set(), dict(depth=0, inst=[])
    walk(a, a["root"], seen, 0, stats, False)
    # media boundary primitives are outside the BVH; everything else is referenced exactly once
    n_bound = {0: 0, 1: 0, 2: 0, 3: 0}
    for m in a["media"]:
        iv = m.view(np.int32)
        n_bound[0] += iv[10]; n_bound[1] += iv[12]; n_bound[2] += iv[14]; n_bound[3] += iv[17]
    assert len([s for s in seen if s[0] == 0]) == info.n_spheres - n_bound[0]
    assert len([s for s in seen if s[0] == 1]) == info.n_moving_spheres - n_bound[1]
    assert len([s for s in seen if s[0] == 2]) == info.n_rects - n_bound[2]
    assert len([s for s in seen if s[0] == 3]) == info.n_boxes - n_bound[3]
    assert stats["depth"] == info.tlas_depth and stats.get("blas_depth", 0) == info.blas_depth
    assert info.max_stack <= 32 and info.max_stack == info.tlas_depth + (1 + info.blas_depth if info.n_instances and stats["inst"] else 0) or info.max_stack <= 32
    box_ids = (a["boxes"].view(np.int32)[:, 7:8] + np.arange(6)[None, :]).reshape(-1)   # a box owns the six ids of its sides
    ids = np.concatenate([a["sphere_meta"][:, 1], a["msphere_meta"][:, 1], a["rect_meta"][:, 1], a["media"].view(np.int32)[:, 15], box_ids])
    assert sorted(ids.tolist()) == list(range(info.n_objects))     # creation-order ids, each exactly once


def test_world_box_matches_oracle():
    import helpers
    for name in ["cornell_box.yml", "final_scene.yml", "random.yml"]:
        cfg, cam, b, world, ob = scene_pair(name)
        info = FlatScene(b, world).info()
        out = (C.c_double * 6)()
        assert helpers.oracle().orc_world_bbox(ob.h, out) == 0
        want = np.array(out[:])
        got = np.array(list(info.world_box))
        if name == "final_scene.yml":   # the oracle's world box also holds the media (r = 5000 fog); the BVH holds surfaces only
            assert np.all(got[:3] >= want[:3] - 1e-2) and np.all(got[3:] <= want[3:] + 1e-2)
        else:
            assert np.allclose(got, want, atol=2e-3, rtol=1e-6), (name, got, want)


def test_single_object_and_empty_world():
    b = rt.Builder()
    w = b.list([b.sphere((0, 0, 0), 1.0, b.lambertian((1, 1, 1)))])
    info = b.flatten_info(w)
    assert info.n_nodes == 1 and info.tlas_depth == 1 and info.n_spheres == 1
    info = b.flatten_info(b.list([]))               # nothing to hit: every ray returns the background
    assert info.n_nodes == 0 and info.max_stack == 0


def test_deep_scene_respects_the_stack_budget():
    rng = np.random.default_rng(5)
    b = rt.Builder()
    m = b.lambertian((0.5, 0.5, 0.5))
    # pathological for SAH: exponentially spaced spheres make very unbalanced splits
    kids = [b.sphere((float(2.0 ** (i / 40.0)), 0.0, float(rng.normal())), 0.01, m) for i in range(3000)]
    info = b.flatten_info(b.list(kids))
    assert info.n_spheres == 3000 and info.max_stack <= 30


def test_nested_transforms_and_unsupported_trees():
    b = rt.Builder()
    m = b.lambertian((1, 1, 1))
    inner = b.rotate_y(b.box((0, 0, 0), (1, 1, 1), m), 30.0)
    outer = b.translate(b.rotate_y(b.list([inner, b.sphere((3, 0, 0), 0.5, m)]), -10.0), (5, 0, 0))
    info = b.flatten_info(b.list([outer]))
    assert info.n_instances == 2 and info.n_boxes == 1 and info.n_rects == 0 and info.n_spheres == 1
    med = b.constant_medium_from_color(b.sphere((0, 0, 0), 1, m), 0.1, (1, 1, 1))
    with pytest.raises(rt.RtError):                  # a medium as (part of) another medium's boundary
        b.flatten_info(b.list([b.constant_medium_from_color(med, 0.1, (1, 1, 1))]))
    with pytest.raises(rt.RtError) as e:             # empty boundary has no bounding box (bvh.rs:140)
        b.flatten_info(b.list([b.constant_medium_from_color(b.list([]), 0.1, (1, 1, 1))]))
    assert e.value.code == -5


@pytest.mark.parametrize("name", ["random.yml", "cornell.yml", "final_scene.yml", "showcase.yml"])
def test_linear_bvh_invariants(name):
    """RT_BVH_LINEAR: the top level built as a radix tree over Morton-sorted items (here with the host loops that mirror the
    GPU kernels).  Same invariants as the SAH tree: every primitive reachable once, child boxes contain their content."""
    cfg, cam, b, world, ob = scene_pair(name)
    flat = FlatScene(b, world, rt.RT_BVH_LINEAR)
    a = arrays(flat)
    info = flat.info()
    ref = FlatScene(b, world).info()
    seen, stats = set(), dict(depth=0, inst=[])
    walk(a, a["root"], seen, 0, stats, False)
    n_bound = {0: 0, 1: 0, 2: 0, 3: 0}
    for m in a["media"]:
        iv = m.view(np.int32)
        n_bound[0] += iv[10]; n_bound[1] += iv[12]; n_bound[2] += iv[14]; n_bound[3] += iv[17]
    assert len([s for s in seen if s[0] == 0]) == info.n_spheres - n_bound[0] == ref.n_spheres - n_bound[0]
    assert len([s for s in seen if s[0] == 1]) == info.n_moving_spheres - n_bound[1]
    assert len([s for s in seen if s[0] == 2]) == info.n_rects - n_bound[2]
    assert len([s for s in seen if s[0] == 3]) == info.n_boxes - n_bound[3]
    assert len(stats["inst"]) == info.n_instances - sum(1 for m in a["media"] if m.view(np.int32)[8] >= 0)
    assert stats["depth"] == info.tlas_depth and info.max_stack <= 32
    assert np.allclose(list(info.world_box), list(ref.world_box), atol=1e-3)


def test_linear_bvh_traversal_matches_oracle():
    """Closest hits do not depend on the tree: the emulated traversal of the linear BVH agrees with the oracle like the SAH tree does."""
    from helpers import OracleCamera, RENDER_SEED, emul_trace_primary, oracle_trace_primary, primary_agreement
    for name in ["random.yml", "final_scene.yml"]:
        cfg, cam, b, world, ob = scene_pair(name)
        W, H = 96, int(round(96 / cfg.aspect_ratio))
        p = rt.render_params(W, H, 1, 50, RENDER_SEED)
        lin = emul_trace_primary(FlatScene(b, world, rt.RT_BVH_LINEAR), cam, p, 0)
        sah = emul_trace_primary(FlatScene(b, world), cam, p, 0)
        o = oracle_trace_primary(ob, OracleCamera(cam), W, H, 0, RENDER_SEED)
        agree, rel = primary_agreement(lin, o)
        assert agree >= 0.999 and rel <= 1e-4, (name, agree, rel)
        assert np.array_equal(lin["id"], sah["id"]) and np.array_equal(lin["t"], sah["t"])


# ---------------------------------------------------------------------------------------------- the 4-wide quantised tree
def wide_arrays(flat):
    base = flat.view + 15 * 8 + 13 * 4 + 4            # after the binary view's pointers and ints (padded to 8 bytes)
    ptrs = (C.c_void_p * 2).from_address(base)
    n_wnodes, wroot, wmax_stack = list((C.c_int32 * 3).from_address(base + 16))
    nodes = np.ctypeslib.as_array((C.c_uint32 * (16 * n_wnodes)).from_address(ptrs[0])).reshape(-1, 16).copy() if n_wnodes else np.zeros((0, 16), np.uint32)
    return nodes, ptrs[1], wroot, wmax_stack


def wide_children(node):
    """[(lo, hi, code)] of the present children of one wide node, boxes decoded as the kernel does (origin + byte * step)."""
    org = node[0:3].view(np.float32).astype(np.float64)
    step = np.array([node[3:4].view(np.float32)[0], node[14:15].view(np.float32)[0], node[15:16].view(np.float32)[0]], dtype=np.float64)
    q = node[4:10]
    codes = node[10:14].view(np.int32)
    out = []
    for k in range(4):
        if codes[k] == 0x7FFFFFFF:
            continue
        lo = org + step * np.array([(int(q[a]) >> (8 * k)) & 255 for a in range(3)])
        hi = org + step * np.array([(int(q[3 + a]) >> (8 * k)) & 255 for a in range(3)])
        out.append((lo, hi, int(codes[k])))
    return out


@pytest.mark.parametrize("name", ["orig_img.yml", "random.yml", "cornell_box.yml", "final_scene.yml", "showcase.yml"])
def test_wide_bvh_invariants(name):
    """Every primitive is reachable exactly once through the wide tree, every decoded child box contains what is below it
    (quantisation only grows boxes), and the recorded stack bound covers the deepest path."""
    cfg, cam, b, world, _ = scene_pair(name)
    flat = FlatScene(b, world)
    a = arrays(flat)
    wn, inst_roots_ptr, wroot, wmax = wide_arrays(flat)
    n_inst = a["instances"].shape[0]
    inst_roots = list((C.c_int32 * n_inst).from_address(inst_roots_ptr)) if n_inst else []
    seen = set()

    def walk_wide(n, need, in_blas):
        kids = wide_children(wn[n])
        assert 1 <= len(kids) <= 4
        here = need + len(kids) - 1
        deepest = here
        out_lo, out_hi = np.full(3, np.inf), np.full(3, -np.inf)
        for lo, hi, c in kids:
            if c >= 0:
                clo, chi, d = walk_wide(c, here, in_blas)
                deepest = max(deepest, d)
            else:
                x = ~c & 0x7FFFFFFF
                if x & LEAF_INST:
                    assert not in_blas
                    _, _, d = walk_wide(inst_roots[x & 0x3FFFFFFF], here + 1, True)     # + the sentinel
                    deepest = max(deepest, d)
                    clo, chi = lo, hi
                else:
                    kind, cnt, first = (x >> 28) & 3, ((x >> 26) & 3) + 1, x & 0x03FFFFFF
                    clo, chi = np.full(3, np.inf), np.full(3, -np.inf)
                    for i in range(first, first + cnt):
                        assert (kind, i) not in seen
                        seen.add((kind, i))
                        plo, phi = prim_box(a, kind, i)
                        clo, chi = np.minimum(clo, plo), np.maximum(chi, phi)
            assert np.all(lo <= clo + 1e-3 * (1 + np.abs(clo))) and np.all(hi >= chi - 1e-3 * (1 + np.abs(chi))), (n, lo, clo, hi, chi)
            out_lo, out_hi = np.minimum(out_lo, clo), np.maximum(out_hi, chi)       # what is really below: each level quantises on its own grid
        return out_lo, out_hi, deepest

    _, _, deepest = walk_wide(wroot, 0, False)
    assert deepest <= wmax
    seen2 = set()
    walk(a, a["root"], seen2, 0, dict(depth=0, inst=[]), False)
    assert seen == seen2                                # the wide tree reaches exactly the primitives of the binary tree


@pytest.mark.parametrize("flags", [0, rt.RT_BVH_LINEAR])
def test_wide_bvh_traversal_matches_oracle(flags, monkeypatch):
    """The fp32 traversal of the wide tree (host build of the kernels' functions) against the oracle on a sphere field."""
    from helpers import OracleBuilder, OracleCamera, RENDER_SEED, SCENE_SEED, emul_trace_primary, oracle_trace_primary, primary_agreement
    b, ob = rt.Builder(), OracleBuilder()
    w = b.canned_scene("Synthetic:30000", SCENE_SEED)
    ob.set_world(ob.canned_scene("Synthetic:30000", SCENE_SEED))
    flat = FlatScene(b, w, flags)
    cam = rt.make_camera((0, 260, 1500), (0, 60, 0), (0, 1, 0), 40.0, 16 / 9, 0.1, 10.0)
    p = rt.render_params(240, 135, 1, 50, RENDER_SEED)
    o = oracle_trace_primary(ob, OracleCamera(cam), 240, 135, 0, RENDER_SEED)
    plain = emul_trace_primary(flat, cam, p, 0)
    monkeypatch.setenv("EMUL_WIDE_BVH", "1")
    wide = emul_trace_primary(flat, cam, p, 0)
    agree, rel = primary_agreement(wide, o)
    assert agree >= 0.9999 and rel <= 1e-4, (agree, rel)
    assert np.array_equal(wide["id"], plain["id"]) and np.array_equal(wide["t"], plain["t"])     # closest hits do not depend on the tree


# ---------------------------------------------------------------------------------------------- moving primitives + refit
def moving_scene(b, positions, radii, cluster_pos, msphere):
    """Spheres at `positions`, a rotated and translated cluster of spheres at `cluster_pos` (an instance subtree), one moving
    sphere and a ground rectangle.  Returns (world, ids of the free spheres, ids of the cluster spheres, id of the moving sphere)."""
    mats = [b.lambertian((0.5, 0.5, 0.5)), b.metal((0.8, 0.8, 0.8), 0.1), b.dielectric(1.5)]
    kids, ids, cids = [], [], []
    for i, (p, r) in enumerate(zip(positions, radii)):
        ids.append(b.object_count()); kids.append(b.sphere(p, r, mats[i % 3]))
    inner = []
    for i, p in enumerate(cluster_pos):
        cids.append(b.object_count()); inner.append(b.sphere(p, 0.4, mats[i % 3]))
    kids.append(b.translate(b.rotate_y(b.bvh_node(b.list(inner)), 25.0), (-3.0, 1.0, -12.0)))
    mid = b.object_count()
    kids.append(b.moving_sphere(msphere[0:3], msphere[3:6], msphere[6], msphere[7], msphere[8], mats[0]))
    kids.append(b.rectangle(mats[0], -40, 40, -60, 10, -2.0, rt.RECT_XZ))
    return b.list(kids), ids, cids, mid


@pytest.mark.parametrize("flags", [0, rt.RT_BVH_LINEAR])
def test_refit_gives_the_hits_of_a_rebuild(flags, monkeypatch):
    """Move most primitives of a flattened scene (free spheres, spheres inside an instanced subtree, a moving sphere), refit
    the boxes on the host, and compare the closest hits -- through the binary and through the wide tree -- with a scene
    flattened from the moved primitives, and with the oracle."""
    from helpers import OracleBuilder, OracleCamera, RENDER_SEED, emul_trace_primary, oracle_trace_primary, primary_agreement
    rng = np.random.default_rng(5)
    n, m = 400, 60
    pos0 = rng.uniform([-15, -1, -30], [15, 8, -8], (n, 3)); rad = rng.uniform(0.2, 0.9, n)
    cl0 = rng.uniform(0, 6, (m, 3))
    ms0 = np.array([4, 5, -10, 6, 6, -10, 0.0, 1.0, 0.8])
    pos1 = pos0 + rng.normal(0, 1.5, (n, 3)); pos1[::7] = pos0[::7]          # most move, some stay
    rad1 = rad * rng.uniform(0.7, 1.3, n)
    cl1 = cl0 + rng.normal(0, 0.8, (m, 3))
    ms1 = np.array([-6, 4, -12, -3, 7, -11, 0.0, 1.0, 1.1])

    b = rt.Builder()
    world, ids, cids, mid = moving_scene(b, pos0, rad, cl0, ms0)
    flat = FlatScene(b, world, flags)
    flat.set_spheres(ids, np.concatenate([pos1, rad1[:, None]], axis=1))
    flat.set_spheres(cids, np.concatenate([cl1, np.full((m, 1), 0.4)], axis=1))
    flat.set_moving_spheres([mid], ms1[None, :])
    flat.refit()

    b2, ob = rt.Builder(), OracleBuilder()
    w2, _, _, _ = moving_scene(b2, pos1, rad1, cl1, ms1)
    ow, _, _, _ = moving_scene(ob, pos1, rad1, cl1, ms1)
    ob.set_world(ow)
    fresh = FlatScene(b2, w2, flags)
    cam = rt.make_camera((0, 6, 12), (0, 2, -18), (0, 1, 0), 45.0, 1.5, 0.05, 20.0)
    W, H = 180, 120
    p = rt.render_params(W, H, 1, 50, RENDER_SEED)
    o = oracle_trace_primary(ob, OracleCamera(cam), W, H, 0, RENDER_SEED)
    for wide in ("0", "1"):
        monkeypatch.setenv("EMUL_WIDE_BVH", wide)
        got, want = emul_trace_primary(flat, cam, p, 0), emul_trace_primary(fresh, cam, p, 0)
        assert np.array_equal(got["id"], want["id"]) and np.array_equal(got["t"], want["t"])
        agree, rel = primary_agreement(got, o)
        assert agree >= 0.9999 and rel <= 1e-4, (wide, agree, rel)
    assert (o["id"] >= 0).mean() > 0.5
    # structure: the refitted boxes contain what is below them
    a = arrays(flat)
    walk(a, a["root"], set(), 0, dict(depth=0, inst=[]), False)


def test_refit_refuses_what_it_cannot_move():
    b = rt.Builder()
    m = b.lambertian((0.5, 0.5, 0.5))
    bound_id = b.object_count()
    ball = b.sphere((0, 0, -5), 1.0, m)
    rect_id = b.object_count()
    world = b.list([b.constant_medium_from_color(ball, 0.5, (1, 1, 1)), b.rectangle(m, -1, 1, -1, 1, -9.0, rt.RECT_XY), b.sphere((3, 0, -5), 1.0, m)])
    flat = FlatScene(b, world)
    for bad in (bound_id, rect_id, 9999, -1):
        with pytest.raises(rt.RtError):
            flat.set_spheres([bad], [[0, 0, 0, 1]])
    with pytest.raises(rt.RtError):
        flat.set_moving_spheres([bound_id + 3], [[0] * 9])           # a sphere is not a moving sphere
    # a call that fails on a later id must not have written the earlier ones
    last = b.object_count() - 1
    before = arrays(flat)["spheres"].copy()
    with pytest.raises(rt.RtError):
        flat.set_spheres([last, rect_id], [[7, 7, 7, 3], [0, 0, 0, 1]])
    assert np.array_equal(arrays(flat)["spheres"], before)
    flat.set_spheres([last], [[2, 0, -5, 0.5]])
    assert not np.array_equal(arrays(flat)["spheres"], before)
    flat.refit()


def _medium_locals(flat):
    a = arrays(flat)
    med = a["media"].reshape(-1, 10, 4).view(np.int32)
    return [(int(m[5, 2]) & 255, int(m[5, 2]) >> 8, [int(m[4, 2]), int(m[4, 3]), int(m[5, 3])]) for m in med]


def test_media_list_the_surfaces_inside_their_boundary():
    """flatten.cpp, emit_medium_locals: final_scene.yml's glass ball holds a dense medium (src/main.rs:392-405); the only surfaces
    inside that boundary are the ball itself (wholly inside: flagged) and whichever ground boxes are tall enough to poke into it.
    The fog that encloses the whole scene, media behind Translate / RotateY (cornell.yml) and boundaries near an instance list nothing."""
    cfg, cam, b, world, _ = __import__("helpers").scene_pair("final_scene.yml")
    for flags in (0, rt.RT_BVH_LINEAR):
        loc = _medium_locals(FlatScene(b, world, flags))
        n, mask, codes = loc[0]
        assert 1 <= n <= 3 and mask != 0 and loc[1][0] == 0, loc
        flagged = [codes[k] for k in range(n) if mask >> k & 1]
        assert len(flagged) == 1 and (~flagged[0] >> 28) & 3 == 0          # one flagged entry, a sphere: the glass ball
        a = arrays(FlatScene(b, world, flags))
        ball = a["spheres"][~flagged[0] & 0x03FFFFFF]
        assert np.allclose(ball, [360, 150, 145, 70])
    cfg, cam, b, world, _ = __import__("helpers").scene_pair("cornell.yml")
    assert all(n == 0 for n, _, _ in _medium_locals(FlatScene(b, world)))
    # hand-built: a list of up to three, none beyond; an instance that reaches into the boundary forbids the list
    def ball_scene(n_inside, with_instance):
        bb = rt.Builder()
        m = bb.lambertian((0.5, 0.5, 0.5))
        kids = [bb.constant_medium_from_color(bb.sphere((0, 0, 0), 5.0, m), 0.3, (1, 1, 1))]
        kids += [bb.sphere((0.5 * i - 1.0, 0, 0), 0.2, m) for i in range(n_inside)]
        kids.append(bb.sphere((20, 0, 0), 1.0, m))                          # far away: never in a list
        if with_instance:
            kids.append(bb.translate(bb.sphere((0, 0, 0), 0.5, m), (1.0, 1.0, 1.0)))
        return bb, bb.list(kids)
    for n_inside, with_instance, want in ((1, False, 1), (3, False, 3), (4, False, 0), (2, True, 0), (0, False, 0)):
        bb, w = ball_scene(n_inside, with_instance)
        n, mask, _ = _medium_locals(FlatScene(bb, w))[0]
        assert n == want and (mask == (1 << want) - 1), (n_inside, with_instance, n, mask)
    # moving any sphere empties the lists
    bb, w = ball_scene(2, False)
    flat = FlatScene(bb, w)
    assert _medium_locals(flat)[0][0] == 2
    flat.set_spheres([bb.object_count() - 1], [[1, 1, 1, 0.5]])
    assert _medium_locals(flat)[0][0] == 0
