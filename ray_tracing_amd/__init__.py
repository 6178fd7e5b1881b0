"""ray_tracing_amd -- Python view of librt_hip.so, the MI355X render path for mkagie/ray_tracing scenes.

The package is a thin ctypes binding of the C ABI declared in ``include/rt_hip.h`` plus a
``Builder`` whose method names follow the reference's prelude (``src/lib.rs:23-35``), so that a
scene is written the way ``src/main.rs:13-451`` writes it.  All arithmetic of the render path runs
in the HIP kernels; there is no CPU fallback: if the shared library is missing the import of
``lib()`` fails, and without a GPU ``Scene`` creation fails with the library's error.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_hip.so")
_lib = None

RT_FLAG_SUMSQ = 1
RT_FLAG_COUNTERS = 2
RT_FLAG_TIME_TRACE = 4
RT_FLAG_WAVEFRONT_PRIMARY = 8   # rt_trace_primary through the kernels a render runs (ray queue + the scene's closest-hit kernel)
RT_FLAG_TILES = 16              # rt_render_multi: shard pixels (row bands) instead of samples
RT_TILE_ROWS = 8
RT_BVH_LINEAR = 1            # rt_scene_create_ex / rt_flat_create_ex: linear top-level BVH (built on the GPU for a Scene)
RECT_XY, RECT_XZ, RECT_YZ = 0, 1, 2   # RectangleType, src/objects.rs:339-343


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rt error %d: %s" % (code, msg))
        self.code = code


class RtConfig(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("look_from", C.c_double * 3), ("look_at", C.c_double * 3),
                ("v_up", C.c_double * 3), ("vertical_fov_deg", C.c_double), ("aspect_ratio", C.c_double),
                ("aperture", C.c_double), ("focus_distance", C.c_double), ("scene_type", C.c_char * 32),
                ("n_list_objects", C.c_int32), ("shutter_open", C.c_double), ("shutter_close", C.c_double),
                ("background", C.c_float * 3)]


class RtSceneInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_nodes", "n_spheres", "n_moving_spheres", "n_rects", "n_instances",
                                         "n_media", "n_materials", "n_textures", "n_perlin", "tlas_depth",
                                         "blas_depth", "max_stack", "n_objects", "device", "n_boxes")] + \
               [("world_box", C.c_double * 6), ("device_bytes", C.c_uint64), ("bvh_build_seconds", C.c_double), ("bvh_refit_seconds", C.c_double)]


class RtCamera(C.Structure):
    _fields_ = [("look_from", C.c_double * 3), ("look_at", C.c_double * 3), ("v_up", C.c_double * 3),
                ("vertical_fov_deg", C.c_double), ("aspect_ratio", C.c_double), ("aperture", C.c_double),
                ("focus_dist", C.c_double), ("shutter_open", C.c_double), ("shutter_close", C.c_double)]


class RtRenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("sample_offset", C.c_int32),
                ("max_depth", C.c_int32), ("background", C.c_float * 3), ("seed", C.c_uint64),
                ("paths_per_pass", C.c_int32), ("flags", C.c_int32), ("tile", C.c_int32 * 4), ("stripe", C.c_int32 * 3),
                ("reserved_", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("rays_per_bounce", C.c_uint64 * 64),
                ("seconds", C.c_double), ("kernel_ms_extend", C.c_double), ("extend_launches", C.c_uint64),
                ("passes", C.c_int32), ("work", C.c_uint64 * 16), ("kernel_ms_trace", C.c_double), ("trace_launches", C.c_uint64),
                ("multi_gpus", C.c_int32), ("multi_comm_ranks", C.c_int32), ("multi_tiles", C.c_int32), ("reserved_", C.c_int32),
                ("rays_settled", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/rt_hip.h declares
_P = C.c_void_p
_D3 = C.POINTER(C.c_double)
API = {
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rtb_create": (C.c_int, [C.POINTER(_P)]),
    "rtb_destroy": (None, [_P]),
    "rtb_object_count": (C.c_int, [_P]),
    "rtb_solid_color": (C.c_int32, [_P, C.c_double, C.c_double, C.c_double]),
    "rtb_checker": (C.c_int32, [_P, C.c_int32, C.c_int32]),
    "rtb_noise": (C.c_int32, [_P, C.c_double, _D3, C.POINTER(C.c_int32)]),
    "rtb_lambertian": (C.c_int32, [_P, C.c_int32]),
    "rtb_metal": (C.c_int32, [_P, C.c_double, C.c_double, C.c_double, C.c_double]),
    "rtb_dielectric": (C.c_int32, [_P, C.c_double]),
    "rtb_diffuse_light": (C.c_int32, [_P, C.c_int32]),
    "rtb_isotropic": (C.c_int32, [_P, C.c_int32]),
    "rtb_sphere": (C.c_int32, [_P, _D3, C.c_double, C.c_int32]),
    "rtb_moving_sphere": (C.c_int32, [_P, _D3, _D3, C.c_double, C.c_double, C.c_double, C.c_int32]),
    "rtb_rectangle": (C.c_int32, [_P, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "rtb_box": (C.c_int32, [_P, _D3, _D3, C.c_int32]),
    "rtb_list": (C.c_int32, [_P, C.POINTER(C.c_int32), C.c_int]),
    "rtb_bvh_node": (C.c_int32, [_P, C.c_int32, C.c_double, C.c_double]),
    "rtb_translate": (C.c_int32, [_P, C.c_int32, _D3]),
    "rtb_rotate_y": (C.c_int32, [_P, C.c_int32, C.c_double]),
    "rtb_constant_medium": (C.c_int32, [_P, C.c_int32, C.c_double, C.c_int32]),
    "rt_config_load": (C.c_int, [C.c_char_p, C.POINTER(RtConfig)]),
    "rt_config_build_scene": (C.c_int, [C.c_char_p, C.c_uint64, _P, C.POINTER(C.c_int32)]),
    "rt_canned_scene": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.POINTER(C.c_int32)]),
    "rt_scene_create": (C.c_int, [_P, C.c_int32, C.c_int, C.POINTER(_P)]),
    "rt_scene_create_ex": (C.c_int, [_P, C.c_int32, C.c_int, C.c_int, C.POINTER(_P)]),
    "rt_scene_destroy": (None, [_P]),
    "rt_scene_info": (C.c_int, [_P, C.POINTER(RtSceneInfo)]),
    "rt_flat_create": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    "rt_flat_create_ex": (C.c_int, [_P, C.c_int32, C.c_int, C.POINTER(_P)]),
    "rt_flat_view": (_P, [_P]),
    "rt_flat_info": (C.c_int, [_P, C.POINTER(RtSceneInfo)]),
    "rt_flat_destroy": (None, [_P]),
    "rt_flat_set_spheres": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32, _D3]),
    "rt_flat_set_moving_spheres": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32, _D3]),
    "rt_flat_refit": (C.c_int, [_P]),
    "rt_scene_set_spheres": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32, _D3]),
    "rt_scene_set_moving_spheres": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32, _D3]),
    "rt_scene_refit": (C.c_int, [_P]),
    "rt_render": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtRenderParams), _P, _P, C.POINTER(RtStats)]),
    "rt_render_device": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtRenderParams), _P, _P, _P, C.POINTER(RtStats)]),
    "rt_render_device_noalloc": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtRenderParams), _P, _P, _P, C.POINTER(RtStats)]),
    "rt_release_workspace": (C.c_int, [C.c_int]),
    "rt_render_multi": (C.c_int, [C.POINTER(_P), C.c_int32, C.POINTER(RtCamera), C.POINTER(RtRenderParams), _P, _P, C.POINTER(RtStats)]),
    "rt_shard_samples": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rt_shard_tiles": (C.c_int, [C.POINTER(RtRenderParams), C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "rt_multi_shutdown": (C.c_int, []),
    "rt_trace_primary": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtRenderParams), C.c_int32, _P, _P, _P, _P, _P]),
    "rt_scatter_rays": (C.c_int, [_P, _P, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, _P]),
    "rt_tonemap_u8": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32]),
    "rt_write_image": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32]),
    "rt_accum_save": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32, C.c_int64, C.c_uint64]),
    "rt_accum_load": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]),
}


def lib():
    """Loads librt_hip.so (built in-tree by ray_tracing_amd/build.py).  Fails loudly if it is absent."""
    global _lib
    if _lib is None:
        path = os.environ.get("RT_HIP_LIB") or LIB_PATH      # RT_HIP_LIB: another build of the same library (A/B measurements)
        if not os.path.exists(path):
            raise ImportError("%s is missing: run `python -m ray_tracing_amd.build` (needs hipcc); "
                              "there is no CPU fallback for the render path" % path)
        # PyTorch-ROCm bundles its own HIP runtime (same soname as the system one).  Two runtimes in one process
        # leave the second user without a GPU, so when torch is installed it is imported first and the library
        # binds to the runtime torch brought in; device pointers are then shared between the two.
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(path)
        for name, (res, args) in API.items():
            fn = getattr(L, name, None)
            if fn is None:
                if path == LIB_PATH:
                    raise ImportError("%s does not export %s: rebuild it (python -m ray_tracing_amd.build)" % (path, name))
                continue                                      # an older build given through RT_HIP_LIB may lack newer entry points
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode())
    return rc


def _handle(h):
    if h < 0:
        raise RtError(h, lib().rt_last_error().decode())
    return h


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


def device_count():
    n = C.c_int(0)
    rc = lib().rt_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Builder:
    """Scene graph under construction.  Methods are the reference's constructors (file:line in include/rt_hip.h)."""

    def __init__(self):
        self._h = _P()
        _check(lib().rtb_create(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and lib is not None:   # module globals are gone at interpreter exit
            _lib.rtb_destroy(self._h)
            self._h = None

    # textures.rs
    def solid_color(self, c): return _handle(lib().rtb_solid_color(self._h, c[0], c[1], c[2]))
    def checker(self, even, odd): return _handle(lib().rtb_checker(self._h, even, odd))
    def checker_from_solid_colors(self, c1, c2): return self.checker(self.solid_color(c1), self.solid_color(c2))

    def noise(self, scale, ranvec, perm):
        rv = np.ascontiguousarray(ranvec, dtype=np.float64).reshape(768)
        pm = np.ascontiguousarray(perm, dtype=np.int32).reshape(768)
        return _handle(lib().rtb_noise(self._h, scale, rv.ctypes.data_as(_D3), pm.ctypes.data_as(C.POINTER(C.c_int32))))

    # materials.rs
    def lambertian_from_texture(self, tex): return _handle(lib().rtb_lambertian(self._h, tex))
    def lambertian(self, color): return self.lambertian_from_texture(self.solid_color(color))
    def metal(self, color, fuzz): return _handle(lib().rtb_metal(self._h, color[0], color[1], color[2], fuzz))
    def dielectric(self, ir): return _handle(lib().rtb_dielectric(self._h, ir))
    def diffuse_light(self, tex): return _handle(lib().rtb_diffuse_light(self._h, tex))
    def diffuse_light_from_color(self, color): return self.diffuse_light(self.solid_color(color))
    def isotropic(self, tex): return _handle(lib().rtb_isotropic(self._h, tex))

    # objects.rs / transrot.rs / mediums.rs / bvh.rs
    def sphere(self, center, radius, mat): return _handle(lib().rtb_sphere(self._h, _d3(center), radius, mat))

    def moving_sphere(self, c0, c1, t0, t1, radius, mat):
        return _handle(lib().rtb_moving_sphere(self._h, _d3(c0), _d3(c1), t0, t1, radius, mat))

    def rectangle(self, mat, x0, x1, y0, y1, k, rtype):
        return _handle(lib().rtb_rectangle(self._h, mat, x0, x1, y0, y1, k, rtype))

    def box(self, box_min, box_max, mat): return _handle(lib().rtb_box(self._h, _d3(box_min), _d3(box_max), mat))

    def list(self, children):
        arr = (C.c_int32 * len(children))(*children)
        return _handle(lib().rtb_list(self._h, arr, len(children)))

    def bvh_node(self, lst, time0=0.0, time1=1.0): return _handle(lib().rtb_bvh_node(self._h, lst, time0, time1))
    def translate(self, obj, offset): return _handle(lib().rtb_translate(self._h, obj, _d3(offset)))
    def rotate_y(self, obj, angle_deg): return _handle(lib().rtb_rotate_y(self._h, obj, angle_deg))

    def constant_medium(self, boundary, density, tex):
        return _handle(lib().rtb_constant_medium(self._h, boundary, density, tex))

    def constant_medium_from_color(self, boundary, density, color):
        return self.constant_medium(boundary, density, self.solid_color(color))

    def object_count(self): return lib().rtb_object_count(self._h)

    # main.rs:520-529
    def canned_scene(self, scene_type, scene_seed):
        w = C.c_int32(-1)
        _check(lib().rt_canned_scene(self._h, scene_type.encode(), scene_seed, C.byref(w)))
        return w.value

    def scene_from_config(self, path, scene_seed):
        w = C.c_int32(-1)
        _check(lib().rt_config_build_scene(path.encode(), scene_seed, self._h, C.byref(w)))
        return w.value

    def flatten_info(self, world, flags=0):
        """Host-only flatten; returns RtSceneInfo (no GPU needed).  flags: RT_BVH_LINEAR."""
        f = _P()
        _check(lib().rt_flat_create_ex(self._h, world, flags, C.byref(f)))
        try:
            info = RtSceneInfo()
            _check(lib().rt_flat_info(f, C.byref(info)))
            return info
        finally:
            lib().rt_flat_destroy(f)


def load_config(path):
    cfg = RtConfig()
    _check(lib().rt_config_load(path.encode(), C.byref(cfg)))
    return cfg


def camera_from_config(cfg, shutter_open=None, shutter_close=None):
    """Camera::new as main.rs:536-547 calls it (the shutter is 0..1 there; a config may set camera.shutter_open / shutter_close)."""
    shutter_open = cfg.shutter_open if shutter_open is None else shutter_open
    shutter_close = cfg.shutter_close if shutter_close is None else shutter_close
    cam = RtCamera()
    for i in range(3):
        cam.look_from[i], cam.look_at[i], cam.v_up[i] = cfg.look_from[i], cfg.look_at[i], cfg.v_up[i]
    cam.vertical_fov_deg, cam.aspect_ratio = cfg.vertical_fov_deg, cfg.aspect_ratio
    cam.aperture, cam.focus_dist = cfg.aperture, cfg.focus_distance
    cam.shutter_open, cam.shutter_close = shutter_open, shutter_close
    return cam


def make_camera(look_from, look_at, v_up, vfov, aspect, aperture, focus_dist, shutter_open=0.0, shutter_close=1.0):
    cam = RtCamera()
    for i in range(3):
        cam.look_from[i], cam.look_at[i], cam.v_up[i] = look_from[i], look_at[i], v_up[i]
    cam.vertical_fov_deg, cam.aspect_ratio, cam.aperture, cam.focus_dist = vfov, aspect, aperture, focus_dist
    cam.shutter_open, cam.shutter_close = shutter_open, shutter_close
    return cam


def render_params(width, height, spp, max_depth, seed, sample_offset=0, background=(0.0, 0.0, 0.0),
                  paths_per_pass=0, flags=0, tile=None, stripe=None):
    """tile = (x0, y0, w, h) and stripe = (band_rows, n_shards, shard) select the pixels rendered (None: all)."""
    p = RtRenderParams()
    p.width, p.height, p.spp, p.sample_offset, p.max_depth = width, height, spp, sample_offset, max_depth
    p.background[0], p.background[1], p.background[2] = background
    p.seed, p.paths_per_pass, p.flags = seed, paths_per_pass, flags
    if tile is not None:
        p.tile[0], p.tile[1], p.tile[2], p.tile[3] = tile
    if stripe is not None:
        p.stripe[0], p.stripe[1], p.stripe[2] = stripe
    return p


def shard_tiles(params, shard, n_shards):
    """A copy of `params` that renders pixel shard `shard` of `n_shards` (rt_shard_tiles), and the shard's pixel count."""
    q = _with_flags(params, params.flags)
    n = C.c_int64(0)
    _check(lib().rt_shard_tiles(C.byref(q), n_shards, shard, C.byref(n)))
    return q, n.value


def multi_shutdown():
    """Destroys the RCCL communicators kept by render_multi."""
    _check(lib().rt_multi_shutdown())


def _set_prims(fn, handle, object_ids, values, width):
    ids = np.ascontiguousarray(object_ids, dtype=np.int32).reshape(-1)
    vals = np.ascontiguousarray(values, dtype=np.float64).reshape(-1, width)
    assert vals.shape[0] == ids.shape[0]
    _check(fn(handle, ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.shape[0], vals.ctypes.data_as(_D3)))


def _with_flags(params, flags):
    q = RtRenderParams()
    C.memmove(C.byref(q), C.byref(params), C.sizeof(RtRenderParams))
    q.flags = flags
    return q


def release_workspace(device=0):
    """Frees the GPU's shared wavefront workspace (queues, hit records, contribution slots); the next render allocates it again."""
    _check(lib().rt_release_workspace(device))


def shard_samples(total_spp, shard, n_shards):
    """(first_sample, n_samples) of one shard, from the library's own arithmetic (rt_shard_samples)."""
    first, count = C.c_int32(0), C.c_int32(0)
    _check(lib().rt_shard_samples(total_spp, n_shards, shard, C.byref(first), C.byref(count)))
    return first.value, count.value


def render_multi(scenes, cam, params, accum=None, want_sumsq=False):
    """rt_render_multi: one Scene per GPU (the same world on each), sample ranges sharded across them inside this process,
    one RCCL reduce to the first scene's GPU.  Returns (accum, sumsq|None, stats) like Scene.render."""
    h, w = params.height, params.width
    if accum is None:
        accum = np.zeros((h, w, 3), dtype=np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.shape == (h, w, 3)
    sumsq = None
    if want_sumsq:
        params = _with_flags(params, params.flags | RT_FLAG_SUMSQ)
        sumsq = np.zeros((h, w, 3), dtype=np.float32)
    handles = (_P * len(scenes))(*[s._h for s in scenes])
    st = RtStats()
    _check(lib().rt_render_multi(handles, len(scenes), C.byref(cam), C.byref(params), accum.ctypes.data,
                                 sumsq.ctypes.data if sumsq is not None else None, C.byref(st)))
    return accum, sumsq, st


class Scene:
    """A flattened scene resident on one GPU (rt_scene_create)."""

    def __init__(self, builder, world, device=0, flags=0):
        self._h = _P()
        _check(lib().rt_scene_create_ex(builder._h, world, device, flags, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def info(self):
        info = RtSceneInfo()
        _check(lib().rt_scene_info(self._h, C.byref(info)))
        return info

    def set_spheres(self, object_ids, xyzr):
        """Rewrites spheres (by object id, creation order) in place: xyzr[i] = (cx, cy, cz, r).  Call refit() afterwards."""
        _set_prims(lib().rt_scene_set_spheres, self._h, object_ids, xyzr, 4)

    def set_moving_spheres(self, object_ids, values):
        """values[i] = (c0 xyz, c1 xyz, time0, time1, radius)."""
        _set_prims(lib().rt_scene_set_moving_spheres, self._h, object_ids, values, 9)

    def refit(self):
        """Recomputes every box of the scene's trees on the GPU from the primitives as they are now (the trees keep their shape)."""
        _check(lib().rt_scene_refit(self._h))

    def render(self, cam, params, accum=None, want_sumsq=False):
        """Adds the per-pixel colour sums of params.spp samples to `accum` ([H,W,3] float32).  Returns (accum, sumsq|None, stats)."""
        h, w = params.height, params.width
        if accum is None:
            accum = np.zeros((h, w, 3), dtype=np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.shape == (h, w, 3)
        sumsq = None
        if want_sumsq:
            params = _with_flags(params, params.flags | RT_FLAG_SUMSQ)      # a copy: the caller's params keep their flags
            sumsq = np.zeros((h, w, 3), dtype=np.float32)
        st = RtStats()
        _check(lib().rt_render(self._h, C.byref(cam), C.byref(params), accum.ctypes.data,
                               sumsq.ctypes.data if sumsq is not None else None, C.byref(st)))
        return accum, sumsq, st

    def render_device(self, cam, params, accum_ptr, sumsq_ptr=None, stream=None):
        """Same, into device buffers given as raw addresses (e.g. torch.Tensor.data_ptr())."""
        st = RtStats()
        _check(lib().rt_render_device(self._h, C.byref(cam), C.byref(params), accum_ptr, sumsq_ptr, stream, C.byref(st)))
        return st

    def render_device_noalloc(self, cam, params, accum_ptr, sumsq_ptr=None, stream=None, want_stats=True):
        """rt_render_device_noalloc: the same render out of the library's static device workspace (no device allocation).
        With want_stats=False nothing is read back and the call returns with the work enqueued on `stream`."""
        st = RtStats() if want_stats else None
        _check(lib().rt_render_device_noalloc(self._h, C.byref(cam), C.byref(params), accum_ptr, sumsq_ptr, stream,
                                              C.byref(st) if want_stats else None))
        return st

    def trace_primary(self, cam, params, sample=0, want_normal=False, want_uv=False, wavefront=False):
        """Closest hit of one sample per pixel.  wavefront=True: through the ray queue and the closest-hit kernel a render of
        this scene launches (RT_FLAG_WAVEFRONT_PRIMARY) instead of the plain per-ray search."""
        if wavefront:
            params = _with_flags(params, params.flags | RT_FLAG_WAVEFRONT_PRIMARY)
        n = params.width * params.height
        ids = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        normal = np.empty((n, 3), dtype=np.float32) if want_normal else None
        front = np.empty(n, dtype=np.int32) if want_normal else None
        uv = np.empty((n, 2), dtype=np.float32) if want_uv else None
        _check(lib().rt_trace_primary(self._h, C.byref(cam), C.byref(params), sample, ids.ctypes.data, t.ctypes.data,
                                      normal.ctypes.data if want_normal else None,
                                      front.ctypes.data if want_normal else None,
                                      uv.ctypes.data if want_uv else None))
        return dict(id=ids, t=t, normal=normal, front=front, uv=uv)

    def scatter_rays(self, rays, pixel0, sample, bounce, seed):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        out = np.zeros((rays.shape[0], 14), dtype=np.float32)
        _check(lib().rt_scatter_rays(self._h, rays.ctypes.data, rays.shape[0], pixel0, sample, bounce, seed, out.ctypes.data))
        return out


def tonemap_u8(accum, spp):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    rgb = np.empty(accum.shape, dtype=np.uint8)
    _check(lib().rt_tonemap_u8(accum.ctypes.data, accum.size // 3, spp, rgb.ctypes.data))
    return rgb


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _check(lib().rt_write_ppm(path.encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]))


def write_image(path, rgb):
    """Saves an 8-bit RGB image; the format follows the extension (.png or .ppm), as `image::save` does in main.rs:600."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _check(lib().rt_write_image(path.encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]))


def accum_save(path, accum, spp_done, seed):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    _check(lib().rt_accum_save(path.encode(), accum.ctypes.data, accum.shape[1], accum.shape[0], spp_done, seed))


def accum_load(path, height, width):
    """Returns (accum, spp_done, seed) of a checkpoint written by accum_save."""
    accum = np.zeros((height, width, 3), dtype=np.float32)
    spp, seed = C.c_int64(0), C.c_uint64(0)
    _check(lib().rt_accum_load(path.encode(), accum.ctypes.data, width, height, C.byref(spp), C.byref(seed)))
    return accum, spp.value, seed.value
