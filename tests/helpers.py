"""Shared test plumbing: ctypes views of the CPU oracle (oracle/librt_oracle.so) and of the test-only
host emulation of the device functions (tests/emul/librt_emul.so), plus helpers that build the same
scene on every side.  Nothing here is imported by the ray_tracing_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

import ray_tracing_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "configs")
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "librt_oracle.so")
EMUL_SO = os.path.join(ROOT, "tests", "emul", "librt_emul.so")
SCENE_SEED = 0x5EED0001
RENDER_SEED = 0x5EED0002

_P = C.c_void_p
_D = C.POINTER(C.c_double)
_oracle = None
_emul = None


def _build_if_stale(target, sources, cmd, cwd):
    newest = max(os.path.getmtime(s) for s in sources)
    if not os.path.exists(target) or os.path.getmtime(target) < newest:
        subprocess.run(cmd, check=True, cwd=cwd)


def oracle():
    global _oracle
    if _oracle is None:
        _build_if_stale(ORACLE_SO, [os.path.join(ROOT, "oracle", "rt_oracle.cpp")], ["make", "-s"], os.path.join(ROOT, "oracle"))
        L = C.CDLL(ORACLE_SO)
        L.orc_scene_new.restype = _P
        L.orc_scene_new.argtypes = [C.c_uint64]
        L.orc_scene_free.argtypes = [_P]
        L.orc_scene_error.restype = C.c_char_p
        L.orc_scene_error.argtypes = [_P]
        L.orc_scene_prim_count.argtypes = [_P]
        L.orc_xi.restype = C.c_double
        L.orc_xi.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
        L.orc_scene_rng.restype = C.c_uint64
        L.orc_scene_rng.argtypes = [C.c_uint64, C.c_int, _D]
        L.orc_solid_color.argtypes = [_P, C.c_double, C.c_double, C.c_double]
        L.orc_checker.argtypes = [_P, C.c_int, C.c_int]
        L.orc_noise.argtypes = [_P, C.c_double, _D, C.POINTER(C.c_int32)]
        L.orc_lambertian.argtypes = [_P, C.c_int]
        L.orc_metal.argtypes = [_P, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_dielectric.argtypes = [_P, C.c_double]
        L.orc_diffuse_light.argtypes = [_P, C.c_int]
        L.orc_isotropic.argtypes = [_P, C.c_int]
        L.orc_sphere.argtypes = [_P, _D, C.c_double, C.c_int]
        L.orc_moving_sphere.argtypes = [_P, _D, _D, C.c_double, C.c_double, C.c_double, C.c_int]
        L.orc_rectangle.argtypes = [_P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.orc_box.argtypes = [_P, _D, _D, C.c_int]
        L.orc_list.argtypes = [_P, C.POINTER(C.c_int), C.c_int]
        L.orc_bvh_node.argtypes = [_P, C.c_int, C.c_double, C.c_double]
        L.orc_translate.argtypes = [_P, C.c_int, _D]
        L.orc_rotate_y.argtypes = [_P, C.c_int, C.c_double]
        L.orc_constant_medium.argtypes = [_P, C.c_int, C.c_double, C.c_int]
        L.orc_canned.argtypes = [_P, C.c_char_p, C.c_uint64]
        L.orc_set_world.argtypes = [_P, C.c_int, C.c_int]
        L.orc_world_bbox.argtypes = [_P, _D]
        L.orc_obj_bbox.argtypes = [_P, C.c_int, C.c_double, C.c_double, _D]
        L.orc_camera_new.restype = _P
        L.orc_camera_new.argtypes = [_D, _D, _D] + [C.c_double] * 6
        L.orc_camera_free.argtypes = [_P]
        L.orc_camera_get.argtypes = [_P, _D]
        L.orc_camera_ray.argtypes = [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64, _D]
        L.orc_hit.argtypes = [_P, _D, _D, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, _D]
        L.orc_scatter.argtypes = [_P, _D, _D, C.c_double, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, _D]
        L.orc_texture_value.argtypes = [_P, C.c_int, C.c_double, C.c_double, _D, _D]
        L.orc_sphere_uv.argtypes = [_D, _D]
        L.orc_reflectance.restype = C.c_double
        L.orc_reflectance.argtypes = [C.c_double, C.c_double]
        L.orc_refract.argtypes = [_D, _D, C.c_double, _D]
        L.orc_reflect.argtypes = [_D, _D, _D]
        L.orc_unit_vector.argtypes = [C.c_double, C.c_double, _D]
        L.orc_image_height.argtypes = [C.c_int, C.c_double]
        L.orc_tonemap.argtypes = [_D, C.c_int, C.c_int, _P]
        L.orc_trace_primary.argtypes = [_P, _P, C.c_int, C.c_int, C.c_uint32, C.c_uint64, _P, _P, _P]
        L.orc_render.argtypes = [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _D, C.c_uint64, C.c_int, _P, _P, _P]
        _oracle = L
    return _oracle


def emul():
    global _emul
    if _emul is None:
        csrc = os.path.join(ROOT, "ray_tracing_amd", "csrc")
        srcs = [os.path.join(ROOT, "tests", "emul", "rt_emul.cpp")] + [os.path.join(csrc, f) for f in
                                                                      ("rt_device.h", "rt_types.h", "frame.h", "scene_builder.cpp", "scene_builder.h", "config.cpp")]
        _build_if_stale(EMUL_SO, srcs,
                        ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wno-unknown-pragmas",
                         "-o", EMUL_SO, srcs[0], os.path.join(csrc, "scene_builder.cpp"), os.path.join(csrc, "config.cpp")],
                        os.path.join(ROOT, "tests", "emul"))
        L = C.CDLL(EMUL_SO)
        L.emul_trace_primary.argtypes = [_P, C.POINTER(rt.RtCamera), C.POINTER(rt.RtRenderParams), C.c_int32, _P, _P, _P, _P, _P]
        L.emul_render.argtypes = [_P, C.POINTER(rt.RtCamera), C.POINTER(rt.RtRenderParams), C.c_int, _P, _P, C.POINTER(C.c_uint64)]
        L.emul_scatter_rays.argtypes = [_P, _P, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, _P]
        L.emul_camera_dev.argtypes = [C.POINTER(rt.RtCamera), C.POINTER(C.c_float)]
        _emul = L
    return _emul


def d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


# ---------------------------------------------------------------------------------------------
# A scene written once and replayed into several builders.  `OracleBuilder` has the method names of
# ray_tracing_amd.Builder, so one Python function can populate both.
class OracleBuilder:
    def __init__(self, bvh_seed=SCENE_SEED):
        self.L = oracle()
        self.h = _P(self.L.orc_scene_new(bvh_seed))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_free(self.h)
            self.h = None

    def solid_color(self, c): return self.L.orc_solid_color(self.h, c[0], c[1], c[2])
    def checker(self, even, odd): return self.L.orc_checker(self.h, even, odd)
    def checker_from_solid_colors(self, c1, c2): return self.checker(self.solid_color(c1), self.solid_color(c2))

    def noise(self, scale, ranvec, perm):
        rv = np.ascontiguousarray(ranvec, dtype=np.float64).reshape(768)
        pm = np.ascontiguousarray(perm, dtype=np.int32).reshape(768)
        return self.L.orc_noise(self.h, scale, rv.ctypes.data_as(_D), pm.ctypes.data_as(C.POINTER(C.c_int32)))

    def lambertian_from_texture(self, tex): return self.L.orc_lambertian(self.h, tex)
    def lambertian(self, color): return self.lambertian_from_texture(self.solid_color(color))
    def metal(self, color, fuzz): return self.L.orc_metal(self.h, color[0], color[1], color[2], fuzz)
    def dielectric(self, ir): return self.L.orc_dielectric(self.h, ir)
    def diffuse_light(self, tex): return self.L.orc_diffuse_light(self.h, tex)
    def diffuse_light_from_color(self, color): return self.diffuse_light(self.solid_color(color))
    def isotropic(self, tex): return self.L.orc_isotropic(self.h, tex)
    def sphere(self, center, radius, mat): return self.L.orc_sphere(self.h, d3(center), radius, mat)

    def moving_sphere(self, c0, c1, t0, t1, radius, mat):
        return self.L.orc_moving_sphere(self.h, d3(c0), d3(c1), t0, t1, radius, mat)

    def rectangle(self, mat, x0, x1, y0, y1, k, rtype): return self.L.orc_rectangle(self.h, mat, x0, x1, y0, y1, k, rtype)
    def box(self, mn, mx, mat): return self.L.orc_box(self.h, d3(mn), d3(mx), mat)

    def list(self, children):
        arr = (C.c_int * len(children))(*children)
        return self.L.orc_list(self.h, arr, len(children))

    def bvh_node(self, lst, t0=0.0, t1=1.0): return self.L.orc_bvh_node(self.h, lst, t0, t1)
    def translate(self, obj, off): return self.L.orc_translate(self.h, obj, d3(off))
    def rotate_y(self, obj, deg): return self.L.orc_rotate_y(self.h, obj, deg)
    def constant_medium(self, boundary, density, tex): return self.L.orc_constant_medium(self.h, boundary, density, tex)

    def constant_medium_from_color(self, boundary, density, color):
        return self.constant_medium(boundary, density, self.solid_color(color))

    def canned_scene(self, scene_type, scene_seed):
        w = self.L.orc_canned(self.h, scene_type.encode(), scene_seed)
        if w < 0:
            raise ValueError("unknown scene type %s" % scene_type)
        return w

    def object_count(self): return self.L.orc_scene_prim_count(self.h)

    def set_world(self, world, use_bvh=True):
        if self.L.orc_set_world(self.h, world, 1 if use_bvh else 0) != 0:
            raise RuntimeError(self.L.orc_scene_error(self.h).decode())
        return self


class OracleCamera:
    def __init__(self, cam):
        self.L = oracle()
        self.h = _P(self.L.orc_camera_new(d3(cam.look_from), d3(cam.look_at), d3(cam.v_up), cam.vertical_fov_deg, cam.aspect_ratio,
                                          cam.aperture, cam.focus_dist, cam.shutter_open, cam.shutter_close))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_camera_free(self.h)
            self.h = None

    def fields(self):
        out = (C.c_double * 22)()
        self.L.orc_camera_get(self.h, out)
        a = np.array(out[:21]).reshape(7, 3)
        return dict(origin=a[0], llc=a[1], horizontal=a[2], vertical=a[3], u=a[4], v=a[5], w=a[6], lens_radius=out[21])


def oracle_trace_primary(ob, ocam, width, height, sample, seed, want_normal=False):
    n = width * height
    ids = np.empty(n, dtype=np.int32)
    t = np.empty(n, dtype=np.float64)
    normal = np.empty((n, 3), dtype=np.float64) if want_normal else None
    oracle().orc_trace_primary(ob.h, ocam.h, width, height, sample, seed, ids.ctypes.data, t.ctypes.data,
                               normal.ctypes.data if want_normal else None)
    return dict(id=ids, t=t, normal=normal)


def oracle_render(ob, ocam, width, height, spp, max_depth, seed, sample_offset=0, background=(0, 0, 0), n_threads=8, want_sumsq=False):
    accum = np.zeros((height, width, 3), dtype=np.float64)
    sumsq = np.zeros((height, width, 3), dtype=np.float64) if want_sumsq else None
    stats = np.zeros(14, dtype=np.uint64)
    oracle().orc_render(ob.h, ocam.h, width, height, spp, sample_offset, max_depth, d3(background), seed, n_threads,
                        accum.ctypes.data, sumsq.ctypes.data if want_sumsq else None, stats.ctypes.data)
    names = ["rays", "box", "sph", "msph", "rect", "inst", "med", "perlin", "lambertian", "metal", "dielectric", "light", "isotropic"]
    return accum, sumsq, dict(zip(names, [int(x) for x in stats[:13]]))


class FlatScene:
    """Host-only flatten of a ray_tracing_amd.Builder tree (no GPU): the RtSceneView the kernels would read."""

    def __init__(self, builder, world, flags=0):
        self.h = _P()
        rc = rt.lib().rt_flat_create_ex(builder._h, world, flags, C.byref(self.h))
        if rc != 0:
            raise rt.RtError(rc, rt.lib().rt_last_error().decode())
        self.view = rt.lib().rt_flat_view(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            rt.lib().rt_flat_destroy(self.h)
            self.h = None

    def info(self):
        info = rt.RtSceneInfo()
        rt.lib().rt_flat_info(self.h, C.byref(info))
        return info

    def set_spheres(self, ids, xyzr): rt._set_prims(rt.lib().rt_flat_set_spheres, self.h, ids, xyzr, 4)
    def set_moving_spheres(self, ids, values): rt._set_prims(rt.lib().rt_flat_set_moving_spheres, self.h, ids, values, 9)

    def refit(self):
        rt._check(rt.lib().rt_flat_refit(self.h))
        self.view = rt.lib().rt_flat_view(self.h)


def emul_trace_primary(flat, cam, params, sample, want_normal=False, want_uv=False):
    n = params.width * params.height
    ids = np.empty(n, dtype=np.int32)
    t = np.empty(n, dtype=np.float32)
    normal = np.empty((n, 3), dtype=np.float32) if want_normal else None
    front = np.empty(n, dtype=np.int32) if want_normal else None
    uv = np.empty((n, 2), dtype=np.float32) if want_uv else None
    emul().emul_trace_primary(flat.view, C.byref(cam), C.byref(params), sample, ids.ctypes.data, t.ctypes.data,
                              normal.ctypes.data if want_normal else None, front.ctypes.data if want_normal else None,
                              uv.ctypes.data if want_uv else None)
    return dict(id=ids, t=t, normal=normal, front=front, uv=uv)


def emul_render(flat, cam, params, n_threads=8, want_sumsq=False):
    accum = np.zeros((params.height, params.width, 3), dtype=np.float32)
    sumsq = np.zeros((params.height, params.width, 3), dtype=np.float32) if want_sumsq else None
    rays = C.c_uint64(0)
    emul().emul_render(flat.view, C.byref(cam), C.byref(params), n_threads, accum.ctypes.data,
                       sumsq.ctypes.data if want_sumsq else None, C.byref(rays))
    return accum, sumsq, rays.value


def config_camera(name):
    cfg = rt.load_config(os.path.join(CONFIGS, name))
    return cfg, rt.camera_from_config(cfg)


class SplitMix:
    """The scene-construction stream (SplitMix64), restated for the YAML mirror below."""
    M = (1 << 64) - 1

    def __init__(self, seed):
        self.s = seed & self.M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        return z ^ (z >> 31)

    def gen(self): return (self.next() >> 11) * (1.0 / 9007199254740992.0)
    def range(self, a, b): return a + (b - a) * self.gen()
    def upto(self, i): return self.next() % (i + 1)


def yaml_list(ob, items, rng):
    """Builds the `objects` of a List scene on the oracle from the PyYAML document: an independent reading of the schema
    that ray_tracing_amd/csrc/config.cpp implements (the reference's sphere items plus the typed items)."""
    v = lambda d: (d["x"], d["y"], d["z"])

    def texture(t):
        if "type" not in t:
            return ob.solid_color(v(t))
        if t["type"] == "SolidColor":
            return ob.solid_color(v(t["color"]))
        if t["type"] == "Checker":
            e = texture(t["even"]); o = texture(t["odd"])
            return ob.checker(e, o)
        if t["type"] == "Noise":          # textures.rs:110-128: 256 vectors in [-1, 1)^3, then three Fisher-Yates permutations
            ranvec = [[rng.range(-1.0, 1.0) for _ in range(3)] for _ in range(256)]
            perms = []
            for _ in range(3):
                p = list(range(256))
                for i in range(255, 0, -1):
                    j = rng.upto(i); p[i], p[j] = p[j], p[i]
                perms.append(p)
            return ob.noise(t["scale"], ranvec, perms)
        raise ValueError(t["type"])

    def tex_or_color(m, key):
        return texture(m["texture"]) if "texture" in m else ob.solid_color(v(m[key]))

    def material(m):
        t = m["type"]
        if t == "Lambertian": return ob.lambertian_from_texture(tex_or_color(m, "albedo"))
        if t == "Metal": return ob.metal(v(m["albedo"]), m["fuzz"])
        if t == "Dielectric": return ob.dielectric(m["ir"])
        if t == "DiffuseLight": return ob.diffuse_light(tex_or_color(m, "color"))
        if t == "Isotropic": return ob.isotropic(tex_or_color(m, "albedo"))
        raise ValueError(t)

    def obj(o):
        t = o.get("type", "Sphere")
        if t == "Sphere": return ob.sphere(v(o["center"]), o["radius"], material(o["material"]))
        if t == "MovingSphere":
            return ob.moving_sphere(v(o["center0"]), v(o["center1"]), o["time0"], o["time1"], o["radius"], material(o["material"]))
        if t == "Rectangle":
            return ob.rectangle(material(o["material"]), o["x0"], o["x1"], o["y0"], o["y1"], o["k"], {"Xy": 0, "Xz": 1, "Yz": 2}[o["plane"]])
        if t == "Box": return ob.box(v(o["min"]), v(o["max"]), material(o["material"]))
        if t == "Translate": return ob.translate(obj(o["object"]), v(o["offset"]))
        if t == "RotateY": return ob.rotate_y(obj(o["object"]), o["angle_deg"])
        if t == "ConstantMedium":
            bnd = obj(o["boundary"])
            return ob.constant_medium(bnd, o["density"], texture(o["albedo"]))
        if t == "List": return yaml_list(ob, o["objects"], rng)
        raise ValueError(t)
    return ob.list([obj(o) for o in items])


def scene_pair(name, scene_seed=SCENE_SEED):
    """Builds configs/<name> on the product builder and the same scene on the oracle.  Returns (cfg, cam, builder, world, oracle_builder)."""
    import yaml
    path = os.path.join(CONFIGS, name)
    cfg, cam = config_camera(name)
    b = rt.Builder()
    world = b.scene_from_config(path, scene_seed)
    ob = OracleBuilder(scene_seed)
    stype = cfg.scene_type.decode()
    if stype == "List":
        doc = yaml.safe_load(open(path))
        objs = (doc.get("scene") or doc["objects"])["objects"]
        ow = yaml_list(ob, objs, SplitMix(scene_seed))
    else:
        ow = ob.canned_scene(stype, scene_seed)
    ob.set_world(ow, True)
    return cfg, cam, b, world, ob


def primary_agreement(gpu, orc):
    """Fraction of pixels with equal ids, and the worst relative t error on the agreeing hit pixels."""
    same = gpu["id"] == orc["id"]
    hit = same & (orc["id"] >= 0)
    rel = np.abs(gpu["t"][hit].astype(np.float64) - orc["t"][hit]) / np.abs(orc["t"][hit])
    return float(same.mean()), (float(rel.max()) if rel.size else 0.0)
