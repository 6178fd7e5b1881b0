"""Host logic that needs no GPU: the YAML-subset config reader (main.rs:453-494), the canned scenes'
construction order, and the C ABI surface (every symbol include/rt_hip.h declares must be exported)."""
import os
import re
import tempfile

import numpy as np
import pytest

import ray_tracing_amd as rt
from helpers import CONFIGS, ROOT, SCENE_SEED, OracleBuilder, scene_pair

# The layout the reference's own config files use (block mappings, one key per line).
BLOCK_STYLE = """image_width: 600
samples_per_pixel: 200
max_depth: 50
camera:
  look_from:
    x: 278.0
    y: 278.0
    z: -800.0
  look_at:
    x: 278.0
    y: 278.0
    z: 0.0
  v_up:
    x: 0.0
    y: 1.0
    z: 0.0
  vertical_fov_deg: 40.0
  aspect_ratio: 1.0
  aperture: 0.1
  focus_distance: 10.0
scene:
  type: CornellSmoke
"""
LEGACY = """image_width: 640
samples_per_pixel: 50
max_depth: 20
camera:
  look_from: {x: 0.0, y: 0.0, z: 10.0}
  look_at: {x: 0.0, y: 0.0, z: 0.0}
  v_up: {x: 0.0, y: 1.0, z: 0.0}
  vertical_fov_deg: 20.0
  aspect_ratio: 1.25
  aperture: 0.1
  focus_distance: 10.0
objects:
  objects:
  - center:
      x: 0.0
      y: -100.5
      z: -1.0
    radius: 100.0
    material:
      type: Lambertian
      albedo:
        x: 0.8
        y: 0.8
        z: 0.0
  # a comment between items
  - center:
      x: -1.0
      y: 0.0
      z: -1.0
    radius: -0.4
    material:
      type: Dielectric
      ir: 1.5
  - center:
      x: 1.0
      y: 0.0
      z: -1.0
    radius: 0.5
    material:
      type: Metal
      albedo:
        x: 0.8
        y: 0.6
        z: 0.2
      fuzz: 0.2
"""


def load_text(text):
    with tempfile.NamedTemporaryFile("w", suffix=".yml", delete=False) as f:
        f.write(text)
    try:
        return rt.load_config(f.name), f.name
    finally:
        pass


def test_block_style_config():
    cfg, path = load_text(BLOCK_STYLE)
    assert (cfg.image_width, cfg.image_height, cfg.samples_per_pixel, cfg.max_depth) == (600, 600, 200, 50)
    assert list(cfg.look_from) == [278.0, 278.0, -800.0] and list(cfg.look_at) == [278.0, 278.0, 0.0] and list(cfg.v_up) == [0, 1, 0]
    assert (cfg.vertical_fov_deg, cfg.aspect_ratio, cfg.aperture, cfg.focus_distance) == (40.0, 1.0, 0.1, 10.0)
    assert cfg.scene_type == b"CornellSmoke"
    os.unlink(path)


def test_legacy_objects_schema_is_a_list_scene():
    cfg, path = load_text(LEGACY)
    assert cfg.scene_type == b"List" and cfg.n_list_objects == 3 and cfg.image_height == 512
    b = rt.Builder()
    w = b.scene_from_config(path, SCENE_SEED)
    info = b.flatten_info(w)
    assert info.n_spheres == 3 and info.n_objects == 3 and info.n_materials == 3
    os.unlink(path)


def test_defaults_and_errors():
    # main.rs:468-480: width 400, spp 500, depth 50 when absent
    body = BLOCK_STYLE.split("camera:")[1]
    cfg, path = load_text("camera:" + body)
    assert (cfg.image_width, cfg.samples_per_pixel, cfg.max_depth, cfg.image_height) == (400, 500, 50, 400)
    os.unlink(path)
    for bad, why in [(BLOCK_STYLE.replace("  aperture: 0.1\n", ""), "camera"),              # cameras.rs:10-18: all 7 required
                     (BLOCK_STYLE.replace("CornellSmoke", "Teapot"), "unknown scene type"),
                     (BLOCK_STYLE.replace("scene:\n  type: CornellSmoke\n", ""), "scene"),
                     (BLOCK_STYLE.replace("image_width: 600", "image_width: -3"), "image_width"),
                     (BLOCK_STYLE.replace("  look_at:", "   look_at:"), "Could not read values")]:
        with pytest.raises(rt.RtError) as e:
            load_text(bad)
        assert e.value.code == -7 and "Could not read values" in str(e.value) and why in str(e.value)
    with pytest.raises(rt.RtError) as e:
        rt.load_config("/nonexistent/cfg.yml")
    assert "Could not open config file." in str(e.value)            # main.rs:511


def test_shipped_configs_match_the_reference_digest():
    want = {"cfg.yml": (400, 320, 100, 50, b"TwoPerlinSpheres"), "rectangle.yml": (400, 320, 400, 50, b"SimpleLight"),
            "orig_img.yml": (640, 512, 50, 20, b"List"), "cornell.yml": (600, 600, 200, 50, b"CornellSmoke"),
            "final_scene.yml": (800, 800, 100000, 50, b"FinalScene")}
    for name, (w, h, spp, depth, scene) in want.items():
        c = rt.load_config(os.path.join(CONFIGS, name))
        assert (c.image_width, c.image_height, c.samples_per_pixel, c.max_depth, c.scene_type) == (w, h, spp, depth, scene)
        assert c.aperture == 0.1 and c.focus_distance == 10.0 and list(c.v_up) == [0, 1, 0]


def test_canned_scenes_assign_the_same_ids_as_the_oracle():
    counts = {"cfg.yml": 2, "rectangle.yml": 3, "two_spheres.yml": 2, "cornell_box.yml": 18, "cornell.yml": 20, "final_scene.yml": 3411}
    for name, n in counts.items():
        cfg, cam, b, world, ob = scene_pair(name)
        assert b.object_count() == ob.object_count() == n
    cfg, cam, b, world, ob = scene_pair("random.yml")
    assert b.object_count() == ob.object_count() and 400 < b.object_count() <= 488
    with pytest.raises(rt.RtError):
        rt.Builder().canned_scene("Teapot", 1)


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(rtb?_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 35
    lib = rt.lib()
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert names == set(rt.API), names ^ set(rt.API)                # the Python binding covers exactly the header


def test_builder_rejects_bad_handles():
    b = rt.Builder()
    for fn in (lambda: b.lambertian_from_texture(99), lambda: b.sphere((0, 0, 0), 1, 42), lambda: b.translate(7, (0, 0, 0)),
               lambda: b.checker(0, 1), lambda: b.constant_medium(3, 0.1, 0), lambda: b.bvh_node(b.list([]))):
        with pytest.raises(rt.RtError):
            fn()
    with pytest.raises(rt.RtError):
        b.rectangle(b.lambertian((1, 1, 1)), 0, 1, 0, 1, 0, 5)


def test_tonemap_matches_utils_rs():  # utils.rs:49-63
    spp = 16
    acc = np.array([[0.25 * spp, 1.0 * spp, 0.0], [4.0 * spp, -1.0, 0.5 * spp]], dtype=np.float32)
    rgb = rt.tonemap_u8(acc, spp)
    assert rgb.tolist() == [[128, 255, 0], [255, 0, int(256 * np.sqrt(0.5))]]


def test_jpeg_writer_decodes_to_the_image(tmp_path):
    """The reference saves to ./image.jpeg by default (main.rs:502, 600).  The baseline JPEG rt_write_image produces must decode
    (with an independent decoder) to the same picture within quantisation error, at sizes that are no multiple of 8 too."""
    Image = pytest.importorskip("PIL.Image")
    for w, h in [(64, 48), (37, 29), (1, 1), (8, 8)]:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1), (xx + yy) * 255 // max(w + h - 2, 1)], -1).astype(np.uint8)
        path = str(tmp_path / ("a%dx%d.jpeg" % (w, h)))
        rt.write_image(path, img)
        raw = open(path, "rb").read()
        assert raw[:2] == b"\xff\xd8" and raw[-2:] == b"\xff\xd9"
        got = np.asarray(Image.open(path).convert("RGB")).astype(int)
        assert got.shape == (h, w, 3)
        err = np.abs(got - img.astype(int))
        assert err.mean() < 3.0 and err.max() < 40, (w, h, err.mean(), err.max())
    with pytest.raises(rt.RtError):
        rt.write_image(str(tmp_path / "a.bmp"), np.zeros((4, 4, 3), np.uint8))


def test_png_and_ppm_writers_and_checkpoints(tmp_path):
    """rt_write_image follows the extension (main.rs:600); the PNG is decoded here with zlib.  Accumulator checkpoints round-trip."""
    import struct
    import zlib
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    png = str(tmp_path / "a.png")
    rt.write_image(png, rgb)
    data = open(png, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        if typ == b"IHDR": ihdr = struct.unpack(">IIBBBBB", body)
        if typ == b"IDAT": idat += body
        pos += 12 + n
    assert ihdr == (53, 37, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(37, 1 + 53 * 3)
    assert np.all(raw[:, 0] == 0) and np.array_equal(raw[:, 1:].reshape(37, 53, 3), rgb)
    ppm = str(tmp_path / "a.ppm")
    rt.write_image(ppm, rgb)
    assert open(ppm, "rb").read() == b"P6\n53 37\n255\n" + rgb.tobytes()
    with pytest.raises(rt.RtError) as e:
        rt.write_image(str(tmp_path / "a.tiff"), rgb)
    assert e.value.code == -6
    acc = rng.random((37, 53, 3), dtype=np.float32) * 100
    ck = str(tmp_path / "acc.bin")
    rt.accum_save(ck, acc, 1234, 0x5EED0002)
    got, spp, seed = rt.accum_load(ck, 37, 53)
    assert np.array_equal(got, acc) and spp == 1234 and seed == 0x5EED0002
    with pytest.raises(rt.RtError):
        rt.accum_load(ck, 53, 37)                      # a checkpoint of another image size is refused


def test_extended_list_schema(tmp_path):
    """Typed List items (every constructor of the prelude), optional background and shutter keys, and their error reports."""
    cfg = rt.load_config(os.path.join(CONFIGS, "showcase.yml"))
    assert cfg.scene_type == b"List" and cfg.n_list_objects == 10 and (cfg.image_width, cfg.image_height) == (480, 320)
    assert (cfg.shutter_open, cfg.shutter_close) == (0.0, 0.5) and np.allclose(list(cfg.background), [0.02, 0.03, 0.06])
    cam = rt.camera_from_config(cfg)
    assert cam.shutter_close == 0.5
    b = rt.Builder()
    info = b.flatten_info(b.scene_from_config(os.path.join(CONFIGS, "showcase.yml"), SCENE_SEED))
    # ground, noise, glass, metal, 2 small spheres + the medium's boundary sphere; a moving sphere; 2 rects; a box under RotateY + Translate
    assert (info.n_spheres, info.n_moving_spheres, info.n_rects, info.n_boxes, info.n_instances, info.n_media, info.n_perlin) == (7, 1, 2, 1, 1, 1, 1)
    old = rt.load_config(os.path.join(CONFIGS, "orig_img.yml"))           # the reference's own item form still reads as before
    assert old.n_list_objects == 5 and (old.shutter_open, old.shutter_close) == (0.0, 1.0) and list(old.background) == [0, 0, 0]
    head = "camera: {look_from: {x: 0, y: 0, z: 1}, look_at: {x: 0, y: 0, z: 0}, v_up: {x: 0, y: 1, z: 0}, vertical_fov_deg: 20, aspect_ratio: 1, aperture: 0, focus_distance: 1}\n"
    bad = {"unknown object type 'Torus'": "scene:\n  type: List\n  objects:\n    - {type: Torus, radius: 1}\n",
           "Rectangle: plane": "scene:\n  type: List\n  objects:\n    - {type: Rectangle, x0: 0, x1: 1, y0: 0, y1: 1, k: 0, material: {type: Dielectric, ir: 1.5}}\n",
           "unknown material type 'Plastic'": "scene:\n  type: List\n  objects:\n    - {center: {x: 0, y: 0, z: 0}, radius: 1, material: {type: Plastic}}\n",
           "background": "background: blue\nscene: {type: TwoSpheres}\n"}
    for msg, text in bad.items():
        path = tmp_path / "bad.yml"
        path.write_text(head + text)
        with pytest.raises(rt.RtError) as e:
            rt.load_config(str(path))
        assert e.value.code == -7 and "Could not read values." in str(e.value) and msg in str(e.value), (msg, str(e.value))


def test_no_kernel_uses_scratch_memory():
    """The library build records what the compiler reports per kernel (ray_tracing_amd/kernel_resources.json).  No kernel may spill
    registers or index a private array (both end up in scratch memory, a round trip to HBM per access), and the kernels that run one
    1024-thread block per CU must fit the 128 registers that leaves each lane."""
    import json
    from ray_tracing_amd import build
    assert os.path.exists(build.RESOURCES), "run python -m ray_tracing_amd.build"
    assert os.path.getmtime(build.RESOURCES) >= os.path.getmtime(build.LIB) - 120
    res = json.load(open(build.RESOURCES))
    names = " ".join(res)
    for k in ("k_raygen", "k_trace<", "k_shade<", "k_shade_lds<", "k_accumulate", "k_refit_pass", "k_lbvh_nodes"):
        assert k in names, k
    for name, r in res.items():
        if not name.startswith("k_"):      # rocPRIM's sort kernels (device BVH build) are the library's business
            continue
        assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs Spill"] == 0 and r["Dynamic Stack"] == "False", (name, r)
        if name.startswith("k_shade_lds<") or ("k_trace<" in name and ", 1024," in name):
            assert r["VGPRs"] + r["AGPRs"] <= 128, (name, r)
