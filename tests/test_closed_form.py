"""Whole renders whose expected image follows in closed form from the reference's own formulas.  They need no second
renderer: the oracle (CPU, f64), the host emulation of the device arithmetic (CPU, fp32) and the HIP path are each held
against the same number.

  * Lambertian plane under a uniform sky: every path is plane -> sky, radiance = albedo * sky, no noise at all
    (materials.rs:77-90, lib.rs:63-76).
  * Lambertian floor under an emitting ceiling: radiance = albedo * emitted (materials.rs:243-249 ends the path).
  * Constant medium of thickness L seen head-on against a bright background, black phase-function albedo:
    radiance = background * 10^(-density * L)  -- base 10 because mediums.rs:64 takes log10 of the draw.
  * Fuzzy mirror (fuzz 1) seen at 60 degrees: a reflection survives when dot(scattered, normal) > 0 (materials.rs:139-144);
    with the normalised in-sphere sample of utils.rs:13-22 that happens with probability (1 + cos 60) / 2 = 0.75.
  * Glass ball seen along its axis with a light behind it: Schlick's R0 = 0.04 at both faces (materials.rs:174-209); the
    fraction of paths that leave towards the light is 0.96^2 / (1 - 0.04^2).
  * Emitting sphere of radius 0.5 that moves 4 units across the line of sight during the shutter (objects.rs:285-289,
    cameras.rs:82-98): it covers the line for 2 * 0.5 / 4 of the shutter time.
  * White medium (isotropic albedo 1, mediums.rs:39-87 and materials.rs:276-283) in a glassless ball under a uniform
    background: whatever the path does, it ends in the background with throughput 1.
  * Emitting unit cube turned 30 degrees about y and moved aside (transrot.rs:34-38, 92-116), seen from 1000 units away in a
    4 x 4 unit field of view: its silhouette is (cos 30 + sin 30) x 1 units of the 16.
"""
import math

import numpy as np
import pytest

import helpers
import ray_tracing_amd as rt


def lambert_plane(b):
    return b.list([b.rectangle(b.lambertian((0.3, 0.5, 0.7)), -1000, 1000, -1000, 1000, 0.0, rt.RECT_XZ)])


def lit_room(b):
    return b.list([b.rectangle(b.lambertian((0.3, 0.5, 0.7)), -1000, 1000, -1000, 1000, 0.0, rt.RECT_XZ),
                   b.rectangle(b.diffuse_light_from_color((4.0, 2.0, 1.0)), -1000, 1000, -1000, 1000, 1.0, rt.RECT_XZ)])


def fog_slab(b):
    boundary = b.box((-50, -50, -0.5), (50, 50, 0.5), b.lambertian((1, 1, 1)))
    return b.list([b.constant_medium_from_color(boundary, 0.5, (0.0, 0.0, 0.0))])


def fuzzy_mirror(b):
    return b.list([b.rectangle(b.metal((0.8, 0.6, 0.4), 1.0), -1000, 1000, -1000, 1000, 0.0, rt.RECT_XZ)])


def glass_ball(b):
    return b.list([b.sphere((0, 0, 0), 1.0, b.dielectric(1.5)),
                   b.rectangle(b.diffuse_light_from_color((2.0, 2.0, 2.0)), -0.5, 0.5, -0.5, 0.5, -3.0, rt.RECT_XY)])


def moving_light(b):
    return b.list([b.moving_sphere((-2, 0, 0), (2, 0, 0), 0.0, 1.0, 0.5, b.diffuse_light_from_color((3.0, 3.0, 3.0)))])


def white_medium(b):
    return b.list([b.constant_medium_from_color(b.sphere((0, 0, 0), 1.0, b.lambertian((1, 1, 1))), 1.0, (1.0, 1.0, 1.0))])


def turned_cube(b):
    cube = b.box((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), b.diffuse_light_from_color((5.0, 5.0, 5.0)))
    return b.list([b.translate(b.rotate_y(cube, 30.0), (0.3, 0.2, 0.0))])


# name: (scene, camera, background, width, height, spp, expected radiance, relative tolerance of the image mean)
T_GLASS = 0.96 * 0.96 / (1.0 - 0.04 * 0.04)
CASES = {
    "lambert_plane": (lambert_plane, dict(look_from=(0, 3, 4), look_at=(0, 0, 0), vfov=30.0), (1.0, 0.9, 0.8), 24, 16, 8,
                      (0.3 * 1.0, 0.5 * 0.9, 0.7 * 0.8), 2e-6),
    "lit_room": (lit_room, dict(look_from=(0, 0.5, 0), look_at=(0.3, 0, 0.3), vfov=20.0), (0, 0, 0), 16, 16, 64,
                 (0.3 * 4.0, 0.5 * 2.0, 0.7 * 1.0), 1e-3),
    "fog_slab": (fog_slab, dict(look_from=(0, 0, 10), look_at=(0, 0, 0), vfov=1.0), (1, 1, 1), 64, 64, 64,
                 (10 ** -0.5,) * 3, 5 * math.sqrt((1 - 10 ** -0.5) / 10 ** -0.5 / (64 * 64 * 64)) + 2e-4),
    "fuzzy_mirror": (fuzzy_mirror, dict(look_from=(0, 1, math.sqrt(3.0)), look_at=(0, 0, 0), vfov=0.2), (1, 1, 1), 64, 64, 32,
                     (0.8 * 0.75, 0.6 * 0.75, 0.4 * 0.75), 5 * math.sqrt(0.25 / 0.75 / (64 * 64 * 32)) + 1e-3),
    "glass_ball": (glass_ball, dict(look_from=(0, 0, 5), look_at=(0, 0, 0), vfov=0.2), (0, 0, 0), 32, 32, 128,
                   (2.0 * T_GLASS,) * 3, 5 * math.sqrt((1 - T_GLASS) / T_GLASS / (32 * 32 * 128)) + 2e-4),
    "moving_light": (moving_light, dict(look_from=(0, 0, 8), look_at=(0, 0, 0), vfov=0.05), (0, 0, 0), 32, 32, 128,
                     (3.0 * 0.25,) * 3, 5 * math.sqrt(0.75 / 0.25 / (32 * 32 * 128)) + 2e-3),
    "white_medium": (white_medium, dict(look_from=(0, 0, 5), look_at=(0, 0, 0), vfov=20.0), (0.9, 0.8, 0.7), 32, 32, 16,
                     (0.9, 0.8, 0.7), 1e-4),
    "turned_cube": (turned_cube, dict(look_from=(0, 0, 1000), look_at=(0, 0, 0), vfov=2 * math.degrees(math.atan(2.0 / 1000.0))), (0, 0, 0), 128, 128, 16,
                    (5.0 * (math.cos(math.radians(30)) + 0.5) / 16.0,) * 3,
                    5 * math.sqrt((1 - 1.366 / 16) / (1.366 / 16) / (128 * 128 * 16)) + 2e-3),
}
SEED = 0xC105ED


def camera(spec, width, height):
    return rt.make_camera(spec["look_from"], spec["look_at"], (0, 1, 0), spec["vfov"], width / height, 0.0, 1.0)


def check(name, mean_radiance):
    expected, tol = np.array(CASES[name][6]), CASES[name][7]
    rel = np.abs(mean_radiance - expected) / expected
    assert np.all(rel <= tol), (name, mean_radiance, expected, rel, tol)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_closed_form(name):
    build, cspec, bg, w, h, spp, _, _ = CASES[name]
    ob = helpers.OracleBuilder()
    ob.set_world(build(ob))
    accum, _, _ = helpers.oracle_render(ob, helpers.OracleCamera(camera(cspec, w, h)), w, h, spp, 50, SEED, background=bg)
    check(name, accum.reshape(-1, 3).mean(axis=0) / spp)
    if name == "lambert_plane":   # no randomness reaches the pixel value: every pixel, not just the mean
        assert np.allclose(accum / spp, np.array(CASES[name][6]), rtol=1e-12)


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_arithmetic_on_host_matches_closed_form(name):
    build, cspec, bg, w, h, spp, _, _ = CASES[name]
    b = rt.Builder()
    flat = helpers.FlatScene(b, build(b))
    accum, _, _ = helpers.emul_render(flat, camera(cspec, w, h), rt.render_params(w, h, spp, 50, SEED, background=bg))
    check(name, accum.reshape(-1, 3).astype(np.float64).mean(axis=0) / spp)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_closed_form(gpu_required, name):
    build, cspec, bg, w, h, spp, _, _ = CASES[name]
    b = rt.Builder()
    scene = rt.Scene(b, build(b), 0)
    accum, _, st = scene.render(camera(cspec, w, h), rt.render_params(w, h, spp, 50, SEED, background=bg))
    check(name, accum.reshape(-1, 3).astype(np.float64).mean(axis=0) / spp)
    if name == "lambert_plane":
        assert np.allclose(accum / spp, np.array(CASES[name][6], np.float32), rtol=2e-6) and st.rays == 2 * w * h * spp
