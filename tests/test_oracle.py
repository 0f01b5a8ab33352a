"""CPU tests of the oracle (oracle/rt_oracle.c) against hand-derived known-answer vectors
(tests/golden/kats.json, written by tests/golden/make_kats.py from closed-form geometry) and
against its own invariants.  The reference ships no vectors, so these pin the checker."""
import json
import os

import numpy as np
import pytest

from raytracer_amd import scenes

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))


def one_sphere_scene(c, r):
    b = scenes._Builder()
    b.sphere(c, r, b.lambertian((0.5, 0.5, 0.5)))
    return b.build("kat", scenes.make_camera())


@pytest.mark.parametrize("k", KATS["hit"], ids=[k["name"] for k in KATS["hit"]])
def test_hit_kats(oracle64, k):
    scene = one_sphere_scene(k["c"], k["r"])
    ids, t, n, front, _ = oracle64.first_hit(scene, [k["o"]], [k["d"]], tmin=k["tmin"])
    assert (ids[0] == 0) == k["hit"]
    if k["hit"]:
        assert t[0] == pytest.approx(k["t"], rel=1e-7)
        assert np.allclose(n[0], k["n"], atol=1e-7)
        assert bool(front[0]) == k["front"]
        p = np.array(k["o"], float) + np.array(k["d"], float) * t[0]
        assert np.allclose(p, k["p"], atol=1e-6)


def test_closest_kats(oracle64):
    k = KATS["closest"]
    b = scenes._Builder()
    m = b.lambertian((0.5, 0.5, 0.5))
    for s in k["spheres"]:
        b.sphere(s[:3], s[3], m)
    scene = b.build("kat", scenes.make_camera())
    org = [r["o"] for r in k["rays"]]
    dirs = [r["d"] for r in k["rays"]]
    ids, t, *_ = oracle64.first_hit(scene, org, dirs, tmin=0.001)
    assert list(ids) == [r["id"] for r in k["rays"]]
    assert np.allclose(t, [r["t"] for r in k["rays"]], rtol=1e-12)


def _mat_array(k):
    from oracle.orc import MATERIAL_DTYPE

    m = np.zeros(1, MATERIAL_DTYPE)
    m["type"] = k["type"]
    m["albedo"] = k["albedo"]
    m["fuzz"] = k["fuzz"]
    m["ior"] = k["ior"]
    return m


@pytest.mark.parametrize("k", KATS["scatter"], ids=[k["name"] for k in KATS["scatter"]])
def test_scatter_kats(oracle64, k):
    att, oo, od, flag = oracle64.scatter(_mat_array(k), [0], [k["d"]], [k["p"]], [k["n"]], [k["front"]], [k["u"]])
    assert flag[0] == k["scattered"]
    if k["scattered"]:
        assert np.allclose(att[0], np.float32(k["albedo"]), atol=1e-7)
        assert np.allclose(oo[0], k["o"], atol=1e-9)  # 1e-4 is a float in the library
        assert np.allclose(od[0], k["dir"], atol=1e-7)


def test_tonemap_kats(oracle64):
    for k in KATS["tonemap"]:
        assert list(oracle64.convert_color(k["rgb"])) == k["rgba"]


def test_viewport_kats(oracle64):
    for k in KATS["viewport"]:
        cam = oracle64.update_viewport(k["center"], k["direction"])
        assert np.allclose(cam["right"][0], k["right"], atol=1e-7)
        assert np.allclose(cam["down"][0], k["down"], atol=1e-7)
        assert np.allclose(cam["upper_left"][0], k["upper_left"], atol=1e-6)
        py = scenes.make_camera(k["center"], k["direction"])
        for f in ("center", "right", "down", "upper_left"):
            assert np.array_equal(py[f], cam[f]), f  # the Python host mirror agrees bit for bit


def test_rng_kats(oracle64):
    for k in KATS["rng"]:
        assert list(oracle64.rng4(*k["key"])) == k["words"]


def test_sky_gradient(oracle64):
    """Empty scene: every sample is the sky lerp on the primary ray's y (reference src/Camera.cpp:155-156)."""
    empty = scenes._Builder().build("empty", scenes.make_camera())
    W, H = 9, 5
    org, dirs = oracle64.primary_rays(empty.camera, W, H, seed=4, smp=2, jitter=True)
    for pid in (0, 22, 44):
        a = 0.5 * (float(dirs[pid, 1]) + 1.0)
        want = np.array([1, 1, 1]) * (1 - a) + np.array([0.5, 0.7, 1.0]) * a
        got = oracle64.sample_radiance(empty, W, H, pid, 2, max_depth=5, seed=4)
        assert np.allclose(got, want, atol=1e-6)  # dirs were rounded to float
    # pixel-centre ray of the middle pixel points along +z exactly
    org, dirs = oracle64.primary_rays(empty.camera, W, H, jitter=False)
    assert np.allclose(dirs[22], [0, 0, 1], atol=1e-7)
    assert KATS["sky"]["rgb"] == [0.75, 0.85, 1.0]


def test_recursive_equals_iterative(oracle64):
    """samplesPerScatter = 1: the reference's recursion (src/Camera.cpp:137-157,180-195) and the
    throughput loop a GPU runs are the same function."""
    for scene, depth in ((scenes.small_scene(light=True), 10), (scenes.random_spheres_scene(), 50)):
        a, acc_a, rays_a = oracle64.render(scene, 64, 36, 4, depth, seed=3, iterative=False)
        b, acc_b, rays_b = oracle64.render(scene, 64, 36, 4, depth, seed=3, iterative=True)
        assert rays_a == rays_b
        assert np.allclose(acc_a, acc_b, rtol=1e-12, atol=1e-12)
        assert np.array_equal(a, b)
    assert oracle64.refract_clamped() == 0


def test_samples_per_scatter_has_same_expectation(oracle64):
    """Branching (samplesPerScatter = 3, the reference default) changes variance, not the mean."""
    scene = scenes.small_scene()
    _, acc1, rays1 = oracle64.render(scene, 32, 18, 256, 4, seed=1, samples_per_scatter=1)
    _, acc3, rays3 = oracle64.render(scene, 32, 18, 64, 4, seed=2, samples_per_scatter=3)
    m1 = acc1.mean(axis=(0, 1)) / 256
    m3 = acc3.mean(axis=(0, 1)) / 64
    assert np.allclose(m1, m3, rtol=0.02)
    assert rays3 > 2 * rays1 / 4  # the tree traces many more rays per sample


def test_render_is_thread_count_invariant(oracle64, monkeypatch):
    """Streams are keyed on the pixel id (the role of gm::setThreadSeed(pixelId), reference
    src/Camera.cpp:84), so the image cannot depend on the OpenMP schedule; rows rendered
    separately reassemble the frame."""
    scene = scenes.small_scene()
    full, acc, _ = oracle64.render(scene, 40, 24, 3, 10, seed=5)
    top, acc_t, _ = oracle64.render(scene, 40, 24, 3, 10, seed=5, y0=0, y1=10)
    bot, acc_b, _ = oracle64.render(scene, 40, 24, 3, 10, seed=5, y0=10, y1=24)
    assert np.array_equal(full[:10], top[:10]) and np.array_equal(full[10:], bot[10:])
    # sample_offset continues the sample sequence
    _, a1, _ = oracle64.render(scene, 40, 24, 2, 10, seed=5)
    _, a2, _ = oracle64.render(scene, 40, 24, 1, 10, seed=5, sample_offset=2)
    assert np.allclose(a1 + a2, acc, rtol=1e-12)


def test_single_precision_oracle_agrees(oracle64, oracle32):
    """The float build of the oracle renders the same image within noise: single precision by itself
    does not bias the estimate (what the GPU parity bounds rely on)."""
    assert oracle32.lib.orc_real_bytes() == 4 and oracle64.lib.orc_real_bytes() == 8
    scene = scenes.small_scene()
    a, acc_a, _ = oracle64.render(scene, 64, 36, 64, 10, seed=7)
    b, acc_b, _ = oracle32.render(scene, 64, 36, 64, 10, seed=7)
    assert np.allclose(acc_a.mean(axis=(0, 1)), acc_b.mean(axis=(0, 1)), rtol=5e-3)


def direct_light_scene(case):
    k = KATS["direct_light"]
    b = scenes._Builder()
    b.sphere(k["ground"]["c"], k["ground"]["r"], b.lambertian(k["ground"]["diffuse"]))
    if case["blocker"]:
        b.sphere(case["blocker"]["c"], case["blocker"]["r"], b.lambertian((0.5, 0.5, 0.5)))
    lt = case["light"]
    b.light(lt["position"], lt["ambient"], lt["diffuse"], (1, 1, 1), lt["view_pow"])
    return b.build("kat", scenes.make_camera(k["camera"]["center"], k["camera"]["direction"]))


@pytest.mark.parametrize("case", KATS["direct_light"]["cases"], ids=[c["name"] for c in KATS["direct_light"]["cases"]])
def test_direct_light_kats(oracle64, case):
    """Depth-1 render: the centre pixel of a fine frame is LDirect at P = (0,0,0), n = (0,1,0)."""
    scene = direct_light_scene(case)
    N = 401  # the jitter moves the hit point by < 1/401 of the viewport
    _, acc, rays = oracle64.render(scene, N, N, 1, 1, seed=1, y0=N // 2, y1=N // 2 + 1, direct_light=True)
    assert np.allclose(acc[N // 2, N // 2], case["rgb"], atol=0.02)
    assert rays == 2 * N  # one primary + one shadow query per pixel of the row (every ray hits the ground)


def prim_kat_scene(case):
    """The scene of one entry of KATS["prims"]."""
    b = scenes._Builder()
    m = b.lambertian((0.5, 0.5, 0.5))
    for spec in case["prims"]:
        kind = spec[0]
        if kind == "sphere":
            b.sphere(spec[1], spec[2], m)
        elif kind == "plane":
            b.plane(spec[1], spec[2], m)
        elif kind == "cube":
            b.cube(spec[1], spec[2], m)
        else:
            b.polygon(spec[1], m)
    for i in case["selected"]:
        b.select(i)
    scene = b.build("kat", scenes.make_camera())
    scene.needs_general = True
    return scene


def check_prim_kat(case, first_hit, atol):
    """first_hit(scene, org, dir, expanded_state) -> ids, t, normal, front, expanded"""
    scene = prim_kat_scene(case)
    for es in (0, 1):
        rays = [r for r in case["rays"] if r["es"] == es]
        if not rays:
            continue
        org = np.array([r["o"] for r in rays], dtype=np.float32)
        dirs = np.array([r["d"] for r in rays], dtype=np.float32)
        ids, t, nrm, front, exp = first_hit(scene, org, dirs, bool(es))
        for k, r in enumerate(rays):
            what = f'{case["name"]}: ray {r["o"]} -> {r["d"]} es={es}'
            assert ids[k] == r["id"], what
            if r["id"] < 0:
                continue
            assert abs(t[k] - r["t"]) <= atol * max(1.0, abs(r["t"])), what
            assert np.allclose(nrm[k], r["n"], atol=atol), what
            assert int(front[k]) == r["front"] and int(exp[k]) == r["exp"], what


@pytest.mark.parametrize("case", KATS["prims"], ids=[c["name"] for c in KATS["prims"]])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_primitive_and_halo_kats(oracle64, oracle32, case, prec):
    orc = oracle64 if prec == "f64" else oracle32

    def first_hit(scene, org, dirs, es):
        ids, t, nrm, front, _, exp = orc.first_hit(scene, org, dirs, tmin=0.001, general=True, expanded_state=es,
                                                   want_expanded=True)
        return ids, t, nrm, front, exp

    if prec == "f32" and "tie" in case["name"]:
        pytest.skip("an exact tie is a double-precision statement")
    check_prim_kat(case, first_hit, 1e-6 if prec == "f64" else 1e-5)
