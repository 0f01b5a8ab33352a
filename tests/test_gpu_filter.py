"""The matrix-core sphere filter against a double-precision evaluation of the same quantity.

The scan of scenes with more than 8 spheres evaluates, per (ray, sphere), a conservative filter on the
matrix cores and decides exactly only what passes.  A frame can only be right if the filter never drops
a sphere the ray's line meets; these tests check that property and the size of its error directly, on
the values the device code of the render kernel computes (rt_filter_values_host).  The reference has no
counterpart of the filter itself: it tests every sphere exactly (src/RayTracer.cpp:77-82,
inc/RTObjects.h:135-149)."""
import numpy as np
import pytest

from raytracer_amd import Renderer, RenderParams, scenes

pytestmark = pytest.mark.gpu


def _xyzr(scene):
    sp = scene.spheres
    return np.stack([sp["cx"], sp["cy"], sp["cz"], sp["r"]], axis=1).astype(np.float64)


def _rays(scene, n, rng, spread):
    """Origins: around the cloud, far from it, and exactly on sphere surfaces; directions: anywhere, at
    sphere centres, and grazing sphere rims."""
    c = _xyzr(scene)[:, :3]
    r = np.abs(_xyzr(scene)[:, 3])
    small = r < 50
    org = rng.normal(size=(n, 3)) * spread
    far = rng.random(n) < 0.15
    org[far] *= 60.0
    on = rng.random(n) < 0.3
    k = rng.integers(0, len(c), n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    org[on] = c[k[on]] + u[on] * r[k[on], None]
    d = rng.normal(size=(n, 3))
    aim = rng.random(n) < 0.6
    t = rng.integers(0, int(small.sum()), n)
    cs, rs = c[small][t], r[small][t]
    w = rng.normal(size=(n, 3))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    graze = rng.random(n) < 0.5
    target = cs + w * (rs * np.where(graze, 1.0 + rng.normal(size=n) * 1e-4, rng.random(n)))[:, None]
    d[aim] = (target - org)[aim]
    d *= (0.1 + 5.0 * rng.random(n))[:, None]  # unnormalised, as scattered rays are
    return org.astype(np.float32), d.astype(np.float32)


def _exact(scene, info, org, d):
    """disc / radius^2 for every (ray, sphere) in double, from the float inputs the device sees."""
    o = org.astype(np.float64)
    dd = d.astype(np.float64)
    # the unit direction the kernel uses is a float; its error is part of what the margin must cover
    dh = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    c = _xyzr(scene)[:, :3]
    r = _xyzr(scene)[:, 3]
    oc = o[:, None, :] - c[None, :, :]
    hb = np.einsum("rsk,rk->rs", oc, dh)
    perp2 = np.einsum("rsk,rsk->rs", oc, oc) - hb * hb
    disc = r[None, :] ** 2 - perp2
    R = float(info["radius"])
    oprime = (o - info["center"].astype(np.float64)[None, :]) / R
    m_ray = info["margin"] + info["margin_far"] * np.abs(oprime).max(axis=1)
    p = oprime - dh * np.einsum("rk,rk->r", oprime, dh)[:, None]
    return disc / (R * R), m_ray, np.linalg.norm(p, axis=1)


@pytest.mark.parametrize("name", ["random_spheres", "grid30"])
def test_filter_is_a_superset_and_its_error_is_within_the_margin(name):
    scene = scenes.by_name(name)
    rng = np.random.default_rng(7)
    with Renderer(0) as r:
        r.set_scene(scene)
        info = r.filter_describe()
        assert info["in_use"] and info["n_rows"] >= len(scene.spheres)
        org, d = _rays(scene, 4096, rng, spread=info["radius"])
        vals = r.filter_values(org, d).astype(np.float64)
    F, m_ray, pnorm = _exact(scene, info, org, d)
    n = len(scene.spheres)
    table = np.ones(n, dtype=bool)
    table[info["big"]] = False
    # rows of spheres outside the table, and the padding, never pass
    assert (vals[:, ~np.pad(table, (0, vals.shape[1] - n))] > 0).all()
    v = vals[:, :n][:, table]
    Ft = F[:, table]
    # 1. no sphere the line meets is dropped (the exact decision may also accept a discriminant that
    #    rounding made barely negative: allow for 1e-6 of the sphere's size, far inside the margin)
    meets = Ft >= -1e-6 * (np.abs(_xyzr(scene)[table, 3]) / info["radius"]) ** 2
    assert (v[meets] < 0).all(), f"{(v[meets] >= 0).sum()} hits dropped"
    # 2. the value is -(disc/R^2 + margin) to within half the margin, for lines that reach the cloud
    near = pnorm <= 1.0
    err = (np.abs(v + (Ft + m_ray[:, None])) / m_ray[:, None])[near]
    print(f"{name}: {meets.sum()} line-sphere meetings, largest error {err.max():.3f} of the ray's margin "
          f"(margin {info['margin']:.3e}), passing {100.0 * (v < 0).mean():.3f} % vs meeting {100.0 * (Ft >= 0).mean():.3f} %")
    assert err.max() <= 0.5
    # 3. a line that passes the whole cloud by meets nothing and passes nothing
    away = pnorm > 1.01
    assert not meets[away].any()
    assert (v[away] > 0).all()


def test_scenes_without_a_table_say_so():
    with Renderer(0) as r:
        r.set_scene(scenes.by_name("small"))  # four spheres: decided directly
        assert r.filter_describe()["n_rows"] == 0


def test_big_spheres_are_kept_out_of_the_frame():
    scene = scenes.by_name("random_spheres")
    with Renderer(0) as r:
        r.set_scene(scene)
        info = r.filter_describe()
    ground = int(np.argmax(np.abs(_xyzr(scene)[:, 3])))
    assert info["big"] == [ground]
    assert info["radius"] < 30.0  # the cloud of the small spheres, not the r = 1000 ground


def test_frames_are_the_same_bytes_with_either_filter(monkeypatch):
    """The filter only selects what the exact decision sees: scalar-load packed-FP32 filter and
    matrix-core filter render the same bytes and count the same rays."""
    scene = scenes.by_name("random_spheres")
    p = RenderParams(192, 108, 24, 50, seed=11)
    with Renderer(0) as r:
        r.set_scene(scene)
        assert r.filter_describe()["in_use"]
        a = r.render_host(p, want_accum=True)
    monkeypatch.setenv("RT_AMD_SPHERE_SOURCE", "scalar")
    with Renderer(0) as r:
        r.set_scene(scene)
        assert not r.filter_describe()["in_use"]
        b = r.render_host(p, want_accum=True)
    assert a[2] == b[2]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- scenes built to strain the filter's frame: where its centre, its scale and its table layout come from ----

def _cloud(name, n, seed, centre=(0.0, 0.0, 0.0), box=(10.0, 10.0, 3.0), radii=(0.05, 0.6), extra=(), more=None):
    """n spheres of mixed materials scattered in a box around `centre`, then `extra` (centre, radius) spheres;
    `more(builder, materials)` adds whatever else the scene shall hold."""
    rng = np.random.default_rng(seed)
    b = scenes._Builder()
    mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5),
            b.lambertian((0.2, 0.5, 0.7))]
    c0 = np.asarray(centre, dtype=np.float64)
    for i in range(n):
        c = c0 + (rng.random(3) - 0.5) * 2.0 * np.asarray(box)
        r = radii[0] * (radii[1] / radii[0]) ** rng.random()  # log-uniform
        b.sphere(c, r, mats[i % len(mats)])
    for c, r in extra:
        b.sphere(np.asarray(c, dtype=np.float64) + c0, r, mats[0])
    if more is not None:
        more(b, mats)
    eye = c0 + np.array([1.6 * box[0], -1.3 * box[1], 2.5 * box[2] + 2.0])
    return b.build(name, scenes.make_camera(eye, c0 - eye))


def _strain_scenes():
    return {
        # the frame's centre is rounded to float, and ray origins sit far from the world origin
        "offset": _cloud("offset", 300, 1, centre=(5000.0, -3000.0, 2000.0)),
        # three decades of radii, some of them negative (the reference's hitDetail only ever squares the radius)
        "radii": _cloud("radii", 200, 2, radii=(0.003, 3.0),
                        extra=[((1.0, 2.0, 0.5), -0.7), ((-3.0, 1.0, 0.0), -0.05)]),
        # more very large spheres than the filter decides outside its table: the frame has to span them
        "many_big": _cloud("many_big", 120, 3, extra=[((40.0 * np.cos(k), 40.0 * np.sin(k), -300.0 - 20.0 * k), 300.0 + 10.0 * k)
                                                       for k in range(12)]),
        # nested shells around one point and a row of spheres on a line: degenerate clouds
        "shells": _cloud("shells", 0, 4, extra=[((0.0, 0.0, 0.0), 0.1 * (k + 1)) for k in range(20)]),
        "row": _cloud("row", 0, 5, extra=[((0.9 * k - 18.0, 0.0, 0.0), 0.4) for k in range(40)]),
        # the smallest scene that gets a table and one streamed through LDS in tiles
        "sixteen": _cloud("sixteen", 16, 6),
        "streamed": _cloud("streamed", 2600, 7, box=(30.0, 30.0, 4.0), radii=(0.1, 0.5)),
        # the largest table that stays resident in LDS (45 blocks of 32) and the first that does not
        "edge_resident": _cloud("edge_resident", 1440, 8, box=(25.0, 25.0, 4.0), radii=(0.1, 0.5)),
        "edge_streamed": _cloud("edge_streamed", 1441, 8, box=(25.0, 25.0, 4.0), radii=(0.1, 0.5)),
    }


@pytest.mark.parametrize("name", ["offset", "radii", "many_big", "shells", "row", "sixteen", "streamed", "edge_resident",
                                  "edge_streamed"])
def test_strained_frames_drop_no_hit(name):
    """The superset property on scenes that stress how the filter's frame is chosen, with origins up to 10^4
    cloud radii away; and a frame rendered with either filter has the same bytes and ray count."""
    scene = _strain_scenes()[name]
    rng = np.random.default_rng(11)
    p = RenderParams(96, 64, 16, 12, seed=5)
    with Renderer(0) as r:
        r.set_scene(scene)
        info = r.filter_describe()
        assert info["in_use"] and info["n_rows"] >= len(scene.spheres) - len(info["big"])
        if name.startswith("edge"):
            assert not info["big"] and info["n_rows"] == (1440 if name == "edge_resident" else 1920)  # whole tiles of 20 blocks
        org, d = _rays(scene, 4096, rng, spread=info["radius"])
        # a few origins very far out, aimed back at the cloud
        far = rng.random(len(org)) < 0.05
        org[far] = (info["center"] + (org[far] - info["center"]) * 1.0e4 / 60.0).astype(np.float32)
        d[far] = (info["center"] + rng.normal(size=(far.sum(), 3)) * 0.3 * info["radius"] - org[far]).astype(np.float32)
        vals = r.filter_values(org, d).astype(np.float64)
        a = r.render_host(p, want_accum=True)
    F, m_ray, pnorm = _exact(scene, info, org, d)
    n = len(scene.spheres)
    table = np.ones(n, dtype=bool)
    table[info["big"]] = False
    v = vals[:, :n][:, table]
    Ft = F[:, table]
    # a double evaluation from the same float inputs; what the exact decision can still accept lies within
    # rounding of zero, scaled by the size of the terms that cancel (|O'|^2 for far origins)
    oprime = np.abs((org.astype(np.float64) - info["center"][None, :]) / info["radius"]).max(axis=1)
    slack = 1e-6 * (np.abs(_xyzr(scene)[table, 3]) / info["radius"])[None, :] ** 2
    meets = Ft >= -slack
    dropped = meets & ~(v < 0)
    assert not dropped.any(), f"{dropped.sum()} hits dropped, e.g. ray {np.argwhere(dropped)[0]}, |O'| {oprime[np.argwhere(dropped)[0][0]]:.3g}"
    assert (vals[:, ~np.pad(table, (0, vals.shape[1] - n))] > 0).all()
    near = (pnorm <= 1.0) & (oprime <= 100.0)
    err = (np.abs(v + (Ft + m_ray[:, None])) / m_ray[:, None])[near]
    print(f"{name}: {n} spheres, {len(info['big'])} outside the table, {meets.sum()} meetings, largest error "
          f"{err.max():.3f} of the margin, passing {100.0 * (v < 0).mean():.3f} % vs meeting {100.0 * (Ft >= 0).mean():.3f} %")
    assert err.max() <= 0.75

    import os
    old = os.environ.get("RT_AMD_SPHERE_SOURCE")
    os.environ["RT_AMD_SPHERE_SOURCE"] = "scalar"
    try:
        with Renderer(0) as r:
            r.set_scene(scene)
            assert not r.filter_describe()["in_use"]
            b = r.render_host(p, want_accum=True)
    finally:
        if old is None:
            del os.environ["RT_AMD_SPHERE_SOURCE"]
        else:
            os.environ["RT_AMD_SPHERE_SOURCE"] = old
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("name", ["random_spheres", "offset", "radii", "streamed"])
def test_plain_first_hit_runs_the_matrix_core_scan_and_agrees_with_double(name, oracle64):
    """SceneManager::hitClosest (reference src/RayTracer.cpp:68-82) for rays of every kind -- aimed, grazing, starting
    on surfaces, origins far out -- through the scan the render kernel runs (matrix-core filter, queue, exact
    single-precision decision): the object id may differ from the double-precision oracle's only where the
    oracle itself reports a marginal decision, and never because the filter dropped a sphere."""
    scene = scenes.by_name(name) if name == "random_spheres" else _strain_scenes()[name]
    rng = np.random.default_rng(23)
    with Renderer(0) as r:
        r.set_scene(scene)
        info = r.filter_describe()
        assert info["in_use"]
        org, d = _rays(scene, 30000, rng, spread=info["radius"])
        far = rng.random(len(org)) < 0.05
        org[far] = (info["center"] + (org[far] - info["center"]) * 100.0).astype(np.float32)
        d[far] = (info["center"] + rng.normal(size=(far.sum(), 3)) * 0.3 * info["radius"] - org[far]).astype(np.float32)
        ids, t, nrm, front = r.first_hit(org, d, tmin=np.float32(0.001), refine=False)
        rid, rt, _, _ = r.first_hit(org, d, tmin=np.float32(0.001), refine=True)
    oid, ot, onrm, ofront, margin = oracle64.first_hit(scene, org, d, tmin=float(np.float32(0.001)), want_margin=True)
    assert np.array_equal(rid, oid)
    bad = ids != oid
    print(f"{name}: {len(oid)} rays, {int((oid >= 0).sum())} hits, {int(bad.sum())} ids differ from the double evaluation, "
          f"largest margin among them {margin[bad].max() if bad.any() else 0.0:.2e}")
    assert np.all(margin[bad] < 1e-5), f"unambiguous rays disagree: margins {np.sort(margin[bad])[-5:]}"
    ok = (~bad) & (oid >= 0) & (margin >= 1e-3)
    assert ok.sum() > 1000
    # t: relative 1e-4, plus the resolution of single-precision positions at the scale of the ray and of the sphere it
    # hits (the r = 1000 ground sphere is known to 6e-5), which grows like 1 / sqrt(margin) towards grazing incidence
    dlen = np.linalg.norm(d.astype(np.float64), axis=1)
    sp = _xyzr(scene)[np.maximum(oid, 0)]
    reach = np.maximum(np.abs(org).max(axis=1), np.abs(sp[:, :3]).max(axis=1) + np.abs(sp[:, 3])) + np.abs(ot) * dlen
    tol = 1e-4 * np.abs(ot) + 8 * np.finfo(np.float32).eps * reach / dlen / np.sqrt(np.minimum(margin, 1.0))
    assert np.all(np.abs(t[ok] - ot[ok]) <= tol[ok]), np.max(np.abs(t[ok] - ot[ok]) / tol[ok])
    assert np.array_equal(front[ok], ofront[ok])


@pytest.mark.parametrize("name", ["random_spheres_lit", "random_spheres_floor", "random_spheres_lit_floor"])
def test_general_kernel_with_a_long_sphere_list_takes_the_matrix_core_filter(name, oracle64, monkeypatch):
    """Lights (Camera::computeDirectLighting, reference src/Camera.cpp:161-178: a shadow query per light and hit),
    other primitive kinds and the selection halo run in the general kernel; with 485 spheres its sphere scan is the
    matrix-core filter too.  Same bytes and ray count as with the packed filter, and the oracle's image."""
    from raytracer_amd.api import FLAG_DIRECT_LIGHT

    scene = scenes.by_name(name)
    lit = scene.lights is not None and len(scene.lights) > 0
    W, H, spp, depth = 112, 64, 24, 12
    p = RenderParams(W, H, spp, depth, seed=9, flags=FLAG_DIRECT_LIGHT if lit else 0)
    with Renderer(0) as r:
        r.set_scene(scene)
        assert r.filter_describe()["in_use"]
        a = r.render_host(p, want_accum=True)
        ms_mfma = r.last_kernel_ms()
    monkeypatch.setenv("RT_AMD_SPHERE_SOURCE", "scalar")
    with Renderer(0) as r:
        r.set_scene(scene)
        assert not r.filter_describe()["in_use"]
        b = r.render_host(p, want_accum=True)
        ms_packed = r.last_kernel_ms()
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ref, _, ref_rays = oracle64.render(scene, W, H, spp, depth, seed=9, direct_light=lit)
    diff = a[0][:, :, :3].astype(np.int32) - ref[:, :, :3].astype(np.int32)
    close = float(np.mean(np.abs(diff) <= 1))
    print(f"{name}: kernel {ms_mfma:.2f} ms (packed filter {ms_packed:.2f} ms), rays {a[2]} vs oracle {ref_rays}, "
          f"{100 * close:.2f} % of bytes within 1 of the oracle's")
    assert close >= 0.99
    assert abs(a[2] - ref_rays) <= 4e-3 * ref_rays


def _light_above(box):
    from raytracer_amd.api import LIGHT_DTYPE
    lights = np.zeros(1, dtype=LIGHT_DTYPE)
    lights[0]["position"] = (0.3 * box[0], -0.2 * box[1], 3.0 * box[2] + 2.0)
    lights[0]["ambient"] = (0.03, 0.03, 0.03)
    lights[0]["diffuse"] = (0.5, 0.5, 0.45)
    lights[0]["specular"] = (0.8, 0.8, 0.8)
    lights[0]["view_pow"] = 16.0
    return lights


@pytest.mark.parametrize("n,variant,others,cull", [(1473, 8 + 3, False, True), (2600, 8 + 5, False, True), (2600, 8 + 4, False, False),
                                                   (2600, 8 + 5, True, True), (2600, 8 + 4, True, False)])
def test_general_kernel_streams_a_table_too_long_for_lds(n, variant, others, cull, oracle64, monkeypatch):
    """A lit scene (Camera::computeDirectLighting, reference src/Camera.cpp:161-178) of more spheres than a table
    resident in LDS holds: the general kernel streams the table through LDS in tiles like the sphere-only kernel
    does (its workgroup is smaller, so a table the sphere-only kernel already streams may still be resident for it:
    the first case; the last two add a floor plane, a box and a selected sphere with its halo).  By default such a table
    is read from memory block by block where a wave's rays can meet a block's bounding sphere (variant 5); with
    RT_AMD_BLOCK_CULL=0 the whole table passes through LDS in tiles (variant 4).  Which kernel ran is asked of the
    library; bytes, sums and ray counts equal the packed filter's, and the frame is the oracle's."""
    from raytracer_amd.api import FLAG_DIRECT_LIGHT

    box = (30.0, 30.0, 4.0)
    def more(b, mats):
        b.plane((0.0, 0.0, -box[2] - 0.5), (0.0, 0.0, 1.0), mats[3])
        b.cube((3.0, -2.0, box[2] + 1.5), (1.5, 1.0, 0.8), mats[1])
        b.select(5)

    scene = _cloud("lit_streamed", n, 7, box=box, radii=(0.1, 0.5), more=more if others else None)
    scene.lights = _light_above(box)
    W, H, spp, depth = 96, 64, 12, 12
    p = RenderParams(W, H, spp, depth, seed=4, flags=FLAG_DIRECT_LIGHT)
    if not cull:
        monkeypatch.setenv("RT_AMD_BLOCK_CULL", "0")
    with Renderer(0) as r:
        r.set_scene(scene)
        assert r.last_kernel_variant() == -1
        a = r.render_host(p, want_accum=True)
        assert r.last_kernel_variant() == variant  # general kernel, table resident (3) / streamed (4) / read where bounds are met (5)
        ms_mfma = r.last_kernel_ms()
        r.render_host(RenderParams(W, H, spp, depth, seed=4))  # no lights asked for: the sphere-only kernel, if spheres are all there is
        assert r.last_kernel_variant() == (variant if others else (5 if cull else 4))
    monkeypatch.setenv("RT_AMD_SPHERE_SOURCE", "scalar")
    with Renderer(0) as r:
        r.set_scene(scene)
        b = r.render_host(p, want_accum=True)
        assert r.last_kernel_variant() == 8 + 2
        ms_packed = r.last_kernel_ms()
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ref, _, ref_rays = oracle64.render(scene, W, H, spp, depth, seed=4, direct_light=True)
    diff = a[0][:, :, :3].astype(np.int32) - ref[:, :, :3].astype(np.int32)
    close = float(np.mean(np.abs(diff) <= 1))
    print(f"{n} spheres lit: kernel {ms_mfma:.2f} ms (packed filter {ms_packed:.2f} ms), rays {a[2]} vs oracle {ref_rays}, "
          f"{100 * close:.2f} % of bytes within 1 of the oracle's, mean |difference| {np.abs(diff).mean():.3f}")
    # 12 samples per pixel among thousands of small spheres: a path that grazes one goes the other way in the
    # oracle's double arithmetic and moves its pixel by more than one step (the packed filter's frame is the same)
    assert close >= 0.96 and np.abs(diff).mean() < 0.5
    assert abs(a[2] - ref_rays) <= 4e-3 * ref_rays


def test_single_ray_queries_through_the_table(oracle64):
    """SceneManager::hitClosest is asked one ray at a time by interactive callers (picking): a batch of one, of 63
    and of 65 rays through the matrix-core scan, where most lanes of a wave carry no ray."""
    scene = scenes.by_name("random_spheres")
    org, d = oracle64.primary_rays(scene.camera, 13, 5)
    with Renderer(0) as r:
        r.set_scene(scene)
        assert r.filter_describe()["in_use"]
        for n in (1, 63, 65):
            ids, t, nrm, front = r.first_hit(org[:n], d[:n], tmin=np.float32(0.001), refine=False)
            oid, ot, _, ofront, _ = oracle64.first_hit(scene, org[:n], d[:n], tmin=float(np.float32(0.001)))
            assert np.array_equal(ids, oid) and np.array_equal(front[oid >= 0], ofront[oid >= 0])
            assert np.allclose(t[oid >= 0], ot[oid >= 0], rtol=1e-4)


def random_cloud_case(seed, setenv, lit=False):
    """One random scene (16 to 3000 spheres, radii over two decades, up to three very large spheres, the camera
    inside or outside the cloud) rendered with the matrix-core filter and with the packed filter.
    -> (same, in_view, what): `same`: frames, radiance sums and ray counts are equal; `in_view`: the scene is in
    view (paths bounce and the frame is not one colour), i.e. the comparison says something."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([16, 17, 31, 33, 64, 200, 485, 1000, 1440, 1441, 2000, 3000]))
    rmax = float(rng.uniform(0.1, 1.0))
    # a box the spheres fill well enough for paths to bounce: about six largest-sphere areas of floor per sphere
    side, stretch = float(np.sqrt(n * 6.0) * rmax), float(rng.uniform(0.5, 2.0))
    box = (0.5 * side * stretch, 0.5 * side / stretch, rmax * float(rng.uniform(1.0, 6.0)))
    extra = [((float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5)), -float(rng.uniform(200, 2000))), None)
             for _ in range(int(rng.integers(0, 4)))]
    extra = [(c, -c[2] - box[2]) for c, _ in extra]  # large spheres whose tops lie just under the cloud
    scene = _cloud(f"rand{seed}", n, 2000 + seed, box=box, radii=(rmax / float(rng.uniform(1.5, 30)), rmax), extra=extra)
    if rng.random() < 0.4:  # the camera inside the cloud
        c = np.array([0.0, 0.0, 0.2 * box[2]])
        dvec = rng.normal(size=3)
        dvec[2] = -abs(dvec[2]) * 0.3 - 0.05  # along the cloud, slightly downwards
        scene.camera = scenes.make_camera(c, dvec)
    p = RenderParams(80, 48, int(rng.choice([4, 16, 40])), int(rng.choice([3, 12, 50])), seed=seed)
    if lit:  # tools/stress_filters.py: the general kernel, with a light above the cloud
        from raytracer_amd.api import FLAG_DIRECT_LIGHT
        scene.lights = _light_above(box)
        p.flags = FLAG_DIRECT_LIGHT
    with Renderer(0) as r:
        r.set_scene(scene)
        assert r.filter_describe()["in_use"]
        a = r.render_host(p, want_accum=True)
    setenv("RT_AMD_SPHERE_SOURCE", "scalar")
    with Renderer(0) as r:
        r.set_scene(scene)
        b = r.render_host(p, want_accum=True)
    same = a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    in_view = (a[2] > 1.003 * p.width * p.height * p.samples_per_pixel and
               len(np.unique(a[0][:, :, :3].reshape(-1, 3), axis=0)) > 50)
    return same, in_view, f"n={n}: rays {a[2]} vs {b[2]}, {int((a[0] != b[0]).sum())} bytes differ"


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_clouds_render_the_same_bytes_with_either_filter(seed, monkeypatch):
    """The matrix-core filter and the packed filter select the same hits, so frames, radiance sums and ray counts
    of random scenes are equal."""
    same, in_view, what = random_cloud_case(seed, monkeypatch.setenv)
    assert same, what
    assert in_view, f"seed {seed} shows nothing: {what}"


@pytest.mark.parametrize("case", ["c3_small", "few_samples", "one_pixel_tiles", "inside", "behind", "wide", "thin", "thin8", "depth1", "depth0",
                                  "far", "far_thin"])
def test_primary_ray_lists_change_no_byte(case, monkeypatch):
    """The sphere-only matrix-core kernel decides a new sample's primary ray (Camera::genRay, reference
    src/Camera.cpp:124-135: from the camera centre through the pixel) against the spheres its tile's pyramid can meet,
    listed once per tile, and runs scans for bounced rays only.  With the lists switched off (every ray takes the scan)
    frames, radiance sums and ray counts are the same; with them the kernel runs fewer scans, and every ray that is
    not a listed primary still gets one."""
    scene = scenes.by_name("random_spheres")
    W, H, spp, depth = 160, 96, 24, 50
    if case == "few_samples":      # 8x8-pixel tiles: wide pyramids, long lists, some of them too long
        spp = 3
    elif case == "one_pixel_tiles":
        W, H, spp = 48, 30, 1100
    elif case == "inside":         # the camera inside one of the large spheres (glass), looking along the ground
        scene.camera = scenes.make_camera((0.0, 1.0, 0.0), (1.0, -0.1, 0.3))
    elif case == "behind":         # most of the scene behind the camera
        scene.camera = scenes.make_camera((3.0, 0.6, 1.0), (1.0, 0.05, 0.2))
    elif case in ("thin", "thin8"):  # 1x1- and 8x1-pixel tiles of a 4K frame: pyramids 10^-4 rad across, where the cosine
        W, H, spp = 3840, 2160, 6    # of the opening angle rounds to 1 and neighbouring corner directions are parallel in float
        monkeypatch.setenv("RT_AMD_TILE", "0" if case == "thin" else "3")
    elif case == "depth1":         # primary rays only: with lists hardly a scan (tiles whose list is too long)
        depth = 1
    elif case == "depth0":         # no ray at all (the reference returns black at depth 0, src/Camera.cpp:138)
        depth = 0
    elif case == "wide":           # a low frame of very coarse pixels: most tiles' lists are too long, their rays take the scan
        W, H = 96, 4
    elif case in ("far", "far_thin"):
        # scene and camera moved 10^4 away from the origin: a ray's viewport point is rounded at that magnitude
        # (2 ulp of 10^4 is a 4K pixel and more), the list's side planes are built from the small difference
        # ul - c; the list's margin must cover the difference or silhouette hits are lost
        shift = np.array([1.0e4, -2.0e4, 5.0e3], np.float32)
        for k, ax in enumerate(("cx", "cy", "cz")):
            scene.spheres[ax] = scene.spheres[ax] + shift[k]
        c = np.array([13.0, -3.0, 2.0]) + shift.astype(np.float64)
        scene.camera = scenes.make_camera(c, (-13.0, 3.0, -2.0))
        if case == "far_thin":
            W, H, spp = 3840, 2160, 4
            monkeypatch.setenv("RT_AMD_TILE", "0")
    p = RenderParams(W, H, spp, depth, seed=21)
    with Renderer(0) as r:
        r.set_scene(scene)
        a = r.render_host(p, want_accum=True)
        assert r.last_kernel_variant() == 3
        scans_with = r.last_scans()
    monkeypatch.setenv("RT_AMD_PRIM_MIN", "0")
    with Renderer(0) as r:
        r.set_scene(scene)
        b = r.render_host(p, want_accum=True)
        scans_without = r.last_scans()
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    primaries = W * H * spp
    print(f"{case}: rays {a[2]} ({primaries} primary), scans {scans_with} with lists, {scans_without} without")
    assert 64 * scans_without >= a[2] and 64 * scans_with >= a[2] - primaries
    if case == "depth1":
        assert a[2] == primaries and scans_with < 0.1 * scans_without
    elif case == "depth0":
        assert a[2] == 0 and not a[0][:, :, :3].any()
    assert scans_with <= scans_without if case in ("wide", "depth0") else scans_with < 0.8 * scans_without


def test_the_whole_c4_frame_is_the_same_bytes_with_and_without_primary_ray_lists(monkeypatch):
    """Benchmark configuration c4 (485 spheres, 3840x2160, 1000 samples per pixel, depth 50) in full: primary rays
    decided from their tile's list (Camera::genRay, reference src/Camera.cpp:124-135) against every ray taking the
    scan: frame, radiance sums and ray count equal, about half as many scans.  The thin pyramids of 4K tiles are
    where a too-tight list loses silhouette hits."""
    scene = scenes.by_name("random_spheres")
    p = RenderParams(3840, 2160, 1000, 50, seed=0)
    with Renderer(0) as r:
        r.set_scene(scene)
        a = r.render_host(p, want_accum=True)
        assert r.last_kernel_variant() == 3
        ms_with, scans_with = r.last_kernel_ms(), r.last_scans()
    monkeypatch.setenv("RT_AMD_PRIM_MIN", "0")
    with Renderer(0) as r:
        r.set_scene(scene)
        b = r.render_host(p, want_accum=True)
        ms_without, scans_without = r.last_kernel_ms(), r.last_scans()
    print(f"c4: {ms_with:.1f} ms with lists ({scans_with} scans), {ms_without:.1f} ms without ({scans_without}), rays {a[2]}")
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]), f"{int((a[0] != b[0]).sum())} bytes of the frame differ"
    assert np.array_equal(a[1], b[1]), "radiance sums differ"
    assert 64 * scans_without >= a[2] and scans_with < 0.6 * scans_without
