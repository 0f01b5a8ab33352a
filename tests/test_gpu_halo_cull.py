"""The selection-halo pass cut down, per tile, to the selected primitives the tile's primary rays can meet.

The reference runs its halo pass (SceneManager::hitClosest, src/RayTracer.cpp:84-97) for every primary ray
against every selected primitive.  The render kernel tests, when it opens a tile, a sphere around every
selected primitive grown by EXPAND_COEF (inc/RTObjects.h:26) against the pyramid of the tile's primary rays
and leaves out what lies outside.  That may not change a byte: these tests render frames with the cut and
with every bound infinite (RT_AMD_HALO_CULL=0) and compare frame, radiance sums and ray count."""
import numpy as np
import pytest

from raytracer_amd import Renderer, RenderParams, scenes

pytestmark = pytest.mark.gpu


def selection_scene(seed, n, camera_in_shell=False):
    """Small spheres, boxes and polygons scattered through the view volume of the default camera (and a few
    beside and behind it), most of them selected, over a ground plane."""
    rng = np.random.default_rng(seed)
    b = scenes._Builder()
    mats = [b.lambertian((0.6, 0.6, 0.5)), b.metal((0.8, 0.6, 0.2), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5),
            b.lambertian((0.1, 0.2, 0.5))]
    b.plane((0.0, -1.2, 0.0), (0.0, 1.0, 0.0), mats[0])
    count = 1
    if camera_in_shell:
        # the camera sits between a selected sphere's surface and its grown shell, and inside a selected box's
        b.sphere((0.0, 0.0, -0.52), 0.5, mats[3])
        b.select(count)
        b.cube((0.3, 0.0, 0.0), (0.29, 0.2, 0.2), mats[1])
        b.select(count + 1)
        count += 2
    for _ in range(n):
        z = rng.uniform(1.2, 9.0) if rng.random() < 0.9 else rng.uniform(-3.0, 0.5)
        c = np.array([rng.uniform(-0.7, 0.7) * abs(z), rng.uniform(-0.6, 0.6) * abs(z), z])
        size = rng.uniform(0.03, 0.12) * max(abs(z), 1.0)
        kind = int(rng.integers(0, 3))
        m = mats[int(rng.integers(0, 4))]
        if kind == 0:
            b.sphere(c, size * (-1 if rng.random() < 0.1 else 1), m)
        elif kind == 1:
            b.cube(c, rng.uniform(0.4, 1.0, 3) * size, m)
        else:
            nv = int(rng.integers(3, 6))
            ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
            u, v = np.linalg.qr(rng.normal(size=(3, 2)))[0].T
            b.polygon([c + size * (np.cos(a) * u + np.sin(a) * v) for a in ang], m)
        if rng.random() < 0.7:
            b.select(count)
        count += 1
    return b.build("selection", scenes.make_camera())


def halo_share(renderer, scene, W, H):
    """Share of the pixel-centre primary rays whose first hit lies on a grown shell."""
    cam = scene.camera[0]
    px, py = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    pts = (cam["upper_left"][None, :] + cam["right"][None, :] * (px.reshape(-1, 1) / W)
           + cam["down"][None, :] * (py.reshape(-1, 1) / H))
    dirs = (pts - cam["center"][None, :]).astype(np.float32)
    org = np.broadcast_to(cam["center"].astype(np.float32), dirs.shape).copy()
    _, _, _, _, exp = renderer.first_hit(org, dirs, tmin=0.001, expanded_state=True, want_expanded=True)
    return float(np.mean(exp != 0))


@pytest.mark.parametrize("seed,n,W,H,tile,lights,in_shell", [
    (1, 40, 320, 180, None, False, False),
    (2, 30, 203, 101, "0", False, False),   # 1x1 tiles, frame not a multiple of anything
    (3, 50, 640, 360, "6", True, False),    # 8x8 tiles, shadow queries between the primary rays
    (4, 24, 96, 54, None, False, True),     # the camera inside grown shells: those are in every tile's view
    (5, 90, 160, 90, None, False, False),   # more entries than the mask has bits: nothing is cut
])
def test_halo_pass_cut_to_what_a_tile_can_see_changes_no_byte(monkeypatch, seed, n, W, H, tile, lights, in_shell):
    scene = selection_scene(seed, n, camera_in_shell=in_shell)
    if lights:
        scene.lights = scenes.mixed_scene(phong=1).lights
    if tile is not None:
        monkeypatch.setenv("RT_AMD_TILE", tile)
    p = RenderParams(W, H, 8, 6, seed=11, flags=2 if lights else 0)
    with Renderer(0) as r:
        r.set_scene(scene)
        a = r.render_host(p, want_accum=True)
        share = halo_share(r, scene, W, H)
    monkeypatch.setenv("RT_AMD_HALO_CULL", "0")
    with Renderer(0) as r:
        r.set_scene(scene)
        b = r.render_host(p, want_accum=True)
    print(f"seed {seed}: {share:.3f} of the primary rays end on a grown shell, rays {a[2]}")
    assert share > 0.01, "the scene shows no halo"
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]), f"{int((a[0] != b[0]).sum())} bytes of the frame differ"
    assert np.array_equal(a[1], b[1]), "radiance sums differ"
