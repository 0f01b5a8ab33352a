"""A filter table too long for LDS is read block by block, and only where some ray of the wave can meet the sphere
around the block (SceneManager::hitClosest, reference src/RayTracer.cpp:68-82, asks every object; a block that no
ray's line comes near holds no sphere that query could return).  The cut must change nothing: the frames of the kernel
that walks the wanted blocks (variant 5, the default) and of the one that passes the whole table through LDS
(RT_AMD_BLOCK_CULL=0, variant 4) are compared byte for byte, with their radiance sums and ray counts."""
import numpy as np
import pytest

from raytracer_amd import Renderer, RenderParams, scenes
from raytracer_amd.api import FLAG_DIRECT_LIGHT

from tests.test_gpu_filter import _cloud, _light_above

pytestmark = pytest.mark.gpu


def _render(scene, p, cull, monkeypatch):
    monkeypatch.setenv("RT_AMD_BLOCK_CULL", "1" if cull else "0")
    with Renderer(0) as r:
        r.set_scene(scene)
        rgba, acc, rays = r.render_host(p, want_accum=True)
        return rgba, acc, rays, r.last_kernel_variant(), r.last_scans(), r.last_blocks(), r.filter_describe()


def _camera_at(scene, eye, at):
    scene.camera = scenes.make_camera(np.asarray(eye, dtype=np.float64), np.asarray(at, dtype=np.float64) - np.asarray(eye, dtype=np.float64))
    return scene


def _cases():
    box = (30.0, 30.0, 4.0)
    return {
        # the benchmark's grid of 10 001 spheres (flat: the bounds are discs of neighbours), whole and clipped tiles
        "grid10k": (lambda: scenes.by_name("grid10k"), RenderParams(160, 90, 8, 50, seed=3)),
        "grid10k_clipped": (lambda: scenes.by_name("grid10k"), RenderParams(67, 35, 5, 30, seed=4)),
        # 4096 spheres: 128 blocks, four blocks of bounds
        "grid64": (lambda: scenes.by_name("grid64"), RenderParams(128, 72, 8, 50, seed=5)),
        # a deep cloud, radii over a decade, some negative (hollow shells), glass and metal among them
        "cloud": (lambda: _cloud("cloud", 3000, 7, box=box, radii=(0.05, 0.9), extra=[((1.0, 2.0, 0.5), -0.7), ((-3.0, 1.0, 0.0), -1.5)]),
                  RenderParams(96, 64, 12, 20, seed=6)),
        # the same far from the world's origin: the frame's centre and the bounds are rounded where coordinates are large
        "offset": (lambda: _cloud("offset", 2500, 8, centre=(5000.0, -3000.0, 2000.0), box=box, radii=(0.1, 0.6)), RenderParams(96, 64, 12, 20, seed=7)),
        # the camera inside the cloud, looking along it: rays start within the bounds of many blocks
        "inside": (lambda: _camera_at(_cloud("inside", 2600, 9, box=box, radii=(0.1, 0.5)), (0.5, 0.2, 0.1), (20.0, 9.0, 1.0)),
                   RenderParams(96, 64, 12, 20, seed=8)),
        # the camera some 100 cloud radii out: every ray's origin term is large against the bounds
        "far": (lambda: _camera_at(_cloud("far", 2600, 10, box=box, radii=(0.2, 0.9)), (2500.0, -3000.0, 1800.0), (0.0, 0.0, 0.0)),
                RenderParams(96, 64, 12, 20, seed=9)),
        # 28 900 spheres: the table of bounds itself no longer fits in LDS and is read from memory like the table
        "grid170": (lambda: scenes.by_name("grid170"), RenderParams(96, 54, 4, 30, seed=11)),
        # one more sphere than the largest table that stays resident (45 blocks): the shortest table that is cut
        "shortest": (lambda: _cloud("shortest", 1441, 8, box=(25.0, 25.0, 4.0), radii=(0.1, 0.5)), RenderParams(96, 64, 12, 12, seed=10)),
    }


@pytest.mark.parametrize("name", ["grid10k", "grid10k_clipped", "grid64", "cloud", "offset", "inside", "far", "grid170", "shortest"])
def test_walking_only_the_blocks_in_reach_changes_no_byte(name, monkeypatch):
    make, p = _cases()[name]
    scene = make()
    a = _render(scene, p, True, monkeypatch)
    b = _render(scene, p, False, monkeypatch)
    assert a[3] == 5 and b[3] == 4, f"kernel variants {a[3]}, {b[3]}"
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] > 1.1 * p.width * p.height * p.samples_per_pixel or name == "far", "hardly any ray bounced: the frame says little"
    rows = a[6]["table_rows"]
    # the whole table: every scan walks every block; the cut table: the blocks of bounds and what they let through
    assert b[5] == b[4] * (rows // 32)
    assert a[5] < a[4] * (rows // 32)
    print(f"{name}: {len(scene.spheres)} spheres, {rows // 32} blocks; blocks walked per scan {a[5] / max(a[4], 1):.1f} "
          f"(whole table: {b[5] / max(b[4], 1):.1f}), scans {a[4]} / {b[4]}")


@pytest.mark.parametrize("others", [False, True])
def test_the_general_kernel_walks_only_the_blocks_in_reach_too(others, monkeypatch):
    """Lights (a shadow query per light and hit, Camera::computeDirectLighting, reference src/Camera.cpp:161-178), a floor
    plane, a box and a selected sphere beside 2600 spheres: the general kernel's sphere scan takes the same cut."""
    box = (30.0, 30.0, 4.0)

    def more(b, mats):
        b.plane((0.0, 0.0, -box[2] - 0.5), (0.0, 0.0, 1.0), mats[3])
        b.cube((3.0, -2.0, box[2] + 1.5), (1.5, 1.0, 0.8), mats[1])
        b.select(5)

    scene = _cloud("lit_cut", 2600, 7, box=box, radii=(0.1, 0.5), more=more if others else None)
    scene.lights = _light_above(box)
    p = RenderParams(96, 64, 12, 12, seed=4, flags=FLAG_DIRECT_LIGHT)
    a = _render(scene, p, True, monkeypatch)
    b = _render(scene, p, False, monkeypatch)
    assert a[3] == 8 + 5 and b[3] == 8 + 4, f"kernel variants {a[3]}, {b[3]}"
    assert a[2] == b[2], f"ray counts differ: {a[2]} vs {b[2]}"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[5] < b[5]


def test_a_frame_rendered_in_strips_is_the_same_with_the_cut(monkeypatch):
    """The multi-GPU partition (8-row strips dealt round-robin) renders the same pixels whichever kernel scans."""
    scene = scenes.by_name("grid10k")
    whole = _render(scene, RenderParams(96, 48, 6, 30, seed=2), True, monkeypatch)
    rows = []
    for rank in range(3):
        part = _render(scene, RenderParams(96, 48, 6, 30, seed=2, strip_first=rank, strip_stride=3), True, monkeypatch)
        rows.append(part[0])
    frame = np.zeros_like(whole[0])
    for rank, part in enumerate(rows):
        for k in range(part.shape[0] // 8):
            s = rank + 3 * k
            frame[8 * s:8 * s + 8] = part[8 * k:8 * k + 8]
    assert np.array_equal(frame, whole[0])


def _field(n, lit):
    """n - 1 spheres of radius 0.3 scattered over a square on a ground sphere, overlapping where chance has it (glass,
    metal and diffuse in turn), seen from above one corner; with a light every hit sends a shadow query."""
    rng = np.random.default_rng(n)
    b = scenes._Builder()
    mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5)]
    b.sphere((0, 0, -1000.0), 1000.0, mats[0])
    side = max(2.0, n ** 0.5)
    for i in range(n - 1):
        b.sphere(((rng.random() - 0.5) * side, (rng.random() - 0.5) * side, 0.3), 0.3, mats[i % 3])
    if lit:
        b.light((3.0, -2.0, 6.0), (0.03, 0.03, 0.03), (0.5, 0.5, 0.45), (0.8, 0.8, 0.8), 24.0)
    return b.build("field", scenes.make_camera((side, -0.8 * side, 0.5 * side + 1), (-side, 0.8 * side, -0.5 * side - 0.5)))


@pytest.mark.parametrize("n,lit", [(2000, True), (5000, True), (5000, False)])
def test_a_large_frame_is_the_same_bytes_every_time_and_with_every_kernel(n, lit, monkeypatch):
    """A hundred million rays through thousands of waves: the kernel that walks the wanted blocks renders the frame three
    times to the same bytes, and they are the bytes of the kernel that streams the whole table and of the packed
    filter's kernel.  (What a wave walks depends on which rays share it, which changes from run to run; the frame may
    not.  The general kernel with primary-ray lists once failed exactly this, in a few tiles of some runs.)"""
    scene = _field(n, lit)
    p = RenderParams(960, 540, 64 if lit else 32, 50, seed=0, flags=FLAG_DIRECT_LIGHT if lit else 0)
    want = (8 if lit else 0) + 5
    runs = []
    monkeypatch.setenv("RT_AMD_BLOCK_CULL", "1")
    with Renderer(0) as r:
        r.set_scene(scene)
        for _ in range(3):
            runs.append(r.render_host(p, want_accum=True))
        assert r.last_kernel_variant() == want
    streamed = _render(scene, p, False, monkeypatch)
    assert streamed[3] == want - 1
    monkeypatch.setenv("RT_AMD_SPHERE_SOURCE", "scalar")
    with Renderer(0) as r:
        r.set_scene(scene)
        packed = r.render_host(p, want_accum=True)
        assert r.last_kernel_variant() == (8 if lit else 0) + 2
    for k, a in enumerate(runs):
        assert a[2] == packed[2] == streamed[2], f"run {k}: ray counts {a[2]}, streamed {streamed[2]}, packed filter {packed[2]}"
        assert np.array_equal(a[1], packed[1]) and np.array_equal(a[0], packed[0]), f"run {k} differs from the packed filter's frame"
    assert np.array_equal(streamed[1], packed[1]) and np.array_equal(streamed[0], packed[0])
