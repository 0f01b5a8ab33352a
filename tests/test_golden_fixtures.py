"""Regression fixtures under tests/golden/ (written by tests/golden/make_fixtures.py from the
double-precision oracle): first-hit buffers and hashes of small renders.  On the CPU the oracle must
reproduce them; on the GPU the refined first hit must match them without the oracle in the loop."""
import json
import os

import numpy as np
import pytest

from raytracer_amd import scenes
from tests.golden.make_fixtures import FIRST_HIT, RENDERS, fnv1a

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TMIN = float(np.float32(0.001))


@pytest.mark.parametrize("name,W,H", FIRST_HIT)
def test_oracle_reproduces_first_hit_fixture(oracle64, name, W, H):
    fx = np.load(os.path.join(GOLDEN, f"first_hit_{name}.npz"))
    scene = scenes.by_name(name)
    general = bool(scene.needs_general)
    ids, t, nrm, front, _, exp = oracle64.first_hit(scene, fx["org"], fx["dir"], tmin=TMIN, general=general,
                                                    expanded_state=general, want_expanded=True)
    assert np.array_equal(ids, fx["id"]) and np.array_equal(front, fx["front"]) and np.array_equal(exp, fx["expanded"])
    assert np.array_equal(t.astype(np.float32), fx["t"]) and np.array_equal(nrm.astype(np.float32), fx["normal"])
    assert (fx["id"] >= 0).mean() > 0.3


def test_oracle_reproduces_render_hashes(oracle64):
    want = json.load(open(os.path.join(GOLDEN, "render_hashes.json")))
    for name, W, H, spp, depth, lit in RENDERS:
        scene = scenes.by_name(name)
        rgba, _, rays = oracle64.render(scene, W, H, spp, depth, seed=7, direct_light=lit,
                                        general=bool(scene.needs_general))
        key = f"{name}_{W}x{H}_{spp}spp_d{depth}"
        assert fnv1a(rgba.tobytes()) == want[key]["fnv1a"], key
        assert int(rays) == want[key]["rays"], key


@pytest.mark.gpu
@pytest.mark.parametrize("name,W,H", FIRST_HIT)
def test_gpu_matches_first_hit_fixture(renderer, name, W, H):
    """First-hit ids, face and halo flags bit-exact against the committed buffers; t and normals to 1e-5."""
    fx = np.load(os.path.join(GOLDEN, f"first_hit_{name}.npz"))
    scene = scenes.by_name(name)
    renderer.set_scene(scene)
    general = bool(scene.needs_general)
    ids, t, nrm, front, exp = renderer.first_hit(fx["org"], fx["dir"], tmin=TMIN, refine=True, expanded_state=general,
                                                 want_expanded=True)
    assert np.array_equal(ids, fx["id"]) and np.array_equal(exp, fx["expanded"])
    hit = fx["id"] >= 0
    assert np.array_equal(front[hit], fx["front"][hit])
    assert np.all(np.abs(t[hit] - fx["t"][hit]) <= 1e-5 * np.abs(fx["t"][hit]) + 1e-7)
    assert np.max(np.abs(nrm[hit] - fx["normal"][hit])) <= 1e-5


@pytest.mark.gpu
def test_gpu_ray_counts_match_render_fixtures(renderer):
    """The number of closest-hit queries of a render is a sensitive summary of every decision on the path:
    the GPU's differs from the committed oracle count by a few per ten thousand at most."""
    from raytracer_amd import RenderParams
    from raytracer_amd.api import FLAG_DIRECT_LIGHT

    want = json.load(open(os.path.join(GOLDEN, "render_hashes.json")))
    for name, W, H, spp, depth, lit in RENDERS:
        renderer.set_scene(scenes.by_name(name))
        p = RenderParams(W, H, spp, depth, seed=7, flags=FLAG_DIRECT_LIGHT if lit else 0)
        _, _, rays = renderer.render_host(p)
        ref = want[f"{name}_{W}x{H}_{spp}spp_d{depth}"]["rays"]
        assert abs(rays - ref) <= max(4e-3 * ref, 20), (name, rays, ref)
