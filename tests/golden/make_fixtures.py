#!/usr/bin/env python3
"""Writes tests/golden/first_hit_*.npz and tests/golden/render_hashes.json from the CPU oracle
(double precision).  The reference ships no vectors of its own and cannot be built here, so these
fixtures do not pin the oracle to the reference; they pin the oracle (and everything compared with it)
to itself across changes: a regression in the restatement shows up as a changed fixture.

Run:  python tests/golden/make_fixtures.py     (needs oracle/liborc_f64.so: make -C oracle)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.orc import Oracle  # noqa: E402
from raytracer_amd import scenes  # noqa: E402


def fnv1a(data: bytes) -> str:
    h = 0xCBF29CE484222325
    for chunk in (data[i:i + 65536] for i in range(0, len(data), 65536)):
        for b in chunk:
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


FIRST_HIT = [("small_bubble", 64, 36), ("random_spheres", 64, 36), ("mixed_selected", 64, 36)]
RENDERS = [("small", 400, 225, 10, 10, False), ("small_bubble", 96, 54, 16, 50, False),
           ("random_spheres", 96, 54, 8, 50, False), ("small_phong", 96, 54, 8, 10, True),
           ("mixed_lit", 96, 54, 8, 10, True)]


def main():
    orc = Oracle("f64")
    for name, W, H in FIRST_HIT:
        scene = scenes.by_name(name)
        org, dirs = orc.primary_rays(scene.camera, W, H)
        general = bool(scene.needs_general)
        ids, t, nrm, front, _, exp = orc.first_hit(scene, org, dirs, tmin=float(np.float32(0.001)), general=general,
                                                   expanded_state=general, want_expanded=True)
        np.savez_compressed(os.path.join(HERE, f"first_hit_{name}.npz"), org=org.astype(np.float32),
                            dir=dirs.astype(np.float32), id=ids.astype(np.int32), t=t.astype(np.float32),
                            normal=nrm.astype(np.float32), front=front.astype(np.uint8), expanded=exp.astype(np.uint8))
    hashes = {}
    for name, W, H, spp, depth, lit in RENDERS:
        scene = scenes.by_name(name)
        rgba, _, rays = orc.render(scene, W, H, spp, depth, seed=7, direct_light=lit,
                                   general=bool(scene.needs_general))
        hashes[f"{name}_{W}x{H}_{spp}spp_d{depth}"] = dict(fnv1a=fnv1a(rgba.tobytes()), rays=int(rays))
    with open(os.path.join(HERE, "render_hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1)
    print("wrote", len(FIRST_HIT), "first-hit fixtures and", len(hashes), "render hashes")


if __name__ == "__main__":
    main()
