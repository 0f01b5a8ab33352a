"""Tuning aid: the sphere-only matrix-core kernel with and without primary-ray lists (RT_AMD_PRIM_MIN): frames,
radiance sums and ray counts must be equal; kernel times side by side.
usage: ab_prim.py [min ...]      (default: 48)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
CASES = [("random_spheres", 1200, 675, 100, 50), ("random_spheres", 320, 180, 40, 50), ("random_spheres", 160, 96, 500, 50),
         ("random_spheres", 1920, 1080, 4, 50), ("grid30", 640, 360, 32, 50), ("random_spheres", 97, 61, 37, 3)]
mins = [int(x) for x in sys.argv[1:]] or [48]
def run(name, W, H, spp, depth, pm):
    os.environ["RT_AMD_PRIM_MIN"] = str(pm)
    with Renderer(0) as r:
        r.set_scene(scenes.by_name(name))
        p = RenderParams(W, H, spp, depth, seed=3)
        r.render_host(p)
        out = r.render_host(p, want_accum=True)
        return out, r.last_kernel_ms(), r.last_kernel_variant()
bad = 0
for c in CASES:
    (a, ms0, v0) = run(*c, 0)
    line = f"{c}: variant {v0}, off {ms0:.3f} ms"
    for pm in mins:
        (b, ms1, v1) = run(*c, pm)
        same = a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        bad += not same
        line += f" | min {pm}: {ms1:.3f} ms {'same' if same else 'DIFFERENT (rays %d vs %d, bytes differing %d)' % (a[2], b[2], int((a[0] != b[0]).sum()))}"
    print(line, flush=True)
sys.exit(1 if bad else 0)
