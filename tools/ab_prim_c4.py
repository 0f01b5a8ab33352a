"""Confidence aid: the c4 frame (3840x2160, 1000 spp) with and without primary-ray lists must be the same bytes."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
def run(pm):
    os.environ["RT_AMD_PRIM_MIN"] = str(pm)
    with Renderer(0) as r:
        r.set_scene(scenes.by_name("random_spheres"))
        out = r.render_host(RenderParams(3840, 2160, 1000, 50, seed=0), want_accum=True)
        return out, r.last_kernel_ms(), r.last_scans()
a, ms0, s0 = run(0)
b, ms1, s1 = run(64)
same = a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
print(f"c4: off {ms0:.1f} ms ({s0} scans), lists {ms1:.1f} ms ({s1} scans), rays {a[2]} / {b[2]}: {'same bytes' if same else 'DIFFERENT'}")
sys.exit(0 if same else 1)
