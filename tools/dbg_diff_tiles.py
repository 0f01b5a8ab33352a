"""Debug aid: renders the lit field of tools/dbg_repeat.py several times with the library in RT_AMD_LIB and once with the
packed filter's kernel, and prints where and by how much the frames that differ do."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
n, W, H, spp, reps = (int(x) for x in sys.argv[1:6])
rng = np.random.default_rng(n)
b = scenes._Builder()
mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5)]
b.sphere((0, 0, -1000.0), 1000.0, mats[0])
side = max(2.0, n ** 0.5)
for i in range(n - 1):
    b.sphere(((rng.random() - 0.5) * side, (rng.random() - 0.5) * side, 0.3), 0.3, mats[i % 3])
b.light((3.0, -2.0, 6.0), (0.03, 0.03, 0.03), (0.5, 0.5, 0.45), (0.8, 0.8, 0.8), 24.0)
sc = b.build("lit", scenes.make_camera((side, -0.8 * side, 0.5 * side + 1), (-side, 0.8 * side, -0.5 * side - 0.5)))
p = RenderParams(W, H, spp, 50, seed=0, flags=2)
os.environ["RT_AMD_SPHERE_SOURCE"] = "scalar"
with Renderer(0) as r:
    r.set_scene(sc)
    _, ref, ref_rays = r.render_host(p, want_accum=True)
    print("packed filter: variant", r.last_kernel_variant(), "rays", ref_rays, flush=True)
del os.environ["RT_AMD_SPHERE_SOURCE"]
with Renderer(0) as r:
    r.set_scene(sc)
    for k in range(reps):
        _, acc, rays = r.render_host(p, want_accum=True)
        d = np.any(acc != ref, axis=2)
        ys, xs = np.nonzero(d)
        print(f"run {k}: variant {r.last_kernel_variant()} rays {rays} ({rays - ref_rays:+d}), pixels that differ {len(ys)}", flush=True)
        for y, x in list(zip(ys, xs))[:40]:
            print(f"   ({x},{y}) x%8={x % 8}: got {acc[y, x]} want {ref[y, x]} diff*spp {(acc[y, x] - ref[y, x])[:3]}")
