"""Debug aid: renders the lit cloud of tools/perf_lit_sizes.py several times with one library and prints ray counts and hashes."""
import os, sys, hashlib
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
n, W, H, spp, reps = (int(x) for x in sys.argv[1:6])
lit = len(sys.argv) <= 6
rng = np.random.default_rng(n)
b = scenes._Builder()
mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5)]
b.sphere((0, 0, -1000.0), 1000.0, mats[0])
side = max(2.0, n ** 0.5)
for i in range(n - 1):
    b.sphere(((rng.random() - 0.5) * side, (rng.random() - 0.5) * side, 0.3), 0.3, mats[i % 3])
if lit: b.light((3.0, -2.0, 6.0), (0.03, 0.03, 0.03), (0.5, 0.5, 0.45), (0.8, 0.8, 0.8), 24.0)
sc = b.build("lit", scenes.make_camera((side, -0.8 * side, 0.5 * side + 1), (-side, 0.8 * side, -0.5 * side - 0.5)))
with Renderer(0) as r:
    r.set_scene(sc)
    out = []
    for k in range(reps):
        rgba, acc, rays = r.render_host(RenderParams(W, H, spp, 50, seed=0, flags=2 if lit else 0), want_accum=True)
        out.append((rays, hashlib.sha256(acc.tobytes()).hexdigest()[:10]))
    print(os.environ.get("RT_AMD_LIB", "default"), "variant", r.last_kernel_variant(), out, flush=True)
