"""Debug aid for libraries whose render kernel was patched in assembly to OR, per lane, 'this lane's value differs from
lane 0's' into the frame's scan and block counters: prints the upper bits of both after each frame of the lit field."""
import sys
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
n, W, H, spp, reps = 5000, 960, 540, 64, int(sys.argv[1])
rng = np.random.default_rng(n)
b = scenes._Builder()
mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5)]
b.sphere((0, 0, -1000.0), 1000.0, mats[0])
side = max(2.0, n ** 0.5)
for i in range(n - 1):
    b.sphere(((rng.random() - 0.5) * side, (rng.random() - 0.5) * side, 0.3), 0.3, mats[i % 3])
b.light((3.0, -2.0, 6.0), (0.03, 0.03, 0.03), (0.5, 0.5, 0.45), (0.8, 0.8, 0.8), 24.0)
sc = b.build("lit", scenes.make_camera((side, -0.8 * side, 0.5 * side + 1), (-side, 0.8 * side, -0.5 * side - 0.5)))
with Renderer(0) as r:
    r.set_scene(sc)
    for k in range(reps):
        rgba, acc, rays = r.render_host(RenderParams(W, H, spp, 50, seed=0, flags=2), want_accum=True)
        hi1, hi2 = r.last_scans() >> 32, r.last_blocks() >> 32
        print(f"run {k}: rays {rays}, upper words: {hi1:08x} / {hi2:08x} as floats {np.array([hi1, hi2], dtype=np.uint32).view(np.float32)}", flush=True)
