"""Tuning aid: kernel time of sphere clouds of given sizes (the table resident in LDS up to 1440 spheres, streamed beyond)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from raytracer_amd import Renderer, RenderParams, scenes
for n in [int(x) for x in sys.argv[1:]] or [1000, 1440, 1441, 2000, 3000]:
    rng = np.random.default_rng(1)
    b = scenes._Builder()
    mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5)]
    side = n ** 0.5
    for i in range(n):
        b.sphere(((rng.random() - 0.5) * side, (rng.random() - 0.5) * side, 0.3 + rng.random()), 0.3, mats[i % 3])
    sc = b.build("cloud", scenes.make_camera((side, -0.8 * side, 0.5 * side + 1), (-side, 0.8 * side, -0.5 * side - 0.5)))
    with Renderer(0) as r:
        r.set_scene(sc)
        p = RenderParams(960, 540, 64, 50)
        r.render_host(p); _, _, rays = r.render_host(p)
        info = r.filter_describe()
        ms = r.last_kernel_ms()
        print(f"n={n} rows={info['n_rows']}: {ms:.2f} ms, rays {rays}, {ms * 1e6 / (rays / 64 * info['n_rows'] / 32):.1f} ns per block per wave-scan")
