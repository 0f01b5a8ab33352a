"""Debug aid: the lit cloud of tools/perf_lit_sizes.py rendered by the kernel that culls table blocks, the one that
streams the table and the packed filter's: ray counts and frame hashes.  usage: dbg_lit_cull.py n W H spp [plain]"""
import os, sys, hashlib
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
n, W, H, spp = (int(x) for x in sys.argv[1:5])
plain = len(sys.argv) > 5
rng = np.random.default_rng(n)
b = scenes._Builder()
mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5)]
b.sphere((0, 0, -1000.0), 1000.0, mats[0])
side = max(2.0, n ** 0.5)
for i in range(n - 1):
    b.sphere(((rng.random() - 0.5) * side, (rng.random() - 0.5) * side, 0.3), 0.3, mats[i % 3])
if not plain:
    b.light((3.0, -2.0, 6.0), (0.03, 0.03, 0.03), (0.5, 0.5, 0.45), (0.8, 0.8, 0.8), 24.0)
sc = b.build("lit", scenes.make_camera((side, -0.8 * side, 0.5 * side + 1), (-side, 0.8 * side, -0.5 * side - 0.5)))
res = {}
for tag, env in (("culled", {"RT_AMD_BLOCK_CULL": "1"}), ("streamed", {"RT_AMD_BLOCK_CULL": "0"}), ("packed", {"RT_AMD_SPHERE_SOURCE": "scalar"})):
    for k in ("RT_AMD_BLOCK_CULL", "RT_AMD_SPHERE_SOURCE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with Renderer(0) as r:
        r.set_scene(sc)
        rgba, acc, rays = r.render_host(RenderParams(W, H, spp, 50, seed=0, flags=0 if plain else 2), want_accum=True)
        res[tag] = (rgba, acc, rays)
        print(f"{tag:9s} variant {r.last_kernel_variant():2d} rays {rays} rgba {hashlib.sha256(rgba.tobytes()).hexdigest()[:12]} sums {hashlib.sha256(acc.tobytes()).hexdigest()[:12]}", flush=True)
for a, b_ in (("culled", "packed"), ("streamed", "packed"), ("culled", "streamed")):
    d = np.argwhere((res[a][1] != res[b_][1]).any(axis=2))
    print(f"{a} vs {b_}: {len(d)} pixels differ in their sums", d[:5].tolist())
