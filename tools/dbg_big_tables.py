"""Debug aid: grids of tens of thousands of spheres (tables of bounds too long for LDS) with the table walked where its
bounds are met and streamed whole: times, blocks walked, frame hashes."""
import os, sys, hashlib
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
for name in ("grid170", "grid200"):
    sc = scenes.by_name(name)
    res = []
    for env in ({}, {"RT_AMD_SPHERE_SOURCE": "scalar"}):
        os.environ.pop("RT_AMD_SPHERE_SOURCE", None); os.environ.update(env)
        with Renderer(0) as r:
            r.set_scene(sc)
            rgba, acc, rays = r.render_host(RenderParams(160, 90, 4, 30, seed=1), want_accum=True)
            res.append((hashlib.sha256(acc.tobytes()).hexdigest()[:12], rays, r.last_kernel_variant(), round(r.last_kernel_ms(), 2), r.last_blocks() / max(r.last_scans(), 1)))
    print(name, len(sc.spheres), res, "SAME" if res[0][:2] == res[1][:2] else "DIFFERENT")
