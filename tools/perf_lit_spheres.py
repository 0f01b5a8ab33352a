"""Times the general kernel on the random-spheres scene with one Phong light added (tuning aid)."""
import sys, time, hashlib
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Renderer, RenderParams, scenes
from raytracer_amd.api import LIGHT_DTYPE
sc = scenes.by_name("random_spheres")
l = np.zeros(1, LIGHT_DTYPE); l["position"][0] = (5, -8, 12); l["ambient"][0] = (0.02, 0.02, 0.02)
l["diffuse"][0] = (0.5, 0.5, 0.4); l["specular"][0] = (1, 1, 1); l["view_pow"][0] = 16
sc.lights = l
W, H, spp = 600, 338, 50
with Renderer(0) as r:
    r.set_scene(sc)
    p = RenderParams(W, H, spp, 50, seed=1, flags=2)
    r.render_host(p)
    t0 = time.perf_counter(); rgba, _, rays = r.render_host(p); dt = time.perf_counter() - t0
    print(f"random_spheres + light {W}x{H} {spp} spp: kernel {r.last_kernel_ms() if hasattr(r.lib,'rt_last_kernel_ms') else dt*1e3:.2f} ms, "
          f"rays/sample {rays/(W*H*spp):.3f}, sha {hashlib.sha256(rgba.tobytes()).hexdigest()[:12]}")
