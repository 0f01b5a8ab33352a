"""Tuning aid: does the order of the spheres in the filter table matter?  Renders the random-spheres scene as built
and with its spheres sorted along a Morton curve (so that a table block holds neighbours)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from raytracer_amd import Renderer, RenderParams, scenes

def morton(x, y, bits=10):
    def spread(v):
        r = 0
        for i in range(bits):
            r |= ((v >> i) & 1) << (2 * i)
        return r
    return spread(x) | (spread(y) << 1)

name0 = sys.argv[1] if len(sys.argv) > 1 else "random_spheres"
base = scenes.by_name(name0)
W, H, SPP = (1200, 675, 200) if name0 == "random_spheres" else (960, 540, 16)
sp = base.spheres
order = list(range(len(sp)))
ext = float(max(np.abs(sp["cx"][1:]).max(), np.abs(sp["cy"][1:]).max())) + 1.0
xs = ((sp["cx"] + ext) / (2 * ext) * 1023).astype(int).clip(0, 1023)
ys = ((sp["cy"] + ext) / (2 * ext) * 1023).astype(int).clip(0, 1023)
keys = [morton(int(xs[i]), int(ys[i])) for i in order]
sorted_order = [0] + sorted(order[1:], key=lambda i: keys[i])
for name, od in (("as built", order), ("morton", sorted_order), ("as built", order), ("morton", sorted_order)):
    b = scenes._Builder()
    for i in od:
        m = base.mats[base.mat_id[i]]
        mi = b.mat(int(m["type"]), m["albedo"], float(m["fuzz"]), float(m["ior"]), m["emission"])
        b.sphere((sp["cx"][i], sp["cy"][i], sp["cz"][i]), sp["r"][i], mi)
    sc = b.build("rs", base.camera)
    with Renderer(0) as r:
        r.set_scene(sc)
        p = RenderParams(W, H, SPP, 50)
        r.render_host(p); _, _, rays = r.render_host(p)
        print(f"{name}: {r.last_kernel_ms():.2f} ms, rays {rays}")
