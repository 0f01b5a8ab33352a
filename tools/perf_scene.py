"""Time one render of a named scene through the device-pointer API (tuning aid)."""
import argparse, sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Renderer, RenderParams, scenes
ap = argparse.ArgumentParser()
ap.add_argument('--scene', default='grid10k'); ap.add_argument('--w', type=int, default=480); ap.add_argument('--h', type=int, default=270)
ap.add_argument('--spp', type=int, default=16); ap.add_argument('--depth', type=int, default=50); ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--lights', action='store_true', help='render with RT_FLAG_DIRECT_LIGHT')
a = ap.parse_args()
sc = scenes.by_name(a.scene)
r = Renderer(0); r.set_scene(sc)
p = RenderParams(a.w, a.h, a.spp, a.depth, flags=2 if a.lights else 0)
rgba = torch.zeros((a.h, a.w, 4), dtype=torch.uint8, device='cuda'); rays = torch.zeros(1, dtype=torch.int64, device='cuda')
r.render_device(p, rgba, None, rays); torch.cuda.synchronize(); rays.zero_()
t0 = time.perf_counter()
for i in range(a.reps): r.render_device(p, rgba, None, rays)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
nr = rays.item() / a.reps; n = sc.n_spheres
print(f"{a.scene} n={n} {a.w}x{a.h} spp={a.spp}: {dt*1e3:.2f} ms, {a.w*a.h*a.spp/dt/1e6:.1f} Msamples/s, rays/sample {nr/(a.w*a.h*a.spp):.3f}, "
      f"{nr*(17*n+80)/dt/1e12:.1f} TFLOP/s ({nr*(17*n+80)/dt/157.3e12*100:.1f}% of peak)")
