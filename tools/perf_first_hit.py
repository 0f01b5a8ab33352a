"""Times the batch closest-hit entry point (rt_first_hit, device pointers) on the primary rays of a frame,
plain and with the double-precision refinement (tuning / profiling aid)."""
import sys, time
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import torch
from raytracer_amd import Renderer, scenes
from oracle.orc import Oracle
name, W, H = (sys.argv[1:4] + ["random_spheres", 1920, 1080][len(sys.argv) - 1:])[:3]
W, H = int(W), int(H)
sc = scenes.by_name(name)
org, dirs = Oracle("f64").primary_rays(sc.camera, W, H, seed=1, smp=0, jitter=True)
r = Renderer(0); r.set_scene(sc)
n = len(org)
d_org, d_dir = torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda()
ids = torch.zeros(n, dtype=torch.int32, device='cuda'); t = torch.zeros(n, device='cuda')
nrm = torch.zeros((n, 3), device='cuda'); front = torch.zeros(n, dtype=torch.uint8, device='cuda')
for flags, what in ((0, "fp32"), (1, "refined")):
    for rep in range(4):
        if rep == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = r.lib.rt_first_hit(r.h, n, C.c_void_p(d_org.data_ptr()), C.c_void_p(d_dir.data_ptr()), C.c_float(0.001), flags,
                                C.c_void_p(ids.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(nrm.data_ptr()),
                                C.c_void_p(front.data_ptr()), None)
        assert rc == 0
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"first hit {what}: {name} {sc.n_spheres} spheres, {n} rays: {dt*1e3:.3f} ms, {n/dt/1e6:.1f} Mrays/s, "
          f"hits {int((ids >= 0).sum())}")
