"""Tuning aid: where the waves of the render kernel spend their cycles (a -DRT_SECTIONS build stamps
s_memtime at the boundaries of the loop body's parts and sums the differences per wave).
  python __graft_entry__.py variant tools/v_sections.so -DRT_SECTIONS
usage: dbg_sections.py [scene W H spp depth [lights]]"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["RT_AMD_LIB"] = os.path.abspath(os.environ.get("SECTIONS_LIB", "tools/v_sections.so"))
from raytracer_amd import Renderer, RenderParams, scenes
from raytracer_amd.api import load_library
name = sys.argv[1] if len(sys.argv) > 1 else "random_spheres"
W, H, spp, depth = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1200, 675, 500, 50)))
flags = 2 if len(sys.argv) > 6 else 0
lib = load_library()
out = (C.c_ulonglong * 24)()
r = Renderer(0); r.set_scene(scenes.by_name(name))
r.render_host(RenderParams(W, H, spp, depth, flags=flags))
lib.rt_debug_sections(out, 1)
_, _, rays = r.render_host(RenderParams(W, H, spp, depth, flags=flags))
lib.rt_debug_sections(out, 0)
v = list(out)
clock_ghz = 0.1 * v[14] / max(v[15], 1)  # shader cycles per tick of the 100 MHz counter, summed over the waves
v[14] = v[15] = 0
tot = sum(v)
names = ["ray set-up + scan", "shade", "accumulate", "store tiles", "hand-out", "rng + raygen/scatter", "-", "outside loop",
         "scan: matrix products + sign masks", "scan: recording candidates", "scan: deciding candidates",
         "scan: big spheres + result", "streamed: first tile + barrier", "streamed: tile barriers", "-", "-",
         "general: other primitive kinds", "general: halo pass", "general: shadow query's result", "-", "-", "-", "-", "-"]
print(f"{name} {W}x{H} {spp} spp: kernel {r.last_kernel_ms():.2f} ms, rays {rays}, scans {r.last_scans()}, in-kernel shader clock {clock_ghz:.3f} GHz")
for n, x in zip(names, v):
    if x: print(f"  {n:34s} {100.0 * x / tot:6.2f} % of wave cycles")
