import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["RT_AMD_LIB"] = os.path.abspath("tools/v_stats.so")
from raytracer_amd import Renderer, RenderParams, scenes
from raytracer_amd.api import load_library
lib = load_library()
out = (C.c_ulonglong * 16)()
r = Renderer(0); r.set_scene(scenes.by_name("random_spheres"))
lib.rt_debug_stats(out, 1)
_, _, rays = r.render_host(RenderParams(1200, 675, 100, 50))
lib.rt_debug_stats(out, 0)
it, lanes = out[0], out[1]
sc = r.last_scans()
print(f"rays {rays}, loop turns {it}, scans {sc}, turns without scan {it - sc} ({(it - sc) / sc:.2f} per scan), lanes with a resolved ray per turn {lanes / it:.1f}")
