"""Tuning aid: the turn counters of a -DRT_STATS build (tools/v_stats.so) for one render: turns with and without a scan,
decided primary rays and waiting lanes per turn."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["RT_AMD_LIB"] = os.path.abspath("tools/v_stats.so")
from raytracer_amd import Renderer, RenderParams, scenes
from raytracer_amd.api import load_library
lib = load_library()
out = (C.c_ulonglong * 16)()
r = Renderer(0); r.set_scene(scenes.by_name("random_spheres"))
turns = (C.c_ulonglong * 8)()
lib.rt_debug_stats(out, 1)
lib.rt_debug_turns(turns, 1)
_, _, rays = r.render_host(RenderParams(1200, 675, 100, 50))
lib.rt_debug_stats(out, 0)
it, lanes = out[0], out[1]
sc = r.last_scans()
print(f"rays {rays}, loop turns {it}, scans {sc}, turns without scan {it - sc} ({(it - sc) / sc:.2f} per scan), lanes with a resolved ray per turn {lanes / it:.1f}")
lib.rt_debug_turns(turns, 0)
t = list(turns)
print(f"turns without a scan that had work {t[0]}: lanes with a decided primary {t[1] / max(t[0], 1):.1f}, lanes waiting for a scan {t[2] / max(t[0], 1):.1f}")
print(f"primary rays decided from a list: miss {t[3]} ({100.0 * t[3] / max(t[3] + t[4] + t[5], 1):.1f} %), hit {t[4]}, left to the scan {t[5]}")
