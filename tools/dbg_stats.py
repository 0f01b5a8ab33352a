"""Tuning aid: event counters of a -DRT_STATS build for one render of a scene.

Build the instrumented library first (add -DRT_STATS_TIMES_ONLY to keep only the per-wave timing, which
does not slow the kernel down):
  python __graft_entry__.py variant tools/v_stats.so -DRT_STATS
usage: dbg_stats.py [scene W H spp depth]
"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["RT_AMD_LIB"] = os.path.abspath(os.environ.get("STATS_LIB", "tools/v_stats.so"))
from raytracer_amd import Renderer, RenderParams, scenes
from raytracer_amd.api import load_library
name = sys.argv[1] if len(sys.argv) > 1 else "random_spheres"
W, H, spp, depth = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (600, 338, 32, 50)))
lib = load_library()
out = (C.c_ulonglong * 16)()
r = Renderer(0); r.set_scene(scenes.by_name(name))
lib.rt_debug_stats(out, 1)
_, _, rays = r.render_host(RenderParams(W, H, spp, depth))
lib.rt_debug_stats(out, 0)
q_w, q_l, b_w, b_l, s_w, s_l, upd, g8 = list(out)[:8]
print(f"{name}: rays {rays}, scan iterations (per wave) {q_w}, lanes with a ray {q_l/max(q_w,1):.2f} of 64")
if r.filter_describe()["in_use"]:
    print(f"  matrix-core filter: blocks gathered {b_w/max(q_w,1):.2f} per scan, recording turns {b_l/max(q_w,1):.2f} per scan "
          f"({b_l/max(b_w,1):.2f} per gathered block), pairs recorded {s_l/max(q_w,1):.1f} per scan = {s_l/max(rays,1):.2f} per ray")
print(f"  groups of eight entered: per wave {g8} = {g8/max(q_w,1):.2f} per scan")
print(f"  quad bodies: per wave {b_w} = {b_w/max(q_w,1):.2f} per scan; lanes inside {b_l/max(b_w,1):.2f}")
print(f"  sphere bodies: per wave {s_w} = {s_w/max(b_w,1):.2f} per quad body; lanes inside {s_l/max(s_w,1):.2f}")
print(f"  per ray: candidate spheres {s_l/max(rays,1):.2f}, closest-hit updates {upd/max(rays,1):.2f}")
t0, e0, e1, tsum, nw, maxit = list(out)[8:14]
tick = 1e-5  # wall_clock64 ticks at 100 MHz: 10 ns -> ms
print(f"  waves with work {nw}: first finished after {(e0-t0)*tick:.2f} ms, last after {(e1-t0)*tick:.2f} ms, "
      f"mean duration {tsum/max(nw,1)*tick:.2f} ms, most items on one wave {maxit}")
longest, longest_small = list(out)[14:16]
print(f"  longest item: {(longest >> 32)*tick:.2f} ms (item {longest & 0xffffffff}); longest small-tile item: "
      f"{(longest_small >> 32)*tick:.2f} ms (item {longest_small & 0xffffffff})")
