"""Renders a set of scenes with two builds of the library and compares the frames byte for byte
(tuning aid: a change of the kernel's work distribution or scan must not change a byte).
usage: ab_bytes.py libA.so libB.so"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("random_spheres", 320, 180, 40, 50), ("random_spheres", 160, 96, 500, 50), ("small", 256, 144, 64, 50),
         ("small_bubble", 200, 120, 16, 10), ("grid10k", 128, 72, 8, 50), ("mixed_selected", 240, 136, 24, 10),
         ("mixed_lit", 240, 136, 16, 8), ("small_phong", 160, 90, 16, 8), ("halo_overlap", 160, 90, 16, 6),
         # frames whose last tiles are clipped by the image's edge, few and many samples per pixel
         ("small_bubble", 37, 19, 7, 50), ("random_spheres", 1201, 75, 20, 20), ("random_spheres", 203, 101, 300, 50),
         ("grid10k", 67, 35, 5, 30)]
if len(sys.argv) == 4 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    os.environ["RT_AMD_LIB"] = os.path.abspath(sys.argv[2])
    import hashlib
    from raytracer_amd import Renderer, RenderParams, scenes
    out = {}
    for name, W, H, spp, depth in CASES:
        with Renderer(0) as r:
            sc = scenes.by_name(name)
            r.set_scene(sc)
            flags = 2 if sc.lights is not None and len(sc.lights) else 0
            rgba, acc, rays = r.render_host(RenderParams(W, H, spp, depth, seed=3, flags=flags), want_accum=True)
        out[f"{name} {W}x{H} {spp}spp"] = [hashlib.sha256(rgba.tobytes()).hexdigest()[:16],
                                          hashlib.sha256(acc.tobytes()).hexdigest()[:16], rays]
    json.dump(out, open(sys.argv[3], "w"))
    sys.exit(0)
res = []
for i, lib in enumerate(sys.argv[1:3]):
    f = f"/tmp/ab_bytes_{i}.json"
    subprocess.check_call(["timeout", "-k", "10", "200", sys.executable, os.path.abspath(__file__), "--child", lib, f])
    res.append(json.load(open(f)))
bad = 0
for k in res[0]:
    same = res[0][k] == res[1][k]
    bad += not same
    print(("same   " if same else "DIFFER ") + k, res[0][k], "" if same else res[1][k])
sys.exit(1 if bad else 0)
