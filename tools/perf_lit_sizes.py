"""Tuning aid: the general kernel on lit clouds of n spheres, with the table taken from RT_AMD_MF_FULL_MIN spheres on."""
import os, sys, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    from raytracer_amd import Renderer, RenderParams, scenes
    n = int(sys.argv[2])
    rng = np.random.default_rng(n)
    b = scenes._Builder()
    mats = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric((1.0, 1.0, 1.0), 1.5)]
    b.sphere((0, 0, -1000.0), 1000.0, mats[0])
    side = max(2.0, n ** 0.5)
    for i in range(n - 1):
        b.sphere(((rng.random() - 0.5) * side, (rng.random() - 0.5) * side, 0.3), 0.3, mats[i % 3])
    if os.environ.get("PLAIN") != "1":
        b.light((3.0, -2.0, 6.0), (0.03, 0.03, 0.03), (0.5, 0.5, 0.45), (0.8, 0.8, 0.8), 24.0)
    sc = b.build("lit", scenes.make_camera((side, -0.8 * side, 0.5 * side + 1), (-side, 0.8 * side, -0.5 * side - 0.5)))
    r = Renderer(0); r.set_scene(sc)
    p = RenderParams(960, 540, 64, 50, flags=0 if os.environ.get('PLAIN') == '1' else 2)
    r.render_host(p); _, _, rays = r.render_host(p)
    print(f"n={n} min={os.environ.get('RT_AMD_MF_FULL_MIN')} source={os.environ.get('RT_AMD_SPHERE_SOURCE', 'table')} in_use={r.filter_describe()['in_use']}: {r.last_kernel_ms():.2f} ms, rays {rays}")
    sys.exit(0)
if os.environ.get("PLAIN") == "1":  # sphere-only kernel: table against packed filter
    for n in (9, 12, 16, 24, 32, 64):
        for src in ("", "scalar"):
            env = dict(os.environ)
            if src: env["RT_AMD_SPHERE_SOURCE"] = src
            subprocess.check_call(["timeout", "-k", "10", "120", sys.executable, __file__, "--child", str(n)], env=env)
    sys.exit(0)
for n in (16, 32, 64, 128, 256):
    for m in ("9", "100000"):
        env = dict(os.environ, RT_AMD_MF_FULL_MIN=m)
        subprocess.check_call(["timeout", "-k", "10", "120", sys.executable, __file__, "--child", str(n)], env=env)
