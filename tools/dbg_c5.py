"""Tuning aid: kernel time of a sphere scene under the matrix-core filter and under the scalar-load filter."""
import os, sys, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from raytracer_amd import Renderer, RenderParams, scenes
    name, W, H, spp = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    with Renderer(0) as r:
        r.set_scene(scenes.by_name(name))
        info = r.filter_describe()
        r.render_host(RenderParams(W, H, spp, 50))
        _, _, rays = r.render_host(RenderParams(W, H, spp, 50))
        print(os.environ.get("RT_AMD_SPHERE_SOURCE", "default"), name, f"{r.last_kernel_ms():.2f} ms", "rays", rays,
              {k: info[k] for k in ("in_use", "radius", "n_rows", "big")})
    sys.exit(0)
name, W, H, spp = (sys.argv[1:5] + ["grid10k", "480", "270", "16"][len(sys.argv) - 1:])[:4]
for src in ("", "scalar"):
    env = dict(os.environ)
    if src: env["RT_AMD_SPHERE_SOURCE"] = src
    subprocess.check_call(["timeout", "-k", "10", "200", sys.executable, __file__, "--child", name, W, H, spp], env=env)
