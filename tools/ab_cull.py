"""A table too long for LDS walked where its blocks' bounds are met (default) against the same table streamed whole
(RT_AMD_BLOCK_CULL=0): bytes, sums, ray counts must be equal; prints kernel times, scans and blocks walked.
usage: ab_cull.py [scene W H spp depth]..."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import Renderer, RenderParams, scenes

args = sys.argv[1:]
cases = [(args[i], *map(int, args[i + 1:i + 5])) for i in range(0, len(args), 5)] or [
    ("grid10k", 480, 270, 16, 50), ("grid10k", 67, 35, 5, 30), ("grid64", 320, 180, 16, 50), ("grid10k", 3840, 2160, 4, 50)]
bad = 0
for name, W, H, spp, depth in cases:
    res = []
    for cull in ("1", "0"):
        os.environ["RT_AMD_BLOCK_CULL"] = cull
        with Renderer(0) as r:
            sc = scenes.by_name(name)
            r.set_scene(sc)
            flags = 2 if sc.lights is not None and len(sc.lights) else 0
            p = RenderParams(W, H, spp, depth, seed=3, flags=flags)
            r.render_host(p)
            rgba, acc, rays = r.render_host(p, want_accum=True)
            res.append((hashlib.sha256(rgba.tobytes()).hexdigest()[:16], hashlib.sha256(acc.tobytes()).hexdigest()[:16], rays,
                        r.last_kernel_variant(), r.last_kernel_ms(), r.last_scans(), r.last_blocks()))
    same = res[0][:3] == res[1][:3]
    bad += not same
    for tag, x in zip(("culled  ", "streamed"), res):
        print(f"{name} {W}x{H} {spp} spp: {tag} variant {x[3]}, {x[4]:.2f} ms, rays {x[2]}, scans {x[5]}, blocks per scan {x[6] / max(x[5], 1):.1f}")
    print("   " + ("same bytes, sums and rays" if same else f"DIFFERENT: {res[0][:3]} vs {res[1][:3]}"), flush=True)
sys.exit(1 if bad else 0)
