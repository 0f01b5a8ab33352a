"""Confidence aid: random sphere clouds and cameras at 1280x720 with small work-item tiles (thin pyramids of primary
rays), rendered with and without primary-ray lists by the same kernel: frames, sums and ray counts must be equal.
usage: stress_prim.py [first_seed last_seed]"""
import os, sys, time
sys.path.insert(0, '.')
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location('tf', 'tests/test_gpu_filter.py')
tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
from raytracer_amd import Renderer, RenderParams, scenes

first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
bad, t0 = 0, time.time()
for seed in range(first, last):
    rng = np.random.default_rng(70000 + seed)
    n = int(rng.choice([16, 40, 200, 485, 1000, 1440, 2000]))
    rmax = float(rng.uniform(0.1, 1.0))
    side, stretch = float(np.sqrt(n * 6.0) * rmax), float(rng.uniform(0.5, 2.0))
    box = (0.5 * side * stretch, 0.5 * side / stretch, rmax * float(rng.uniform(1.0, 6.0)))
    extra = [((0.0, 0.0, -1000.0 - box[2]), 1000.0)] if rng.random() < 0.7 else []
    scene = tf._cloud(f"p{seed}", n, 3000 + seed, box=box, radii=(rmax / float(rng.uniform(1.5, 30)), rmax), extra=extra)
    if rng.random() < 0.5:  # the camera inside the cloud, any direction
        scene.camera = scenes.make_camera(rng.uniform(-0.5, 0.5, size=3) * np.asarray(box), rng.normal(size=3))
    tile = str(int(rng.choice([0, 1, 2, 3])))
    p = RenderParams(1280, 720, int(rng.choice([1, 2, 5])), int(rng.choice([2, 6, 50])), seed=seed)
    out = []
    for pm in ("64", "0"):
        os.environ["RT_AMD_PRIM_MIN"] = pm
        os.environ["RT_AMD_TILE"] = tile
        with Renderer(0) as r:
            r.set_scene(scene)
            out.append((r.render_host(p, want_accum=True), r.last_scans(), r.last_kernel_variant()))
    (a, sa, va), (b, sb, vb) = out
    same = a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if not same:
        bad += 1
        print(f"seed {seed} DIFFERENT: n={n} tile={tile} variant {va}, rays {a[2]} vs {b[2]}, bytes differing {int((a[0] != b[0]).sum())}")
    elif seed % 10 == 0:
        print(f"seed {seed}: n={n} tile={tile} variant {va}, rays {a[2]}, scans {sa} with lists, {sb} without", flush=True)
print(f"{last - first} scenes in {time.time() - t0:.1f} s, different: {bad}")
sys.exit(1 if bad else 0)
