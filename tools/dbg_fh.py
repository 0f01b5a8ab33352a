"""Tuning aid: the rays whose plain first-hit t is farthest from the double evaluation."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib.util
spec = importlib.util.spec_from_file_location('tf', 'tests/test_gpu_filter.py'); tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
from raytracer_amd import Renderer, scenes
from oracle.orc import Oracle
scene = scenes.by_name("random_spheres")
rng = np.random.default_rng(23)
orc = Oracle("f64")
with Renderer(0) as r:
    r.set_scene(scene)
    info = r.filter_describe()
    org, d = tf._rays(scene, 30000, rng, spread=info["radius"])
    ids, t, nrm, front = r.first_hit(org, d, tmin=np.float32(0.001), refine=False)
oid, ot, onrm, ofront, margin = orc.first_hit(scene, org, d, tmin=float(np.float32(0.001)), want_margin=True)
ok = (ids == oid) & (oid >= 0) & (margin >= 1e-3)
dlen = np.linalg.norm(d.astype(np.float64), axis=1)
err = np.abs(t - ot) * dlen  # in world units
idx = np.argsort(-(err * ok))[:8]
sp = scene.spheres
for i in idx:
    k = oid[i]
    c = np.array([sp["cx"][k], sp["cy"][k], sp["cz"][k]], np.float64); rr = float(sp["r"][k])
    oc = org[i].astype(np.float64) - c
    print(i, "id", k, "r", rr, "|oc|", np.linalg.norm(oc), "t*|d|", ot[i] * dlen[i], "err(world)", err[i], "margin", margin[i], "org", org[i])
