import sys, numpy as np
sys.path.insert(0,'.')
from oracle.orc import Oracle
from raytracer_amd import Renderer, scenes
o=Oracle('f64'); TMIN=float(np.float32(0.001))
b = scenes._Builder(); m = b.lambertian((0.5, 0.5, 0.5))
b.sphere((0, 0, 3), 0.5, m); b.sphere((0, 0, 3), 0.5, m); b.sphere((0, 2, 3), -0.5, m); b.sphere((5, 0, 0), 1.0, m)
scene = b.build("edge", scenes.make_camera())
r=Renderer(0); r.set_scene(scene)
org = np.array([[0, 0, 0], [0, 2, 0], [0, 0, 3], [0, 0, 0], [3.5, 0, 0], [0, 0.5, 0], [0, 0, 0]], np.float32)
dirs = np.array([[0, 0, 1], [0, 0, 2], [0, 0, 1], [0, 1, 0], [1, 0, 0], [0, 0.5, 1], [0, 0, -1]], np.float32)
dirs[5]=[0,0,1]
for refine in (False,True):
    ids,t,n,f=r.first_hit(org,dirs,tmin=TMIN,refine=refine)
    oid,ot,on,of,_=o.first_hit(scene,org,dirs,tmin=TMIN)
    print(refine, ids, oid); print(t, ot); print(n); print(on)
