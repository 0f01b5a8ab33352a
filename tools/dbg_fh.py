import sys, numpy as np
sys.path.insert(0, '.')
from oracle.orc import Oracle
from raytracer_amd import Renderer, scenes
TMIN=float(np.float32(0.001))
o=Oracle('f64')
scene=scenes.by_name(sys.argv[1] if len(sys.argv)>1 else 'random_spheres')
r=Renderer(0); r.set_scene(scene)
W,H=400,225
org,dirs=o.primary_rays(scene.camera,W,H,seed=3,smp=1,jitter=False)
oid,ot,on,of,margin=o.first_hit(scene,org,dirs,tmin=TMIN,want_margin=True)
for refine in (True,False):
    ids,t,n,f=r.first_hit(org,dirs,tmin=TMIN,refine=refine)
    bad=np.nonzero(ids!=oid)[0]
    print('refine',refine,'mismatches',len(bad),'of',len(ids))
    for i in bad[:12]:
        print(i, 'gpu',ids[i],t[i],'orc',oid[i],ot[i],'margin',margin[i])
    ok=(ids==oid)&(oid>=0)
    rel=np.abs(t[ok]-ot[ok])/np.abs(ot[ok])
    print(' t rel err max',rel.max(),'median',np.median(rel),' n err max',np.linalg.norm(n[ok]-on[ok],axis=1).max(), 'front mism', (f[ok]!=of[ok]).sum())
