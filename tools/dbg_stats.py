import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import raytracer_amd.api as api
api.LIB_PATH='tools/librt_dbg.so'
from raytracer_amd import Renderer, RenderParams, scenes
r=Renderer(0); s=scenes.random_spheres_scene(); r.set_scene(s)
lib=api.load_library()
out=(C.c_ulonglong*8)()
lib.rt_debug_read(out)
p=RenderParams(1200,675,samples_per_pixel=20,max_depth=50)
rgba,acc,rays=r.render_host(p)
lib.rt_debug_read(out)
o=list(out)
nq=(s.n_spheres+3)//4
print('rays',rays,'wave-scans(ideal)',rays/64,'quads/scan',nq)
print('branch taken (wave-level)',o[0],'per wave-scan',o[0]/(rays/64),'of',nq,' avg lanes in branch',o[1]/max(o[0],1))
print('lane-level disc>=0 events',o[2],'per ray',o[2]/rays,' accepted',o[3],'per ray',o[3]/rays)
