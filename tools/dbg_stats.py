import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import raytracer_amd.api as api
api.LIB_PATH='tools/librt_dbg.so'
from raytracer_amd import Renderer, RenderParams, scenes
r=Renderer(0); s=scenes.random_spheres_scene(); r.set_scene(s)
lib=api.load_library()
out=(C.c_ulonglong*8)()
lib.rt_debug_read(out)
W,H,SPP=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
p=RenderParams(W,H,samples_per_pixel=SPP,max_depth=50)
rgba,acc,rays=r.render_host(p)
lib.rt_debug_read(out)
o=list(out)
nq=((s.n_spheres+7)//8)*2
print('rays',rays,'ideal wave-scans',rays/64,'quads/scan',nq)
print('loop iterations',o[4],' with any active lane',o[6],' active lanes per scanning iteration',o[5]/max(o[6],1))
print('quad-branch entries',o[0],'per scan',o[0]/max(o[6],1),' avg lanes',o[1]/max(o[0],1))
print('resolve entries (wave-level)',o[2],'per scan',o[2]/max(o[6],1),' avg lanes',o[3]/max(o[2],1))
