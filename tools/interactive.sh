#!/bin/bash
# Camera::render frame after frame through the C++ host layer (render_scene frames=N): wall time per frame against
# kernel time, for the reference's default scene (c1) and the random-spheres scene (c3).  Runs on the GPU box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -c "
import sys; sys.path.insert(0, '.')
from raytracer_amd import scenes
open('gpurun_out/small.txt', 'w').write(scenes.scene_text(scenes.by_name('small')))
open('gpurun_out/random_spheres.txt', 'w').write(scenes.scene_text(scenes.by_name('random_spheres')))
"
R=raytracer_amd/host/tools/render_scene.bin
timeout -k 10 60 $R gpurun_out/small.txt gpurun_out/c1.ppm 400 225 10 10 0 nodirect frames=6 > gpurun_out/interactive_c1.txt 2>&1 &&
timeout -k 10 120 $R gpurun_out/random_spheres.txt gpurun_out/c3.ppm 1200 675 500 50 0 nodirect frames=4 devices=0,0 > gpurun_out/interactive_c3.txt 2>&1
rm -f gpurun_out/c1.ppm gpurun_out/c3.ppm
cat gpurun_out/interactive_c1.txt gpurun_out/interactive_c3.txt
