// render_scene — renders a scene file (SceneIO.h) on the GPU and writes a binary PPM.
// usage: render_scene <scene.txt> <out.ppm> <W> <H> <spp> <depth> [seed] [nodirect]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../Camera.h"
#include "../Output.h"
#include "../RayTracer.h"
#include "../SceneIO.h"

using namespace rtamd;

int main(int argc, char **argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s <scene.txt> <out.ppm> <W> <H> <spp> <depth> [seed] [nodirect]\n", argv[0]);
        return 2;
    }
    std::ifstream in(argv[1]);
    if (!in) {
        std::fprintf(stderr, "cannot open %s\n", argv[1]);
        return 1;
    }
    SceneManager scene;
    RTMaterialManager materials;
    Camera cam;
    std::string err;
    if (!loadScene(in, scene, materials, &cam, &err)) {
        std::fprintf(stderr, "%s: %s\n", argv[1], err.c_str());
        return 1;
    }
    const int W = atoi(argv[3]), H = atoi(argv[4]);
    cam.renderProperties.samplesPerPixel = atoi(argv[5]);
    cam.renderProperties.samplesPerScatter = 1;
    cam.renderProperties.maxRayDepth = atoi(argv[6]);
    if (argc > 7) cam.seed = (uint32_t)strtoul(argv[7], nullptr, 10);
    cam.renderProperties.enableLDirect = !(argc > 8 && !std::strcmp(argv[8], "nodirect"));
    if (W <= 0 || H <= 0) {
        std::fprintf(stderr, "bad resolution\n");
        return 2;
    }
    std::vector<RTPixelColor> out((size_t)W * (size_t)H);
    try {
        cam.render(scene, {W, H}, out);
    } catch (const BackendError &e) {
        std::fprintf(stderr, "render failed (%d): %s\n", e.status(), e.what());
        return 1;
    }
    if (!writePPM(argv[2], out, W, H)) {
        std::fprintf(stderr, "cannot write %s\n", argv[2]);
        return 1;
    }
    std::cout << scene.primitives().size() << " objects, " << scene.lights().size() << " lights, " << cam
              << ", rays " << cam.lastRaysTraced << "\n";
    return 0;
}
