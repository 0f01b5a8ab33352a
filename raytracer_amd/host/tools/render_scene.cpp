// render_scene — renders a scene file (SceneIO.h) on the GPU and writes a binary PPM.
// usage: render_scene <scene.txt> <out.ppm> <W> <H> <spp> <depth> [seed] [nodirect] [frames=N] [devices=a,b,...]
//   frames=N       after the first frame, N more with the camera moved a little each time, as an interactive
//                  caller drives it (reference src/Camera.cpp:43-67); prints wall / kernel time per frame and
//                  fails if a frame of the unchanged scene uploads or allocates anything
//   devices=a,b,.. renders the frame once more on this list of GPUs (a GPU may repeat) and fails unless it
//                  equals the one-GPU frame byte for byte
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

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
    int frames = 0;
    std::vector<int> deviceList;
    cam.renderProperties.enableLDirect = true;
    for (int i = 7; i < argc; ++i) {
        if (!std::strncmp(argv[i], "frames=", 7)) frames = atoi(argv[i] + 7);
        else if (!std::strncmp(argv[i], "devices=", 8)) {
            for (const char *s = argv[i] + 8; *s;) {
                deviceList.push_back((int)strtol(s, const_cast<char **>(&s), 10));
                if (*s == ',') ++s;
            }
        } else if (!std::strcmp(argv[i], "nodirect")) cam.renderProperties.enableLDirect = false;
        else cam.seed = (uint32_t)strtoul(argv[i], nullptr, 10);
    }
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
    std::printf("frame 0: wall %.3f ms, kernel %.3f ms, scene uploaded %d, device allocations %llu\n", cam.lastRenderMs,
                cam.lastKernelMs, cam.lastRenderUploadedScene ? 1 : 0, (unsigned long long)cam.deviceAllocations());
    try {
        const gm::IPoint3 home = cam.center();
        for (int f = 1; f <= frames; ++f) {
            // a small step sideways and back: the scene is unchanged, only the pose moves
            cam.setCenter(home + cam.viewPort().rightDir_ * (0.01 * (f % 2)));
            const unsigned long long before = cam.deviceAllocations();
            std::vector<RTPixelColor> next((size_t)W * (size_t)H);
            cam.render(scene, {W, H}, next);
            std::printf("frame %d: wall %.3f ms, kernel %.3f ms, scene uploaded %d, device allocations %llu\n", f,
                        cam.lastRenderMs, cam.lastKernelMs, cam.lastRenderUploadedScene ? 1 : 0,
                        (unsigned long long)cam.deviceAllocations());
            if (cam.lastRenderUploadedScene || cam.deviceAllocations() != before) {
                std::fprintf(stderr, "frame %d of the unchanged scene uploaded or allocated\n", f);
                return 3;
            }
            if (f % 2 == 0 && std::memcmp(next.data(), out.data(), out.size() * 4)) {
                std::fprintf(stderr, "frame %d at the first pose differs from frame 0\n", f);
                return 3;
            }
        }
        cam.setCenter(home);
        if (!deviceList.empty()) {
            cam.devices = deviceList;
            std::vector<RTPixelColor> multi((size_t)W * (size_t)H);
            cam.render(scene, {W, H}, multi);
            std::printf("devices x%zu: wall %.3f ms, longest kernel %.3f ms\n", deviceList.size(), cam.lastRenderMs,
                        cam.lastKernelMs);
            if (std::memcmp(multi.data(), out.data(), out.size() * 4)) {
                std::fprintf(stderr, "the frame of the device list differs from the one-GPU frame\n");
                return 3;
            }
        }
    } catch (const BackendError &e) {
        std::fprintf(stderr, "render failed (%d): %s\n", e.status(), e.what());
        return 1;
    }
    std::cout << scene.primitives().size() << " objects, " << scene.lights().size() << " lights, " << cam
              << ", rays " << cam.lastRaysTraced << "\n";
    return 0;
}
