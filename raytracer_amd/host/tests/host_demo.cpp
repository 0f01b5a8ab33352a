// host_demo — builds the small benchmark scene through the reference-shaped host API
// (RTMaterialManager, SphereObject, SceneManager, Camera) and renders it on the GPU.
// usage: host_demo <out.rgba> <W> <H> <spp> <depth> [ppm-path|-] [lit]   raw RGBA8 is written to <out.rgba>;
//            "lit" adds two Phong lights and renders with enableLDirect
//        host_demo --pose cx cy cz dx dy dz        prints the viewport vectors (no GPU needed)
//        host_demo --unsupported                   exercises the rejection paths (needs no GPU)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "../Camera.h"
#include "../Output.h"
#include "../RayTracer.h"

using namespace rtamd;

static void buildSmallScene(SceneManager &scene, RTMaterialManager &mm) {
    RTMaterial *ground = mm.MakeLambertian(gm::IVec3f(0.8, 0.8, 0.0));
    RTMaterial *center = mm.MakeLambertian(gm::IVec3f(0.1, 0.2, 0.5));
    RTMaterial *left = mm.MakeMetal(RTColor(0.8, 0.6, 0.2), 0.3);
    RTMaterial *right = mm.MakeDielectric(gm::IVec3f(1.0, 1.0, 1.0), 1.5);
    scene.addObject(gm::IPoint3(0.0, -100.5, 3.0), new SphereObject(100.0, ground));
    scene.addObject(gm::IPoint3(0.0, 0.0, 3.0), new SphereObject(0.5, center));
    scene.addObject(gm::IPoint3(-1.0, 0.0, 3.0), new SphereObject(0.5, left));
    scene.addObject(gm::IPoint3(1.0, 0.0, 3.0), new SphereObject(0.5, right));
}

int main(int argc, char **argv) {
    if (argc >= 8 && !std::strcmp(argv[1], "--pose")) {
        Camera cam;
        cam.setCenter(gm::IPoint3(atof(argv[2]), atof(argv[3]), atof(argv[4])));
        cam.setDirection(gm::IVec3f(atof(argv[5]), atof(argv[6]), atof(argv[7])));
        Viewport v = cam.viewPort();
        std::printf("%.17g %.17g %.17g\n%.17g %.17g %.17g\n%.17g %.17g %.17g\n", v.upperLeft_.x(), v.upperLeft_.y(),
                    v.upperLeft_.z(), v.rightDir_.x(), v.rightDir_.y(), v.rightDir_.z(), v.downDir_.x(), v.downDir_.y(),
                    v.downDir_.z());
        return 0;
    }
    if (argc >= 2 && !std::strcmp(argv[1], "--unsupported")) {
        int caught = 0;
        {
            SceneManager scene;
            RTMaterialManager mm;
            buildSmallScene(scene, mm);
            scene.primitives()[1]->setSelectFlag(true);
            Camera cam;
            cam.renderProperties.enableLDirect = false;
            std::vector<RTPixelColor> out(16 * 8);
            try { cam.render(scene, {16, 8}, out); } catch (const BackendError &e) { caught += e.status() == -3; }
            std::vector<RTPixelColor> small(3);
            try { cam.render(scene, {16, 8}, small); } catch (const BackendError &e) { caught += e.status() == -1; }
        }
        std::printf("caught %d\n", caught);
        return caught == 2 ? 0 : 1;
    }
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <out.rgba> <W> <H> <spp> <depth> [ppm]\n", argv[0]);
        return 2;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    SceneManager scene;
    RTMaterialManager mm;
    buildSmallScene(scene, mm);
    Camera cam; // default pose: centre 0, direction +z
    cam.renderProperties.samplesPerPixel = atoi(argv[4]);
    cam.renderProperties.samplesPerScatter = 1;
    cam.renderProperties.maxRayDepth = atoi(argv[5]);
    const bool lit = argc > 7 && !std::strcmp(argv[7], "lit");
    cam.renderProperties.enableLDirect = lit;
    if (lit) {
        scene.addLight(gm::IPoint3(2.0, 3.0, 1.0),
                       new Light(gm::IVec3f(0.05, 0.05, 0.05), gm::IVec3f(0.6, 0.6, 0.5), gm::IVec3f(1, 1, 1), 16.0));
        scene.addLight(gm::IPoint3(-3.0, 1.0, 4.5),
                       new Light(gm::IVec3f(0.0, 0.0, 0.02), gm::IVec3f(0.2, 0.3, 0.5), gm::IVec3f(1, 1, 1), 4.0));
    }
    cam.seed = 21;
    std::vector<RTPixelColor> out((size_t)W * H);
    try {
        cam.render(scene, {W, H}, out);
    } catch (const BackendError &e) {
        std::fprintf(stderr, "render failed (%d): %s\n", e.status(), e.what());
        return 1;
    }
    FILE *f = std::fopen(argv[1], "wb");
    if (!f) return 1;
    std::fwrite(out.data(), 4, out.size(), f);
    std::fclose(f);
    if (argc > 6 && std::strcmp(argv[6], "-") && !writePPM(argv[6], out, W, H)) return 1;
    std::cout << cam << " rays " << cam.lastRaysTraced << "\n";
    return 0;
}
