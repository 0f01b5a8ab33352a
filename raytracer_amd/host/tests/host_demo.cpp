// host_demo — drives the reference-shaped host API (RTMaterialManager, the Primitives classes,
// SceneManager, Camera, SceneIO) for the tests.
// usage: host_demo <out.rgba> <W> <H> <spp> <depth> [ppm-path|-] [scene]   renders; raw RGBA8 goes to <out.rgba>
//            scene: small (default) | lit (small + two Phong lights, enableLDirect) |
//                   mixed | mixed_selected | mixed_lit (plane, cubes, polygons, spheres; _lit = selected + a light)
//        host_demo --pose cx cy cz dx dy dz        prints the viewport vectors (no GPU needed)
//        host_demo --unsupported                   exercises the rejection paths (no GPU needed)
//        host_demo --save <scene> <file>           writes the scene file (no GPU needed)
//        host_demo --resave <in> <out>             loads a scene file and writes it again (no GPU needed)
//        host_demo --pick <scene> <W> <H> <expanded 0|1>
//                                                  SceneManager::hitClosest for the W x H pixel-centre rays of the
//                                                  default camera; prints "id t nx ny nz front expanded object" per ray
//        host_demo --pixels <scene> <W> <H> <spp> <depth>
//                                                  renderPixelColor / renderParallel / renderSerial against render
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>

#include "../Camera.h"
#include "../Output.h"
#include "../RayTracer.h"
#include "../SceneIO.h"

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

// every primitive kind in one scene
static void buildMixedScene(SceneManager &scene, RTMaterialManager &mm, bool selected) {
    RTMaterial *ground = mm.MakeLambertian(gm::IVec3f(0.6, 0.6, 0.5));
    RTMaterial *blue = mm.MakeLambertian(gm::IVec3f(0.1, 0.2, 0.5));
    RTMaterial *gold = mm.MakeMetal(RTColor(0.8, 0.6, 0.2), 0.1);
    RTMaterial *glass = mm.MakeDielectric(gm::IVec3f(1.0, 1.0, 1.0), 1.5);
    RTMaterial *red = mm.MakeLambertian(gm::IVec3f(0.7, 0.15, 0.1));
    RTMaterial *green = mm.MakeLambertian(gm::IVec3f(0.2, 0.6, 0.2));
    scene.addObject(new PlaneObject(gm::IPoint3(0.0, -0.5, 0.0), gm::IVec3f(0.0, 1.0, 0.0), ground));
    scene.addObject(gm::IPoint3(0.0, 0.0, 3.0), new SphereObject(0.5, blue));
    scene.addObject(gm::IPoint3(-1.3, -0.1, 3.2), new CubeObject(gm::IVec3f(0.4, 0.4, 0.4), gold));
    scene.addObject(gm::IPoint3(1.1, 0.0, 2.8), new SphereObject(0.5, glass));
    scene.addObject(new PolygonObject({{-2.0, -0.5, 5.0}, {2.0, -0.5, 5.0}, {2.0, 1.5, 5.0}, {-2.0, 1.5, 5.0}}, red));
    scene.addObject(new PolygonObject({{0.2, 0.6, 2.2}, {0.9, 0.9, 2.4}, {0.5, 1.3, 2.6}}, green));
    scene.addObject(gm::IPoint3(0.3, -0.3, 1.8), new CubeObject(gm::IVec3f(0.15, 0.2, 0.15), glass));
    if (selected)
        for (int i : {1, 2, 5}) scene.primitives()[(size_t)i]->setSelectFlag(true);
}

// returns whether the variant renders with lights
static bool buildScene(const std::string &name, SceneManager &scene, RTMaterialManager &mm) {
    if (name == "small" || name == "lit") {
        buildSmallScene(scene, mm);
        if (name == "lit") {
            scene.addLight(gm::IPoint3(2.0, 3.0, 1.0), new Light(gm::IVec3f(0.05, 0.05, 0.05),
                                                                 gm::IVec3f(0.6, 0.6, 0.5), gm::IVec3f(1, 1, 1), 16.0));
            scene.addLight(gm::IPoint3(-3.0, 1.0, 4.5), new Light(gm::IVec3f(0.0, 0.0, 0.02),
                                                                  gm::IVec3f(0.2, 0.3, 0.5), gm::IVec3f(1, 1, 1), 4.0));
        }
        return name == "lit";
    }
    if (name == "mixed" || name == "mixed_selected" || name == "mixed_lit") {
        buildMixedScene(scene, mm, name != "mixed"); // mixed_lit: selection and one light
        if (name == "mixed_lit")
            scene.addLight(gm::IPoint3(2.0, 3.0, 1.0), new Light(gm::IVec3f(0.05, 0.05, 0.05),
                                                                 gm::IVec3f(0.6, 0.6, 0.5), gm::IVec3f(1, 1, 1), 16.0));
        return name == "mixed_lit";
    }
    std::fprintf(stderr, "unknown scene '%s'\n", name.c_str());
    std::exit(2);
}

namespace {
// a primitive kind the GPU path does not know
class TorusObject : public Primitives {
public:
    explicit TorusObject(RTMaterial *m) : Primitives(m) {}
    std::string typeString() const override { return "Torus"; }
};
} // namespace

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
            Camera cam;
            cam.renderProperties.enableLDirect = false;
            std::vector<RTPixelColor> small(3);
            try { cam.render(scene, {16, 8}, small); } catch (const BackendError &e) { caught += e.status() == -1; }
            scene.addObject(gm::IPoint3(0, 0, 2), new TorusObject(mm.MakeLambertian(gm::IVec3f(1, 1, 1))));
            std::vector<RTPixelColor> out(16 * 8);
            try { cam.render(scene, {16, 8}, out); } catch (const BackendError &e) { caught += e.status() == -3; }
            HitRecord rec;
            try {
                scene.hitClosest(Ray({0, 0, 0}, {0, 0, 1}), Interval(0.001, 1e30), rec, false);
            } catch (const BackendError &e) { caught += e.status() == -3; }
        }
        std::printf("caught %d\n", caught);
        return caught == 3 ? 0 : 1;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "--save")) {
        SceneManager scene;
        RTMaterialManager mm;
        buildScene(argv[2], scene, mm);
        Camera cam;
        std::ofstream out(argv[3]);
        saveScene(out, scene, &cam);
        return out ? 0 : 1;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "--resave")) {
        SceneManager scene;
        RTMaterialManager mm;
        Camera cam;
        std::ifstream in(argv[2]);
        std::string err;
        if (!in || !loadScene(in, scene, mm, &cam, &err)) {
            std::fprintf(stderr, "load failed: %s\n", err.c_str());
            return 1;
        }
        std::ofstream out(argv[3]);
        saveScene(out, scene, &cam);
        return out ? 0 : 1;
    }
    if (argc >= 6 && !std::strcmp(argv[1], "--pick")) {
        SceneManager scene;
        RTMaterialManager mm;
        buildScene(argv[2], scene, mm);
        const int W = atoi(argv[3]), H = atoi(argv[4]);
        const bool expanded = atoi(argv[5]) != 0;
        Camera cam;
        Viewport v = cam.viewPort();
        std::vector<Ray> rays;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                gm::IPoint3 p = v.upperLeft_ + v.rightDir_ * ((x + 0.5) / W) + v.downDir_ * ((y + 0.5) / H);
                rays.emplace_back(cam.center(), (p - cam.center()).normalized());
            }
        try {
            std::vector<HitRecord> rec;
            std::vector<char> hit;
            scene.hitClosest(rays, Interval(0.001, std::numeric_limits<double>::infinity()), expanded, rec, hit);
            for (size_t i = 0; i < rays.size(); ++i) {
                int id = -1;
                if (hit[i]) {
                    // the index of HitRecord::object in the scene (stays -1 for the null object of cubes / polygons)
                    for (size_t k = 0; k < scene.primitives().size(); ++k)
                        if (rec[i].object == scene.primitives()[k]) id = (int)k;
                }
                std::printf("%d %d %.9g %.9g %.9g %.9g %d %d\n", hit[i] ? 1 : 0, id, rec[i].time, rec[i].normal.x(),
                            rec[i].normal.y(), rec[i].normal.z(), rec[i].frontFace ? 1 : 0, rec[i].hitExpanded ? 1 : 0);
            }
            // the single-ray form on the middle ray must agree with the batch
            HitRecord one;
            size_t mid = rays.size() / 2;
            bool h1 = scene.hitClosest(rays[mid], Interval(0.001, std::numeric_limits<double>::infinity()), one, expanded);
            if (h1 != (hit[mid] != 0) || one.time != rec[mid].time || one.object != rec[mid].object) return 3;
            // an interval that ends before the hit reports a miss
            if (hit[mid]) {
                HitRecord none;
                if (scene.hitClosest(rays[mid], Interval(0.001, rec[mid].time * 0.5), none, expanded)) return 4;
            }
        } catch (const BackendError &e) {
            std::fprintf(stderr, "hitClosest failed (%d): %s\n", e.status(), e.what());
            return 1;
        }
        return 0;
    }
    if (argc >= 7 && !std::strcmp(argv[1], "--pixels")) {
        SceneManager scene;
        RTMaterialManager mm;
        const bool lit = buildScene(argv[2], scene, mm);
        const int W = atoi(argv[3]), H = atoi(argv[4]);
        Camera cam;
        cam.renderProperties.samplesPerPixel = atoi(argv[5]);
        cam.renderProperties.samplesPerScatter = 1;
        cam.renderProperties.maxRayDepth = atoi(argv[6]);
        cam.renderProperties.enableLDirect = lit;
        cam.seed = 5;
        try {
            std::vector<RTPixelColor> a((size_t)W * H), b((size_t)W * H), c((size_t)W * H);
            cam.render(scene, {W, H}, a);
            cam.renderParallel(scene, {W, H}, b);
            cam.renderSerial(scene, {W, H}, c);
            if (std::memcmp(a.data(), b.data(), a.size() * 4) || std::memcmp(a.data(), c.data(), a.size() * 4)) return 3;
            const int ids[] = {0, W - 1, W * (H / 2) + W / 2, W * (H - 1), W * H - 1, W * 9 + 3};
            for (int id : ids) {
                if (id >= W * H) continue;
                RTPixelColor px = cam.renderPixelColor(scene, id, {W, H});
                if (std::memcmp(&px, &a[(size_t)id], 4)) {
                    std::fprintf(stderr, "pixel %d differs\n", id);
                    return 4;
                }
            }
            bool threw = false;
            try { cam.renderPixelColor(scene, W * H, {W, H}); } catch (const BackendError &e) { threw = e.status() == -1; }
            if (!threw) return 5;
        } catch (const BackendError &e) {
            std::fprintf(stderr, "render failed (%d): %s\n", e.status(), e.what());
            return 1;
        }
        std::printf("pixels ok\n");
        return 0;
    }
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <out.rgba> <W> <H> <spp> <depth> [ppm|-] [scene] [samplesPerScatter]\n", argv[0]);
        return 2;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    SceneManager scene;
    RTMaterialManager mm;
    const bool lit = buildScene(argc > 7 ? argv[7] : "small", scene, mm);
    Camera cam; // default pose: centre 0, direction +z
    cam.renderProperties.samplesPerPixel = atoi(argv[4]);
    cam.renderProperties.samplesPerScatter = argc > 8 ? atoi(argv[8]) : 1; // (the reference's default is 3)
    cam.renderProperties.maxRayDepth = atoi(argv[5]);
    cam.renderProperties.enableLDirect = lit;
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
