// Camera.h — the render driver with the reference's interface (reference inc/Camera.h:10-137).
// Camera::render keeps its signature; underneath it flattens the SceneManager into the plain
// arrays of include/rt_amd.h and calls the HIP library (rt_upload_scene_ex / rt_upload_lights /
// rt_set_camera / rt_render_host).  There is no CPU rendering path here.
#ifndef RTAMD_CAMERA_H
#define RTAMD_CAMERA_H

#include <cstdint>
#include <utility>
#include <vector>

#include "Backend.h"
#include "RTGeometry.h"
#include "RTObjects.h"

struct rt_ctx;

namespace rtamd {

class SceneManager;

struct RTPixelColor {
    uint8_t r, g, b, a;
};

struct CameraRenderProperties {
    int samplesPerPixel;
    int samplesPerScatter;
    int maxRayDepth;
    int threadPixelbunchSize; // never read by the reference either
    bool enableParallelRender; // the GPU path is always parallel
    bool enableLDirect;
    bool enableRayTracerMode; // never read by the reference either
};

struct Viewport {
    static constexpr const double VIEWPORT_WIDTH = 1;
    static constexpr const double VIEWPORT_HEIGHT = 1;
    gm::IPoint3 upperLeft_;
    gm::IVec3f rightDir_;
    gm::IVec3f downDir_;
    gm::IVec2f viewAngle_;
};

class Camera {
public:
    CameraRenderProperties renderProperties = {
        /*samplesPerPixel*/ 3, /*samplesPerScatter*/ 3,   /*maxRayDepth*/ 10,      /*threadPixelbunchSize*/ 64,
        /*enableParallelRender*/ true, /*enableLDirect*/ true, /*enableRayTracerMode*/ true,
    };

    // Extras of this backend (not in the reference): which GPU, the RNG seed, and counters of the last render.
    int device = 0;
    uint32_t seed = 0;
    uint64_t lastRaysTraced = 0;

private:
    static constexpr const double FOCAL_LENGTH = 1;
    gm::IPoint3 center_ = gm::IPoint3(0, 0, 0);
    gm::IVec3f direction_ = gm::IVec3f(0, 0, 1);
    Viewport viewPort_ = {};
    rt_ctx *ctx_ = nullptr;
    int ctxDevice_ = -1;

public:
    Camera();
    ~Camera();
    Camera(const Camera &) = delete;
    Camera &operator=(const Camera &) = delete;

    void rotate(const double widthRadians, const double heightRadians);
    void move(const gm::IVec3f motionVec);

    // outputBufer must hold screenResolution.first * screenResolution.second pixels (row-major, id = y*W + x).
    // samplesPerScatter = S > 1 is rendered as S * samplesPerPixel single-scatter samples: same
    // expectation as the reference's S-way branching at every hit, without its S^depth ray tree.
    void render(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                std::vector<RTPixelColor> &outputBufer);
    // same render (the reference's parallel / serial pixel loops, inc/Camera.h:72-84)
    void renderParallel(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                        std::vector<RTPixelColor> &outputBufer);
    void renderSerial(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                      std::vector<RTPixelColor> &outputBufer);
    // one pixel of the frame (inc/Camera.h:86-91); equals that pixel of render()
    RTPixelColor renderPixelColor(const SceneManager &sceneManager, const int pixelId,
                                  const std::pair<int, int> screenResolution);

    gm::IVec3f direction() const;
    gm::IPoint3 center() const;
    Viewport viewPort() const;
    void setCenter(const gm::IPoint3 center);
    void setDirection(const gm::IVec3f direction);

private:
    void updateViewPort();
    void renderStrips(const SceneManager &sceneManager, int W, int H, int stripFirst, int stripStride,
                      RTPixelColor *rgba);
};

std::ostream &operator<<(std::ostream &stream, const Camera &camera);

} // namespace rtamd
#endif
