// Camera.h — the render driver behind the reference's Camera interface (reference inc/Camera.h:10-137).
//
// What is the same: the class and method names, `renderProperties`, the pose API, the 1x1 viewport at
// focal length 1, RTPixelColor frames in row-major order.  What is different underneath: render()
// flattens the SceneManager into the arrays of include/rt_amd.h and runs the HIP library on one GPU
// (rt_render_host) or on a list of GPUs of this process (rt_group_render_host).  There is no CPU
// rendering path.  A Camera keeps its device context and the scene it last uploaded: rendering an
// unchanged scene again (after move() / rotate(), say) uploads and allocates nothing.
#ifndef RTAMD_CAMERA_H
#define RTAMD_CAMERA_H

#include <cstdint>
#include <iosfwd>
#include <utility>
#include <vector>

#include "Backend.h"
#include "RTGeometry.h"
#include "RTObjects.h"

struct rt_ctx;
struct rt_group;

namespace rtamd {

class SceneManager;

// One RGBA8 pixel of the frame (reference inc/Camera.h:10-12).
struct RTPixelColor {
    uint8_t r, g, b, a;
};

// Field for field the reference's knob struct (inc/Camera.h:14-22); defaults are set in Camera.
struct CameraRenderProperties {
    int samplesPerPixel;
    int samplesPerScatter;     // S > 1 renders S * samplesPerPixel single-scatter samples (same expectation)
    int maxRayDepth;
    int threadPixelbunchSize;  // read by nothing, in the reference as here
    bool enableParallelRender; // the GPU path is always parallel
    bool enableLDirect;        // Phong direct lighting from the scene's lights
    bool enableRayTracerMode;  // read by nothing, in the reference as here
};

// The image plane as genRay uses it (reference inc/Camera.h:24-33).
struct Viewport {
    static constexpr const double VIEWPORT_WIDTH = 1;
    static constexpr const double VIEWPORT_HEIGHT = 1;

    gm::IPoint3 upperLeft_;
    gm::IVec3f rightDir_;
    gm::IVec3f downDir_;
    gm::IVec2f viewAngle_;
};

// Viewport of a camera at `eye` looking along the unit vector `forward` (reference
// src/Camera.cpp:206-221): world-up is +z, or +y when `forward` is within 0.999 of +-z.
Viewport makeViewport(const gm::IPoint3 &eye, const gm::IVec3f &forward);

class Camera {
public:
    CameraRenderProperties renderProperties = {3, 3, 10, 64, true, true, true}; // inc/Camera.h:37-46

    // ---- this backend's additions -------------------------------------------------------------
    int device = 0;           // the GPU render() uses ...
    std::vector<int> devices; // ... unless this lists several: the frame is then split over them
    uint32_t seed = 0;        // key of the random streams
    // figures of the last render(): closest-hit queries, the (longest) render kernel, the whole call
    uint64_t lastRaysTraced = 0;
    double lastKernelMs = 0, lastRenderMs = 0;
    bool lastRenderUploadedScene = false;
    uint64_t deviceAllocations() const; // hipMalloc calls made for this camera so far

    Camera();
    ~Camera();
    Camera(const Camera &) = delete;
    Camera &operator=(const Camera &) = delete;

    // ---- the reference's interface -----------------------------------------------------------
    void move(const gm::IVec3f motionVec);
    void rotate(const double widthRadians, const double heightRadians);
    void setCenter(const gm::IPoint3 center);
    void setDirection(const gm::IVec3f direction);
    gm::IPoint3 center() const;
    gm::IVec3f direction() const;
    Viewport viewPort() const;

    // outputBufer must hold width * height pixels (id = y * width + x), as the reference asserts.
    void render(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                std::vector<RTPixelColor> &outputBufer);
    // The reference's two pixel loops (inc/Camera.h:72-84) are one and the same GPU render here.
    void renderParallel(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                        std::vector<RTPixelColor> &outputBufer);
    void renderSerial(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                      std::vector<RTPixelColor> &outputBufer);
    // One pixel (inc/Camera.h:86-91): bytes equal that pixel of render().
    RTPixelColor renderPixelColor(const SceneManager &sceneManager, const int pixelId,
                                  const std::pair<int, int> screenResolution);

private:
    static constexpr const double FOCAL_LENGTH = 1;
    friend Viewport makeViewport(const gm::IPoint3 &, const gm::IVec3f &);

    gm::IPoint3 center_ = gm::IPoint3(0, 0, 0);
    gm::IVec3f direction_ = gm::IVec3f(0, 0, 1);
    Viewport viewPort_ = {};

    rt_ctx *ctx_ = nullptr;     // single-GPU context
    rt_group *group_ = nullptr; // multi-GPU group
    std::vector<int> boundDevices_;
    FlatScene uploaded_; // what the context / group holds
    bool haveUploaded_ = false;

    void bind();
    void release();
    void renderStrips(const SceneManager &sceneManager, int W, int H, int stripFirst, int stripStride,
                      RTPixelColor *rgba);
};

std::ostream &operator<<(std::ostream &stream, const Camera &camera);

} // namespace rtamd
#endif
