// Camera: pose and viewport on the host, render() as a client of the HIP library's C ABI
// (include/rt_amd.h).  Pose semantics follow reference src/Camera.cpp:38-52 and :206-236.
#include <chrono>
#include <cmath>
#include <cstring>
#include <ostream>
#include <string>

#include "Backend.h"
#include "Camera.h"
#include "RayTracer.h"

namespace rtamd {

// The viewport is a unit square one focal length in front of the eye.  Its axes come from a fixed
// world "up": +z, replaced by +y when the view direction is (anti)parallel to z within 0.999.
Viewport makeViewport(const gm::IPoint3 &eye, const gm::IVec3f &forward) {
    const bool alongZ = std::fabs(forward.z()) > 0.999; // forward . (0,0,1)
    const gm::IVec3f up = alongZ ? gm::IVec3f(0.0, 1.0, 0.0) : gm::IVec3f(0.0, 0.0, 1.0);
    Viewport vp;
    vp.rightDir_ = cross(forward, up).normalized();
    vp.downDir_ = cross(forward, vp.rightDir_).normalized();
    const double halfW = 0.5 * Viewport::VIEWPORT_WIDTH, halfH = 0.5 * Viewport::VIEWPORT_HEIGHT;
    vp.upperLeft_ = eye + forward * Camera::FOCAL_LENGTH - vp.rightDir_ * halfW - vp.downDir_ * halfH;
    // stored by the reference, read by nothing (x from the height, y from the width, as there)
    vp.viewAngle_.setX(2 * std::atan(Viewport::VIEWPORT_HEIGHT / Camera::FOCAL_LENGTH));
    vp.viewAngle_.setY(2 * std::atan(Viewport::VIEWPORT_WIDTH / Camera::FOCAL_LENGTH));
    return vp;
}

Camera::Camera() : viewPort_(makeViewport(center_, direction_)) {}

Camera::~Camera() { release(); }

void Camera::release() {
    if (ctx_) rt_destroy(ctx_);
    if (group_) rt_group_destroy(group_);
    ctx_ = nullptr;
    group_ = nullptr;
    haveUploaded_ = false;
}

void Camera::setCenter(const gm::IPoint3 center) {
    center_ = center;
    viewPort_ = makeViewport(center_, direction_);
}

void Camera::setDirection(const gm::IVec3f direction) {
    direction_ = direction.normalized();
    viewPort_ = makeViewport(center_, direction_);
}

void Camera::move(const gm::IVec3f motionVec) { setCenter(center_ + motionVec); }

// Yaw about the viewport's down axis (positive turns to the right, hence the sign), then pitch about
// its right axis; both axes are those of the viewport before the turn.
void Camera::rotate(const double widthRadians, const double heightRadians) {
    gm::IVec3f turned = direction_;
    turned.rotate(viewPort_.downDir_, -widthRadians);
    turned.rotate(viewPort_.rightDir_, heightRadians);
    direction_ = turned; // not re-normalised, as in the reference
    viewPort_ = makeViewport(center_, direction_);
}

gm::IPoint3 Camera::center() const { return center_; }
gm::IVec3f Camera::direction() const { return direction_; }
Viewport Camera::viewPort() const { return viewPort_; }

std::ostream &operator<<(std::ostream &stream, const Camera &camera) {
    return stream << "Camera{" << camera.center() << ", " << camera.direction() << "}";
}

uint64_t Camera::deviceAllocations() const {
    return group_ ? rt_group_device_allocations(group_) : (ctx_ ? rt_device_allocations(ctx_) : 0);
}

namespace {
void put3(float *dst, const gm::IVec3f &v) {
    dst[0] = (float)v.x();
    dst[1] = (float)v.y();
    dst[2] = (float)v.z();
}

} // namespace

// (Re)creates the context or the group when the device selection changed.
void Camera::bind() {
    std::vector<int> want = devices.empty() ? std::vector<int>{device} : devices;
    if ((ctx_ || group_) && want == boundDevices_) return;
    release();
    if (want.size() == 1) {
        int rc = rt_create(want[0], &ctx_);
        if (rc != RT_OK) throw BackendError(rc, "rt_create failed (no usable GPU?)");
    } else {
        int rc = rt_group_create((int)want.size(), want.data(), &group_);
        if (rc != RT_OK) throw BackendError(rc, "rt_group_create failed (no usable GPU, or RCCL missing?)");
    }
    boundDevices_ = want;
}

// Uploads scene (when it changed) and pose, then renders strips stripFirst, stripFirst + stripStride, ...
// compactly into `rgba` (rt_output_rows x width pixels).  A device list renders whole frames only.
void Camera::renderStrips(const SceneManager &sceneManager, int W, int H, int stripFirst, int stripStride,
                          RTPixelColor *rgba) {
    const auto t0 = std::chrono::steady_clock::now();
    // flatten first: a scene the GPU path cannot take is rejected before any device is touched.  The
    // flat copy is also what tells whether the device already holds this scene: any edit through the
    // reference's API (addObject, setPosition, the mutable material getters, ...) shows up in it.
    FlatScene flat = flattenScene(sceneManager, renderProperties.enableLDirect);
    bind();
    auto fail = [&](int status, const std::string &where) {
        throw BackendError(status, where + ": " + (group_ ? rt_group_last_error(group_) : rt_last_error(ctx_)));
    };
    lastRenderUploadedScene = !(haveUploaded_ && sameFlatScene(flat, uploaded_));
    if (lastRenderUploadedScene) {
        haveUploaded_ = false;
        if (group_) {
            int rc = rt_group_upload_scene_ex(group_, (int)flat.prims.size(), flat.prims.data(),
                                              (int)(flat.verts.size() / 3), flat.verts.data(), (int)flat.mats.size(),
                                              flat.mats.data());
            if (rc != RT_OK) fail(rc, "rt_group_upload_scene_ex");
            rc = rt_group_upload_lights(group_, (int)flat.lights.size(), flat.lights.data());
            if (rc != RT_OK) fail(rc, "rt_group_upload_lights");
        } else {
            uploadScene(ctx_, flat);
        }
        uploaded_ = std::move(flat);
        haveUploaded_ = true;
    }

    rt_camera cam{};
    put3(cam.center, center_);
    put3(cam.upper_left, viewPort_.upperLeft_);
    put3(cam.right, viewPort_.rightDir_);
    put3(cam.down, viewPort_.downDir_);
    int rc = group_ ? rt_group_set_camera(group_, &cam) : rt_set_camera(ctx_, &cam);
    if (rc != RT_OK) fail(rc, "rt_set_camera");

    rt_params p{};
    p.width = W;
    p.height = H;
    // samplesPerScatter = S > 1: S * samplesPerPixel single-scatter samples have the expectation of the
    // reference's S-way branching at every hit (src/Camera.cpp:184-194) without its S^depth ray tree
    const int sps = renderProperties.samplesPerScatter > 1 ? renderProperties.samplesPerScatter : 1;
    p.samples_per_pixel = renderProperties.samplesPerPixel * sps;
    p.max_depth = renderProperties.maxRayDepth;
    p.seed = seed;
    // lights feed Camera::computeDirectLighting (reference src/Camera.cpp:161-178) when enableLDirect is set
    p.flags = uploaded_.lights.empty() ? 0u : (uint32_t)RT_FLAG_DIRECT_LIGHT;
    p.strip_first = stripFirst;
    p.strip_stride = stripStride;
    static_assert(sizeof(RTPixelColor) == 4, "RTPixelColor must be 4 bytes of RGBA");
    if (group_) {
        if (stripFirst != 0 || stripStride != 1)
            throw BackendError(RT_ERR_INVALID, "a device list renders whole frames only");
        rt_group_timing t{};
        rc = rt_group_render_host(group_, &p, reinterpret_cast<uint8_t *>(rgba), &lastRaysTraced, &t);
        if (rc != RT_OK) fail(rc, "rt_group_render_host");
        lastKernelMs = t.kernel_ms_max;
    } else {
        rc = rt_render_host(ctx_, &p, reinterpret_cast<uint8_t *>(rgba), nullptr, &lastRaysTraced);
        if (rc != RT_OK) fail(rc, "rt_render_host");
        lastKernelMs = rt_last_kernel_ms(ctx_);
    }
    lastRenderMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void Camera::render(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                    std::vector<RTPixelColor> &outputBufer) {
    const int W = screenResolution.first, H = screenResolution.second;
    if (W <= 0 || H <= 0) throw BackendError(RT_ERR_INVALID, "Camera::render: resolution must be positive");
    // the reference asserts this (src/Camera.cpp:76); a short buffer cannot be rendered into
    if ((size_t)W * (size_t)H != outputBufer.size())
        throw BackendError(RT_ERR_INVALID, "Camera::render: outputBufer.size() != width*height");
    renderStrips(sceneManager, W, H, 0, 1, outputBufer.data());
}

void Camera::renderParallel(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                            std::vector<RTPixelColor> &outputBufer) {
    render(sceneManager, screenResolution, outputBufer);
}

void Camera::renderSerial(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                          std::vector<RTPixelColor> &outputBufer) {
    render(sceneManager, screenResolution, outputBufer);
}

// One pixel (reference src/Camera.cpp:105-122): the 8-row strip that holds it is rendered on the first
// device and the pixel picked; random streams are keyed on the pixel id, so the bytes are those of render().
RTPixelColor Camera::renderPixelColor(const SceneManager &sceneManager, const int pixelId,
                                      const std::pair<int, int> screenResolution) {
    const int W = screenResolution.first, H = screenResolution.second;
    if (W <= 0 || H <= 0) throw BackendError(RT_ERR_INVALID, "Camera::renderPixelColor: resolution must be positive");
    if (pixelId < 0 || (long long)pixelId >= (long long)W * H)
        throw BackendError(RT_ERR_INVALID, "Camera::renderPixelColor: pixelId outside the frame");
    const int x = pixelId % W, y = pixelId / W;
    const int strips = (H + RT_STRIP_ROWS - 1) / RT_STRIP_ROWS;
    std::vector<RTPixelColor> strip((size_t)RT_STRIP_ROWS * (size_t)W);
    const std::vector<int> list = devices;
    if (list.size() > 1) devices.assign(1, list[0]);
    try {
        // strip_first = the pixel's strip, stride = number of strips: exactly one strip is rendered
        renderStrips(sceneManager, W, H, y / RT_STRIP_ROWS, strips, strip.data());
    } catch (...) {
        devices = list;
        throw;
    }
    devices = list;
    return strip[(size_t)(y % RT_STRIP_ROWS) * (size_t)W + (size_t)x];
}

} // namespace rtamd
