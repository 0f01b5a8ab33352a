// Camera: pose / viewport on the host (reference src/Camera.cpp:38-52, 206-236) and
// render() as a client of the HIP library's C ABI (include/rt_amd.h).
#include <cmath>
#include <ostream>
#include <string>

#include "Backend.h"
#include "Camera.h"
#include "RayTracer.h"

namespace rtamd {

Camera::Camera() { updateViewPort(); }

Camera::~Camera() {
    if (ctx_) rt_destroy(ctx_);
}

void Camera::move(const gm::IVec3f motionVec) {
    center_ = center_ + motionVec;
    updateViewPort();
}

void Camera::rotate(const double widthRadians, const double heightRadians) {
    direction_.rotate(viewPort_.downDir_, -widthRadians);
    direction_.rotate(viewPort_.rightDir_, heightRadians);
    updateViewPort();
}

// reference src/Camera.cpp:206-221: world-up is +z unless the view direction is within 0.999 of it,
// then +y; the viewport is 1x1 at focal length 1.
void Camera::updateViewPort() {
    gm::IVec3f worldUp(0.0, 0.0, 1.0);
    if (std::abs(gm::dot(direction_, worldUp)) > 0.999) worldUp = gm::IVec3f(0.0, 1.0, 0.0);
    viewPort_.rightDir_ = cross(direction_, worldUp).normalized();
    viewPort_.downDir_ = cross(direction_, viewPort_.rightDir_).normalized();
    gm::IVec3f rightFull = viewPort_.rightDir_ * Viewport::VIEWPORT_WIDTH;
    gm::IVec3f downFull = viewPort_.downDir_ * Viewport::VIEWPORT_HEIGHT;
    viewPort_.upperLeft_ = center_ + direction_ * FOCAL_LENGTH - rightFull * 0.5 - downFull * 0.5;
    viewPort_.viewAngle_.setX(2 * std::atan(Viewport::VIEWPORT_HEIGHT / FOCAL_LENGTH));
    viewPort_.viewAngle_.setY(2 * std::atan(Viewport::VIEWPORT_WIDTH / FOCAL_LENGTH));
}

gm::IVec3f Camera::direction() const { return direction_; }
gm::IPoint3 Camera::center() const { return center_; }
Viewport Camera::viewPort() const { return viewPort_; }

void Camera::setCenter(const gm::IPoint3 center) {
    center_ = center;
    updateViewPort();
}
void Camera::setDirection(const gm::IVec3f direction) {
    direction_ = direction.normalized();
    updateViewPort();
}

std::ostream &operator<<(std::ostream &stream, const Camera &camera) {
    return stream << "Camera{" << camera.center() << ", " << camera.direction() << "}";
}

namespace {
void put3(float *dst, const gm::IVec3f &v) {
    dst[0] = (float)v.x();
    dst[1] = (float)v.y();
    dst[2] = (float)v.z();
}

[[noreturn]] void fail(rt_ctx *ctx, int status, const std::string &where) {
    throw BackendError(status, where + ": " + (ctx ? rt_last_error(ctx) : "no context"));
}
} // namespace

// Uploads scene and pose and renders strips strip_first, strip_first + strip_stride, ... compactly into
// `rgba` (rt_output_rows x width pixels).
void Camera::renderStrips(const SceneManager &sceneManager, int W, int H, int stripFirst, int stripStride,
                          RTPixelColor *rgba) {
    // flatten first: a scene the GPU path cannot take is rejected before any device is touched
    const FlatScene flat = flattenScene(sceneManager, renderProperties.enableLDirect);

    if (!ctx_ || ctxDevice_ != device) {
        if (ctx_) rt_destroy(ctx_);
        ctx_ = nullptr;
        int rc = rt_create(device, &ctx_);
        if (rc != RT_OK) throw BackendError(rc, "rt_create failed (no usable GPU?)");
        ctxDevice_ = device;
    }
    uploadScene(ctx_, flat);

    rt_camera cam{};
    put3(cam.center, center_);
    put3(cam.upper_left, viewPort_.upperLeft_);
    put3(cam.right, viewPort_.rightDir_);
    put3(cam.down, viewPort_.downDir_);
    int rc = rt_set_camera(ctx_, &cam);
    if (rc != RT_OK) fail(ctx_, rc, "rt_set_camera");

    rt_params p{};
    p.width = W;
    p.height = H;
    const int sps = renderProperties.samplesPerScatter > 1 ? renderProperties.samplesPerScatter : 1;
    p.samples_per_pixel = renderProperties.samplesPerPixel * sps;
    p.max_depth = renderProperties.maxRayDepth;
    p.seed = seed;
    // lights feed Camera::computeDirectLighting (reference src/Camera.cpp:161-178) when enableLDirect is set
    p.flags = flat.lights.empty() ? 0u : (uint32_t)RT_FLAG_DIRECT_LIGHT;
    p.strip_first = stripFirst;
    p.strip_stride = stripStride;
    static_assert(sizeof(RTPixelColor) == 4, "RTPixelColor must be 4 bytes of RGBA");
    rc = rt_render_host(ctx_, &p, reinterpret_cast<uint8_t *>(rgba), nullptr, &lastRaysTraced);
    if (rc != RT_OK) fail(ctx_, rc, "rt_render_host");
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

// The reference's two loops over pixels (src/Camera.cpp:69-103) differ in threading and in whether the
// generator is re-seeded per pixel; both are the same GPU render here.
void Camera::renderParallel(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                            std::vector<RTPixelColor> &outputBufer) {
    render(sceneManager, screenResolution, outputBufer);
}

void Camera::renderSerial(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                          std::vector<RTPixelColor> &outputBufer) {
    render(sceneManager, screenResolution, outputBufer);
}

// One pixel (reference src/Camera.cpp:105-122): the 8-row strip that holds it is rendered and the pixel
// picked; its bytes equal the same pixel of a full render (random streams are keyed on the pixel id).
RTPixelColor Camera::renderPixelColor(const SceneManager &sceneManager, const int pixelId,
                                      const std::pair<int, int> screenResolution) {
    const int W = screenResolution.first, H = screenResolution.second;
    if (W <= 0 || H <= 0) throw BackendError(RT_ERR_INVALID, "Camera::renderPixelColor: resolution must be positive");
    if (pixelId < 0 || (long long)pixelId >= (long long)W * H)
        throw BackendError(RT_ERR_INVALID, "Camera::renderPixelColor: pixelId outside the frame");
    const int x = pixelId % W, y = pixelId / W;
    const int strips = (H + RT_STRIP_ROWS - 1) / RT_STRIP_ROWS;
    std::vector<RTPixelColor> strip((size_t)RT_STRIP_ROWS * (size_t)W);
    // strip_first = the pixel's strip, stride = number of strips: exactly one strip is rendered
    renderStrips(sceneManager, W, H, y / RT_STRIP_ROWS, strips, strip.data());
    return strip[(size_t)(y % RT_STRIP_ROWS) * (size_t)W + (size_t)x];
}

} // namespace rtamd
