// Camera: pose / viewport on the host (reference src/Camera.cpp:38-52, 206-236) and
// render() as a client of the HIP library's C ABI (include/rt_amd.h).
#include <cmath>
#include <map>
#include <ostream>
#include <string>

#include "../../include/rt_amd.h"
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

void Camera::render(const SceneManager &sceneManager, const std::pair<int, int> screenResolution,
                    std::vector<RTPixelColor> &outputBufer) {
    const int W = screenResolution.first, H = screenResolution.second;
    if (W <= 0 || H <= 0) throw BackendError(RT_ERR_INVALID, "Camera::render: resolution must be positive");
    // the reference asserts this (src/Camera.cpp:76); a short buffer cannot be rendered into
    if ((size_t)W * (size_t)H != outputBufer.size())
        throw BackendError(RT_ERR_INVALID, "Camera::render: outputBufer.size() != width*height");

    // ---- flatten: spheres in insertion order, materials de-duplicated by pointer ----
    std::vector<rt_sphere> spheres;
    std::vector<uint32_t> matId;
    std::vector<rt_material> mats;
    std::map<const RTMaterial *, uint32_t> seen;
    for (const Primitives *p : sceneManager.primitives()) {
        const SphereObject *s = dynamic_cast<const SphereObject *>(p);
        if (!s) throw BackendError(RT_ERR_UNSUPPORTED, "Camera::render: primitive '" + p->typeString() +
                                                           "' is outside the GPU path (spheres only)");
        if (p->selected())
            throw BackendError(RT_ERR_UNSUPPORTED, "Camera::render: selected objects (halo) are outside the GPU path");
        const RTMaterial *m = p->material();
        if (!m) throw BackendError(RT_ERR_INVALID, "Camera::render: primitive without material");
        auto it = seen.find(m);
        if (it == seen.end()) {
            rt_material fm{};
            fm.type = (uint32_t)m->kind();
            gm::IVec3f albedo;
            switch (m->kind()) {
            case RTMaterialKind::Lambertian: albedo = m->diffuse(); break;
            case RTMaterialKind::Metal:
                albedo = m->specular();
                fm.fuzz = (float)static_cast<const RTMetal *>(m)->fuzz();
                break;
            case RTMaterialKind::Dielectric:
                albedo = static_cast<const RTDielectric *>(m)->tint();
                fm.ior = (float)static_cast<const RTDielectric *>(m)->refractionIndex();
                break;
            case RTMaterialKind::Emissive: break;
            }
            put3(fm.albedo, albedo);
            put3(fm.emission, m->emitted());
            put3(fm.diffuse, m->diffuse()); // what Light::getDirectLighting multiplies by
            it = seen.emplace(m, (uint32_t)mats.size()).first;
            mats.push_back(fm);
        }
        gm::IPoint3 c = p->position();
        spheres.push_back(rt_sphere{(float)c.x(), (float)c.y(), (float)c.z(), s->getRadius()});
        matId.push_back(it->second);
    }

    if (!ctx_ || ctxDevice_ != device) {
        if (ctx_) rt_destroy(ctx_);
        ctx_ = nullptr;
        int rc = rt_create(device, &ctx_);
        if (rc != RT_OK) throw BackendError(rc, "rt_create failed (no usable GPU?)");
        ctxDevice_ = device;
    }
    int rc = rt_upload_scene(ctx_, (int)spheres.size(), spheres.data(), matId.data(), (int)mats.size(), mats.data());
    if (rc != RT_OK) fail(ctx_, rc, "rt_upload_scene");

    // lights feed Camera::computeDirectLighting (reference src/Camera.cpp:161-178) when enableLDirect is set
    std::vector<rt_light> lights;
    if (renderProperties.enableLDirect) {
        for (const Light *l : sceneManager.lights()) {
            rt_light fl{};
            put3(fl.position, l->position());
            put3(fl.ambient, l->ambientIntensity());
            put3(fl.diffuse, l->defuseIntensity());
            put3(fl.specular, l->specularIntensity());
            fl.view_pow = (float)l->viewLightPow();
            lights.push_back(fl);
        }
    }
    rc = rt_upload_lights(ctx_, (int)lights.size(), lights.data());
    if (rc != RT_OK) fail(ctx_, rc, "rt_upload_lights");

    rt_camera cam{};
    put3(cam.center, center_);
    put3(cam.upper_left, viewPort_.upperLeft_);
    put3(cam.right, viewPort_.rightDir_);
    put3(cam.down, viewPort_.downDir_);
    rc = rt_set_camera(ctx_, &cam);
    if (rc != RT_OK) fail(ctx_, rc, "rt_set_camera");

    rt_params p{};
    p.width = W;
    p.height = H;
    const int sps = renderProperties.samplesPerScatter > 1 ? renderProperties.samplesPerScatter : 1;
    p.samples_per_pixel = renderProperties.samplesPerPixel * sps;
    p.max_depth = renderProperties.maxRayDepth;
    p.seed = seed;
    p.flags = lights.empty() ? 0u : (uint32_t)RT_FLAG_DIRECT_LIGHT;
    p.strip_first = 0;
    p.strip_stride = 1;
    static_assert(sizeof(RTPixelColor) == 4, "RTPixelColor must be 4 bytes of RGBA");
    rc = rt_render_host(ctx_, &p, reinterpret_cast<uint8_t *>(outputBufer.data()), nullptr, &lastRaysTraced);
    if (rc != RT_OK) fail(ctx_, rc, "rt_render_host");
}

} // namespace rtamd
