// SceneManager: ownership and order of primitives and lights (reference src/RayTracer.cpp:8-66),
// and hitClosest (:68-100) as a client of the HIP library's first-hit entry point.
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <string>

#include "RayTracer.h"

namespace rtamd {

SceneManager::~SceneManager() {
    clear();
    if (ctx_) rt_destroy(ctx_);
}

uint64_t SceneManager::deviceAllocations() const { return ctx_ ? rt_device_allocations(ctx_) : 0; }

void SceneManager::addObject(gm::IPoint3 position, Primitives *object) {
    assert(object);
    assert((object->parent_ == this || object->parent_ == nullptr) && "addObject failed : object parent != this or nullptr");
    object->parent_ = this;
    object->position_ = position; // overwrites the primitive's own position, as the reference does (:26)
    primitives_.push_back(object);
}

void SceneManager::addObject(Primitives *object) { addObject(object->position(), object); }

void SceneManager::eraseObject(Primitives *primitive) {
    auto it = std::find(primitives_.begin(), primitives_.end(), primitive);
    if (it != primitives_.end()) primitives_.erase(it);
}

void SceneManager::addLight(gm::IPoint3 position, Light *light) {
    assert(light);
    assert((light->parent() == this || light->parent() == nullptr) && "addLight failed : light parent != this or nullptr");
    light->setParent(this);
    light->setPosition(position);
    directLightSources_.push_back(light);
}

void SceneManager::addLight(Light *light) { addLight(light->position(), light); }

void SceneManager::clear() {
    for (Primitives *p : primitives_) delete p;
    for (Light *l : directLightSources_) delete l;
    primitives_.clear();
    directLightSources_.clear();
}

void SceneManager::hitClosest(const std::vector<Ray> &rays, Interval rayTime, bool hitExpandedState,
                              std::vector<HitRecord> &records, std::vector<char> &hit) const {
    const size_t n = rays.size();
    records.assign(n, HitRecord{});
    hit.assign(n, 0);
    if (n == 0) return;
    if (n > 0x7fffffffu) throw BackendError(RT_ERR_INVALID, "SceneManager::hitClosest: too many rays");
    FlatScene flat = flattenScene(*this, false);
    if (!ctx_ || ctxDevice_ != device) {
        if (ctx_) rt_destroy(ctx_);
        ctx_ = nullptr;
        haveUploaded_ = false;
        int rc = rt_create(device, &ctx_);
        if (rc != RT_OK) throw BackendError(rc, "rt_create failed (no usable GPU?)");
        ctxDevice_ = device;
    }
    if (!(haveUploaded_ && sameFlatScene(flat, uploaded_))) {
        haveUploaded_ = false;
        uploadScene(ctx_, flat);
        uploaded_ = std::move(flat);
        haveUploaded_ = true;
    }
    std::vector<float> org(3 * n), dir(3 * n), t(n), nrm(3 * n);
    std::vector<int32_t> id(n);
    std::vector<uint8_t> front(n), expanded(n);
    for (size_t i = 0; i < n; ++i) {
        org[3 * i] = (float)rays[i].origin.x();
        org[3 * i + 1] = (float)rays[i].origin.y();
        org[3 * i + 2] = (float)rays[i].origin.z();
        dir[3 * i] = (float)rays[i].direction.x();
        dir[3 * i + 1] = (float)rays[i].direction.y();
        dir[3 * i + 2] = (float)rays[i].direction.z();
    }
    const uint32_t flags = RT_FH_REFINE | (hitExpandedState ? (uint32_t)RT_FH_EXPANDED : 0u);
    int rc = rt_first_hit_ex_host(ctx_, (int)n, org.data(), dir.data(), (float)rayTime.min, flags, id.data(), t.data(),
                                  nrm.data(), front.data(), expanded.data());
    if (rc != RT_OK) throw BackendError(rc, std::string("rt_first_hit_ex_host: ") + rt_last_error(ctx_));
    for (size_t i = 0; i < n; ++i) {
        // the GPU searches (min, +inf); a closest hit at or beyond max means nothing lies inside the
        // interval, because a primitive's hit time does not depend on the upper end that admits it
        if (id[i] < 0 || !((double)t[i] < rayTime.max)) continue;
        const Primitives *p = primitives_[(size_t)id[i]];
        HitRecord &r = records[i];
        r.time = t[i];
        r.point = rays[i].at(r.time);
        r.normal = gm::IVec3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
        r.frontFace = front[i] != 0;
        r.hitExpanded = expanded[i] != 0;
        r.material = p->material();
        // cubes and polygons leave the object pointer null (reference inc/RTObjects.h:507,642)
        const rt_prim &fp = uploaded_.prims[(size_t)id[i]];
        r.object = (fp.type == RT_PRIM_CUBE || fp.type == RT_PRIM_POLYGON) ? nullptr : p;
        hit[i] = 1;
    }
}

bool SceneManager::hitClosest(const Ray &ray, Interval rayTime, HitRecord &hitRecord, bool hitExpandedState) const {
    std::vector<HitRecord> rec;
    std::vector<char> hit;
    hitClosest(std::vector<Ray>{ray}, rayTime, hitExpandedState, rec, hit);
    hitRecord = rec[0];
    return hit[0] != 0;
}

} // namespace rtamd
