// RayTracer.h — SceneManager, the scene container (reference inc/RayTracer.h:8-33,
// src/RayTracer.cpp:8-66).  What stays on the host is ownership and insertion order, which is
// the object id and the tie-break order; hitClosest (src/RayTracer.cpp:68-100) keeps its
// signature and runs on the GPU.
#ifndef RTAMD_RAYTRACER_H
#define RTAMD_RAYTRACER_H

#include <vector>

#include "Backend.h"
#include "RTGeometry.h"
#include "RTObjects.h"

namespace rtamd {

class SceneManager {
    std::vector<Primitives *> primitives_;
    std::vector<Light *> directLightSources_;
    // device side of hitClosest: created by the first query, re-uploaded only when the scene changed
    mutable rt_ctx *ctx_ = nullptr;
    mutable int ctxDevice_ = -1;
    mutable FlatScene uploaded_;
    mutable bool haveUploaded_ = false;

public:
    SceneManager() = default;
    SceneManager(const SceneManager &) = delete;
    SceneManager &operator=(const SceneManager &) = delete;
    ~SceneManager();

    void addObject(const gm::IPoint3 position, Primitives *object);
    void eraseObject(Primitives *primitive); // unlinks without deleting, as the reference does
    void addLight(const gm::IPoint3 position, Light *light);
    void addObject(Primitives *object);
    void addLight(Light *light);
    void clear();

    // Closest hit of one ray, with the reference's signature and record (src/RayTracer.cpp:68-100):
    // one ray is traced on the GPU with the double-precision refinement on, so the primitive picked is
    // the one a double evaluation picks.  The scene is flattened on every call (that is how edits are
    // noticed) but uploaded only when it differs from what the device holds.
    // Meant for picking (a mouse click), not for loops: use the batch form for many rays.
    // Throws BackendError when no GPU is usable or the scene holds an unknown primitive kind.
    bool hitClosest(const Ray &ray, Interval rayTime, HitRecord &hitRecord, bool hitExpandedState) const;
    // Batch form (not in the reference): one upload, one launch; hit[i] tells whether records[i] is a hit.
    void hitClosest(const std::vector<Ray> &rays, Interval rayTime, bool hitExpandedState,
                    std::vector<HitRecord> &records, std::vector<char> &hit) const;
    int device = 0; // which GPU hitClosest uses (not in the reference)
    uint64_t deviceAllocations() const; // hipMalloc calls made for hitClosest so far (not in the reference)

    const std::vector<Light *> &inderectLightSources() const { return directLightSources_; }
    std::vector<Primitives *> &primitives() { return primitives_; }
    std::vector<Light *> &lights() { return directLightSources_; }
    const std::vector<Primitives *> &primitives() const { return primitives_; }
    const std::vector<Light *> &lights() const { return directLightSources_; }
};

} // namespace rtamd
#endif
