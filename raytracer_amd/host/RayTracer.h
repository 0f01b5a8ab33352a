// RayTracer.h — SceneManager, the scene container (reference inc/RayTracer.h:8-33,
// src/RayTracer.cpp:8-66).  hitClosest (src/RayTracer.cpp:68-100) runs on the GPU; what stays
// here is ownership and insertion order, which is the object id and the tie-break order.
#ifndef RTAMD_RAYTRACER_H
#define RTAMD_RAYTRACER_H

#include <vector>

#include "RTObjects.h"

namespace rtamd {

class SceneManager {
    std::vector<Primitives *> primitives_;
    std::vector<Light *> directLightSources_;

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

    const std::vector<Light *> &inderectLightSources() const { return directLightSources_; }
    std::vector<Primitives *> &primitives() { return primitives_; }
    std::vector<Light *> &lights() { return directLightSources_; }
    const std::vector<Primitives *> &primitives() const { return primitives_; }
    const std::vector<Light *> &lights() const { return directLightSources_; }
};

} // namespace rtamd
#endif
