// SceneManager: ownership and order of primitives and lights
// (reference src/RayTracer.cpp:8-66; its hitClosest, :68-100, runs on the GPU).
#include <algorithm>
#include <cassert>

#include "RayTracer.h"

namespace rtamd {

SceneManager::~SceneManager() { clear(); }

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

} // namespace rtamd
