// RTObjects.h — host-side primitives with the reference's names
// (reference inc/RTObjects.h:24-83 Primitives, :85-160 SphereObject, :220-321 Light).
// hit() runs on the GPU.  Only the sphere is inside the accelerated path; scenes holding
// anything the path does not cover (selected objects) are rejected by Camera::render
// instead of being rendered differently from the reference.
#ifndef RTAMD_RTOBJECTS_H
#define RTAMD_RTOBJECTS_H

#include <cassert>
#include <string>

#include "RTMaterial.h"
#include "Vec3.h"

namespace rtamd {

class SceneManager;

class Primitives {
protected:
    const SceneManager *parent_{nullptr};
    RTMaterial *material_{nullptr};
    gm::IPoint3 position_{};
    bool selectFlag_ = false;

    Primitives(RTMaterial *material, const SceneManager *parent = nullptr) : parent_(parent), material_(material) {
        assert(material);
    }
    Primitives(const SceneManager *parent = nullptr) : parent_(parent) {}

public:
    virtual ~Primitives() = default;
    virtual std::string typeString() const { return "Primitive"; }

    virtual void setPosition(const gm::IPoint3 position) { position_ = position; }
    virtual gm::IPoint3 position() const { return position_; }

    void setMaterial(RTMaterial *material) { material_ = material; }
    const RTMaterial *material() const { return material_; }
    RTMaterial *material() { return material_; }

    void setSelectFlag(bool val) { selectFlag_ = val; }
    bool selected() const { return selectFlag_; }

    friend class SceneManager;
};

class SphereObject : public Primitives {
    double radius_ = 0.0;

public:
    SphereObject(const SceneManager *parent = nullptr) : Primitives(parent) {}
    SphereObject(double radius, RTMaterial *material, const SceneManager *parent = nullptr)
        : Primitives(material, parent), radius_(radius) {}

    float getRadius() const { return radius_; } // float, as in the reference (:107)
    void setRadius(const float val) { radius_ = val; }
    std::string typeString() const override { return "Sphere"; }
};

// Phong point light (reference inc/RTObjects.h:220-257).  getDirectLighting (:247-257, :300-320) runs on
// the GPU; the const getters below are what the flattener reads (the reference keeps the fields private).
class Light {
    gm::IVec3f ambientIntensity_, defuseIntensity_, specularIntensity_;
    gm::IPoint3 position_;
    double viewLightPow_ = 1.0;
    const SceneManager *parent_ = nullptr;

public:
    Light(const SceneManager *parent = nullptr) : parent_(parent) {}
    Light(const gm::IVec3f ambientIntensity, const gm::IVec3f defuseIntensity, const gm::IVec3f specularIntensity,
          double viewLightPow, const SceneManager *parent = nullptr)
        : ambientIntensity_(ambientIntensity), defuseIntensity_(defuseIntensity),
          specularIntensity_(specularIntensity), viewLightPow_(viewLightPow), parent_(parent) {}
    virtual ~Light() = default;
    void setPosition(const gm::IPoint3 position) { position_ = position; }
    gm::IPoint3 position() const { return position_; }
    const SceneManager *parent() const { return parent_; }
    void setParent(const SceneManager *parent) { parent_ = parent; }
    virtual std::string typeString() const { return "Light"; }

    const gm::IVec3f &ambientIntensity() const { return ambientIntensity_; }
    const gm::IVec3f &defuseIntensity() const { return defuseIntensity_; }
    const gm::IVec3f &specularIntensity() const { return specularIntensity_; }
    double viewLightPow() const { return viewLightPow_; }
};

} // namespace rtamd
#endif
