// RTMaterial.h — host-side materials with the reference's names and factory
// (reference inc/RTMaterial.h:12-61, 63-96, 98-133, 135-198, 200-221, 223-248).
// scatter() itself runs on the GPU; these classes carry the parameters and add the const
// getters the reference lacks for its private members (RTMetal::fuzz_ :100,
// RTDielectric::specular_local_ / refractionIndex_ :136-137), which the flattener needs.
#ifndef RTAMD_RTMATERIAL_H
#define RTAMD_RTMATERIAL_H

#include <memory>
#include <string>
#include <vector>

#include "Vec3.h"

namespace rtamd {

using RTColor = gm::IVec3f;

enum class RTMaterialKind { Lambertian = 0, Metal = 1, Dielectric = 2, Emissive = 3 };

struct RTMaterial {
    virtual ~RTMaterial() = default;

    gm::IVec3f diffuse_ = gm::IVec3f(0, 0, 0);
    gm::IVec3f specular_ = gm::IVec3f(0, 0, 0);
    gm::IVec3f emission_ = gm::IVec3f(0, 0, 0);

    virtual RTMaterialKind kind() const = 0;
    virtual bool hasSpecular() const { return false; }
    virtual bool hasDiffuse() const { return false; }
    virtual bool hasEmmision() const { return false; }
    virtual std::string typeString() const { return "RTMaterial"; }

    gm::IVec3f &diffuse() { return diffuse_; }
    gm::IVec3f &specular() { return specular_; }
    gm::IVec3f &emitted() { return emission_; }
    const gm::IVec3f &diffuse() const { return diffuse_; }
    const gm::IVec3f &specular() const { return specular_; }
    const gm::IVec3f &emitted() const { return emission_; }
};

class RTLambertian : public RTMaterial {
public:
    explicit RTLambertian(const gm::IVec3f &diffuse) { diffuse_ = diffuse; }
    RTMaterialKind kind() const override { return RTMaterialKind::Lambertian; }
    std::string typeString() const override { return "Lambertian"; }
    bool hasDiffuse() const override { return true; }
};

class RTMetal : public RTMaterial {
    double fuzz_ = 0.0;

public:
    // fuzz is clamped to <= 1 exactly as the reference constructor does (inc/RTMaterial.h:103)
    RTMetal(const RTColor &specularColor, double fuzz) : fuzz_(fuzz < 1 ? fuzz : 1) { specular_ = specularColor; }
    RTMaterialKind kind() const override { return RTMaterialKind::Metal; }
    std::string typeString() const override { return "Metal"; }
    bool hasSpecular() const override { return true; }
    double fuzz() const { return fuzz_; }
};

class RTDielectric : public RTMaterial {
    gm::IVec3f specular_local_;
    double refractionIndex_ = 1.0;

public:
    RTDielectric(const gm::IVec3f specular, const double refractionIndex)
        : specular_local_(specular), refractionIndex_(refractionIndex) {}
    RTMaterialKind kind() const override { return RTMaterialKind::Dielectric; }
    std::string typeString() const override { return "Dielectric"; }
    bool hasSpecular() const override { return true; }
    const gm::IVec3f &tint() const { return specular_local_; } // the attenuation scatter() returns (:147)
    double refractionIndex() const { return refractionIndex_; }
};

class RTEmissive : public RTMaterial {
public:
    explicit RTEmissive(const gm::IVec3f &emission) { emission_ = emission; }
    RTMaterialKind kind() const override { return RTMaterialKind::Emissive; }
    std::string typeString() const override { return "Emissive"; }
    bool hasEmmision() const override { return true; }
};

// Owning factory, reference inc/RTMaterial.h:223-248.
class RTMaterialManager {
    std::vector<std::unique_ptr<RTMaterial>> children;

public:
    RTMaterial *MakeLambertian(const gm::IVec3f &diffuse) {
        children.push_back(std::make_unique<RTLambertian>(diffuse));
        return children.back().get();
    }
    RTMaterial *MakeMetal(const RTColor &specularColor, double fuzz) {
        children.push_back(std::make_unique<RTMetal>(specularColor, fuzz));
        return children.back().get();
    }
    RTMaterial *MakeDielectric(gm::IVec3f specular, double refractionIndex) {
        children.push_back(std::make_unique<RTDielectric>(specular, refractionIndex));
        return children.back().get();
    }
    RTMaterial *MakeEmissive(const gm::IVec3f &emission) {
        children.push_back(std::make_unique<RTEmissive>(emission));
        return children.back().get();
    }
};

} // namespace rtamd
#endif
