// RTMaterial.h — host-side materials with the reference's names and factory
// (reference inc/RTMaterial.h:12-61, 63-96, 98-133, 135-198, 200-221, 223-280).
// scatter() itself runs on the GPU; these classes carry the parameters, the text
// (de)serialisation (dump / scan / deserializeMaterial, :40-57,122-132,177-190,250-272) and the const
// getters the reference lacks for its private members (RTMetal::fuzz_ :100,
// RTDielectric::specular_local_ / refractionIndex_ :136-137), which the flattener needs.
#ifndef RTAMD_RTMATERIAL_H
#define RTAMD_RTMATERIAL_H

#include <istream>
#include <memory>
#include <ostream>
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

protected:
    // "Type d d d s s s e e e", then the fields of the subclass (reference :40-46)
    virtual std::ostream &dump(std::ostream &os) const {
        return os << typeString() << ' ' << diffuse_.x() << ' ' << diffuse_.y() << ' ' << diffuse_.z() << ' '
                  << specular_.x() << ' ' << specular_.y() << ' ' << specular_.z() << ' ' << emission_.x() << ' '
                  << emission_.y() << ' ' << emission_.z();
    }
    // reads what dump wrote after the type word (:48-57)
    virtual std::istream &scan(std::istream &is) {
        float v[9];
        for (float &f : v) is >> f;
        diffuse_ = gm::IVec3f(v[0], v[1], v[2]);
        specular_ = gm::IVec3f(v[3], v[4], v[5]);
        emission_ = gm::IVec3f(v[6], v[7], v[8]);
        return is;
    }

    friend inline std::ostream &operator<<(std::ostream &os, const RTMaterial &m);
    friend inline std::istream &operator>>(std::istream &is, RTMaterial &m);
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

protected:
    std::ostream &dump(std::ostream &os) const override {
        RTMaterial::dump(os);
        return os << ' ' << fuzz_;
    }
    // the scanned fuzz is taken as written, without the constructor's clamp (reference :128-132)
    std::istream &scan(std::istream &is) override {
        RTMaterial::scan(is);
        return is >> fuzz_;
    }
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

protected:
    std::ostream &dump(std::ostream &os) const override {
        RTMaterial::dump(os);
        return os << ' ' << specular_local_.x() << ' ' << specular_local_.y() << ' ' << specular_local_.z() << ' '
                  << refractionIndex_;
    }
    std::istream &scan(std::istream &is) override {
        RTMaterial::scan(is);
        float sx, sy, sz;
        is >> sx >> sy >> sz >> refractionIndex_;
        specular_local_ = gm::IVec3f(sx, sy, sz);
        return is;
    }
};

class RTEmissive : public RTMaterial {
public:
    explicit RTEmissive(const gm::IVec3f &emission) { emission_ = emission; }
    RTMaterialKind kind() const override { return RTMaterialKind::Emissive; }
    std::string typeString() const override { return "Emissive"; }
    // hasEmmision() stays false, as in the reference, which does not override it (inc/RTMaterial.h:28,200-221)
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

    // Reads "Type fields..." as written by operator<<; nullptr for an unknown type word
    // (reference :250-272).
    RTMaterial *deserializeMaterial(std::istream &iss);
};

inline std::ostream &operator<<(std::ostream &os, const RTMaterial &m) { return m.dump(os); }
inline std::istream &operator>>(std::istream &is, RTMaterial &m) { return m.scan(is); }

inline RTMaterial *RTMaterialManager::deserializeMaterial(std::istream &iss) {
    std::string type;
    iss >> type;
    std::unique_ptr<RTMaterial> mat;
    if (type == "Lambertian") mat = std::make_unique<RTLambertian>(gm::IVec3f{});
    else if (type == "Metal") mat = std::make_unique<RTMetal>(gm::IVec3f{}, 0.0);
    else if (type == "Dielectric") mat = std::make_unique<RTDielectric>(gm::IVec3f{}, 1.0);
    else if (type == "Emissive") mat = std::make_unique<RTEmissive>(gm::IVec3f{});
    else return nullptr;
    iss >> *mat;
    children.push_back(std::move(mat));
    return children.back().get();
}

} // namespace rtamd
#endif
