// RTObjects.h — host-side primitives and lights with the reference's names and text format
// (reference inc/RTObjects.h:24-83 Primitives, :85-160 SphereObject, :162-218 PlaneObject, :220-321 Light,
// :323-540 PolygonObject, :543-645 CubeObject, :648-660 stream operators).
// hit() / hitExpanded() of every kind run on the GPU; these classes carry the parameters, the
// selection flag and the text (de)serialisation.
#ifndef RTAMD_RTOBJECTS_H
#define RTAMD_RTOBJECTS_H

#include <cassert>
#include <istream>
#include <ostream>
#include <string>
#include <vector>

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

    // "Type x y z selected", then the fields of the subclass (reference :39-47)
    virtual std::ostream &dump(std::ostream &stream) const {
        return stream << typeString() << ' ' << position().x() << ' ' << position().y() << ' ' << position().z() << ' '
                      << selected();
    }
    // reads what dump wrote after the type word (the caller consumed it to pick the class, :49-57)
    virtual std::istream &scan(std::istream &stream) {
        float x, y, z;
        bool sel;
        stream >> x >> y >> z >> sel;
        setPosition(gm::IPoint3(x, y, z));
        setSelectFlag(sel);
        return stream;
    }

    friend inline std::ostream &operator<<(std::ostream &os, const Primitives &p);
    friend inline std::istream &operator>>(std::istream &is, Primitives &p);

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

protected:
    std::ostream &dump(std::ostream &stream) const override {
        Primitives::dump(stream);
        return stream << ' ' << radius_;
    }
    std::istream &scan(std::istream &stream) override {
        Primitives::scan(stream);
        double r = 0.0;
        stream >> r;
        radius_ = r;
        return stream;
    }
};

// Infinite plane through position() (reference :162-218).  The constructor normalises the normal,
// setNormal and scan do not (:172,196,215).
class PlaneObject : public Primitives {
    gm::IVec3f normal_;

public:
    PlaneObject(const SceneManager *parent = nullptr) : Primitives(parent) {}
    PlaneObject(const gm::IPoint3 point, gm::IVec3f normal, RTMaterial *material, const SceneManager *parent = nullptr)
        : Primitives(material, parent), normal_(normal.normalized()) {
        position_ = point;
    }

    void setNormal(const gm::IVec3f normal) { normal_ = normal; }
    gm::IVec3f getNormal() const { return normal_; }
    std::string typeString() const override { return "Plane"; }

protected:
    std::ostream &dump(std::ostream &stream) const override {
        Primitives::dump(stream);
        return stream << ' ' << normal_.x() << ' ' << normal_.y() << ' ' << normal_.z();
    }
    std::istream &scan(std::istream &stream) override {
        Primitives::scan(stream);
        float nx, ny, nz;
        stream >> nx >> ny >> nz;
        normal_ = gm::IVec3f(nx, ny, nz);
        return stream;
    }
};

// Planar polygon (reference :323-540): normal from the first three vertices, position() is the
// centroid, setPosition() moves every vertex with it.
class PolygonObject : public Primitives {
    std::vector<gm::IPoint3> vertices_;
    gm::IVec3f normal_;
    gm::IPoint3 centroid_;

public:
    PolygonObject(const SceneManager *parent = nullptr) : Primitives(parent) {}
    PolygonObject(const std::vector<gm::IPoint3> &verts, RTMaterial *material, const SceneManager *parent = nullptr)
        : Primitives(material, parent), vertices_(verts) {
        computeNormalAndCentroid();
        if (!vertices_.empty()) position_ = vertices_[0];
    }

    void setVertices(const std::vector<gm::IPoint3> &verts) {
        vertices_ = verts;
        computeNormalAndCentroid();
        if (!vertices_.empty()) position_ = vertices_[0];
    }
    const std::vector<gm::IPoint3> &vertices() const { return vertices_; }
    gm::IVec3f normal() const { return normal_; }
    gm::IPoint3 centroid() const { return centroid_; }

    void setPosition(const gm::IPoint3 position) override {
        gm::IVec3f delta = position - centroid_;
        for (auto &v : vertices_) v = v + delta;
        centroid_ = position;
        position_ = position;
    }
    gm::IPoint3 position() const override { return centroid_; }
    std::string typeString() const override { return "Polygon"; }

protected:
    std::ostream &dump(std::ostream &stream) const override {
        Primitives::dump(stream);
        stream << ' ' << vertices_.size();
        for (const auto &v : vertices_) stream << ' ' << v.x() << ' ' << v.y() << ' ' << v.z();
        return stream;
    }
    std::istream &scan(std::istream &stream) override {
        Primitives::scan(stream);
        size_t n = 0;
        stream >> n;
        vertices_.clear();
        for (size_t i = 0; i < n && stream; ++i) {
            float x, y, z;
            stream >> x >> y >> z;
            vertices_.emplace_back(x, y, z);
        }
        computeNormalAndCentroid();
        return stream;
    }

private:
    // reference :413-442; the mean is taken with a single-precision 1/n there
    void computeNormalAndCentroid() {
        centroid_ = gm::IPoint3(0.0, 0.0, 0.0);
        normal_ = gm::IVec3f(0.0, 0.0, 1.0);
        if (vertices_.empty()) return;
        for (const auto &v : vertices_) centroid_ += v;
        centroid_ = centroid_ * (double)(1.0f / static_cast<float>(vertices_.size()));
        if (vertices_.size() >= 3)
            normal_ = cross(vertices_[1] - vertices_[0], vertices_[2] - vertices_[0]).normalized();
    }
};

// Axis-aligned box around position() (reference :543-645).
class CubeObject : public Primitives {
    gm::IVec3f halfSize_;

public:
    CubeObject(const SceneManager *parent = nullptr) : Primitives(parent), halfSize_(0.5, 0.5, 0.5) {}
    CubeObject(const gm::IVec3f &halfSize, RTMaterial *material, const SceneManager *parent = nullptr)
        : Primitives(material, parent), halfSize_(halfSize) {}

    void setHalfSize(const gm::IVec3f &hs) { halfSize_ = hs; }
    gm::IVec3f getHalfSize() const { return halfSize_; }
    std::string typeString() const override { return "Cube"; }

protected:
    std::ostream &dump(std::ostream &stream) const override {
        Primitives::dump(stream);
        return stream << ' ' << halfSize_.x() << ' ' << halfSize_.y() << ' ' << halfSize_.z();
    }
    std::istream &scan(std::istream &stream) override {
        Primitives::scan(stream);
        float hx, hy, hz;
        stream >> hx >> hy >> hz;
        halfSize_ = gm::IVec3f(hx, hy, hz);
        return stream;
    }
};

// Phong point light (reference inc/RTObjects.h:220-296).  getDirectLighting (:247-257, :300-320) runs on
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

    // "Light px py pz  ax ay az  dx dy dz  sx sy sz  pow" (reference :267-296)
    virtual std::ostream &dump(std::ostream &stream) const {
        auto put = [&stream](const gm::IVec3f &v) { stream << v.x() << ' ' << v.y() << ' ' << v.z() << ' '; };
        stream << typeString() << ' ';
        put(position_);
        put(ambientIntensity_);
        put(defuseIntensity_);
        put(specularIntensity_);
        return stream << viewLightPow_;
    }
    virtual std::istream &scan(std::istream &stream) {
        auto get = [&stream]() {
            float x, y, z;
            stream >> x >> y >> z;
            return gm::IVec3f(x, y, z);
        };
        position_ = get();
        ambientIntensity_ = get();
        defuseIntensity_ = get();
        specularIntensity_ = get();
        stream >> viewLightPow_;
        return stream;
    }
};

inline std::ostream &operator<<(std::ostream &os, const Primitives &p) { return p.dump(os); }
inline std::istream &operator>>(std::istream &is, Primitives &p) { return p.scan(is); }
inline std::ostream &operator<<(std::ostream &os, const Light &l) { return l.dump(os); }
inline std::istream &operator>>(std::istream &is, Light &l) { return l.scan(is); }

} // namespace rtamd
#endif
