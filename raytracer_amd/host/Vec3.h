// Vec3.h — the small vector type of the host layer.
//
// The reference takes gm::IVec3f / gm::IPoint3 from an external library ("GeomLib",
// reference CMakeLists.txt:20) that is not part of its tree.  This host layer carries its
// own double-precision 3-vector with the members the render path's callers use
// (x(), y(), z(), setX.., normalized(), length2(), rotate()), and exposes it under the
// names rtamd::gm::IVec3f / IPoint3 so that scene-building code written against the
// reference compiles unchanged after `using namespace rtamd;`.
#ifndef RTAMD_VEC3_H
#define RTAMD_VEC3_H

#include <cmath>
#include <ostream>

namespace rtamd {

class Vec3 {
    double v_[3];

public:
    Vec3() : v_{0, 0, 0} {}
    Vec3(double s) : v_{s, s, s} {}
    Vec3(double x, double y, double z) : v_{x, y, z} {}

    double x() const { return v_[0]; }
    double y() const { return v_[1]; }
    double z() const { return v_[2]; }
    void setX(double x) { v_[0] = x; }
    void setY(double y) { v_[1] = y; }
    void setZ(double z) { v_[2] = z; }

    double length2() const { return v_[0] * v_[0] + v_[1] * v_[1] + v_[2] * v_[2]; }
    double length() const { return std::sqrt(length2()); }
    Vec3 normalized() const {
        double l = length();
        return Vec3(v_[0] / l, v_[1] / l, v_[2] / l);
    }
    bool nearZero() const { return std::fabs(v_[0]) < 1e-8 && std::fabs(v_[1]) < 1e-8 && std::fabs(v_[2]) < 1e-8; }

    Vec3 &operator+=(const Vec3 &o) {
        v_[0] += o.v_[0]; v_[1] += o.v_[1]; v_[2] += o.v_[2];
        return *this;
    }

    // Rotates this vector about `axis` by `radians` (right-handed, Rodrigues), in place:
    // the use made of it by Camera::rotate (reference src/Camera.cpp:48-52).
    void rotate(const Vec3 &axis, double radians);
};

inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a.x() + b.x(), a.y() + b.y(), a.z() + b.z()); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a.x() - b.x(), a.y() - b.y(), a.z() - b.z()); }
inline Vec3 operator*(const Vec3 &a, double s) { return Vec3(a.x() * s, a.y() * s, a.z() * s); }
inline Vec3 operator*(const Vec3 &a, const Vec3 &b) { return Vec3(a.x() * b.x(), a.y() * b.y(), a.z() * b.z()); }
inline Vec3 operator/(const Vec3 &a, double s) { return Vec3(a.x() / s, a.y() / s, a.z() / s); }
inline double dot(const Vec3 &a, const Vec3 &b) { return a.x() * b.x() + a.y() * b.y() + a.z() * b.z(); }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return Vec3(a.y() * b.z() - a.z() * b.y(), a.z() * b.x() - a.x() * b.z(), a.x() * b.y() - a.y() * b.x());
}
inline std::ostream &operator<<(std::ostream &os, const Vec3 &v) {
    return os << '(' << v.x() << ", " << v.y() << ", " << v.z() << ')';
}

inline void Vec3::rotate(const Vec3 &axis, double radians) {
    Vec3 k = axis.normalized();
    double c = std::cos(radians), s = std::sin(radians);
    Vec3 r = (*this) * c + cross(k, *this) * s + k * (dot(k, *this) * (1.0 - c));
    *this = r;
}

struct Vec2 {
    double x_ = 0, y_ = 0;
    void setX(double x) { x_ = x; }
    void setY(double y) { y_ = y; }
    double x() const { return x_; }
    double y() const { return y_; }
};

namespace gm {
using IVec3f = ::rtamd::Vec3;
using IPoint3 = ::rtamd::Vec3;
using IVec2f = ::rtamd::Vec2;
using ::rtamd::cross;
using ::rtamd::dot;
} // namespace gm

} // namespace rtamd
#endif
