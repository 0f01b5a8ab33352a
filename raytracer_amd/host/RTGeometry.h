// RTGeometry.h — the plain value types of the render path (reference inc/RTGeometry.h:36-90).
#ifndef RTAMD_RTGEOMETRY_H
#define RTAMD_RTGEOMETRY_H

#include <algorithm>
#include <cmath>
#include <limits>

#include "Vec3.h"

namespace rtamd {

// Mirror reflection and Snell refraction as the materials use them (reference inc/RTGeometry.h:18-34),
// for host code that wants the same formulas; the render path evaluates them on the GPU.
inline gm::IVec3f reflect(const gm::IVec3f &v, const gm::IVec3f &n) { return v - n * 2 * dot(v, n); }

inline gm::IVec3f refract(const gm::IVec3f &uv, const gm::IVec3f &n, double refractionCoef) {
    gm::IVec3f unitUv = uv.normalized();
    double cosTheta = std::fmin(dot(unitUv * (-1), n), 1.0);
    gm::IVec3f rOutPerp = (unitUv + n * cosTheta) * refractionCoef;
    // the reference rescales an over-long perpendicular part by 1/coef^2 (:28) and then asserts (:30);
    // here the square root is clamped at zero instead of asserting
    if (1.0 - rOutPerp.length2() < 0) rOutPerp = rOutPerp * (1.0 / (refractionCoef * refractionCoef));
    gm::IVec3f rOutParallel = n * (-std::sqrt(std::max(0.0, 1.0 - rOutPerp.length2())));
    return rOutPerp + rOutParallel;
}

struct Ray {
    gm::IPoint3 origin;
    gm::IVec3f direction;
    Ray() {}
    Ray(const gm::IPoint3 &origin, const gm::IVec3f &direction) : origin(origin), direction(direction) {}
    gm::IPoint3 at(double t) const { return origin + direction * t; }
};

struct RTMaterial;
class Primitives;

// What a closest-hit query reports (reference inc/RTGeometry.h:49-62).  `object` is null for cubes
// and polygons, as in the reference (inc/RTObjects.h:507,642).
struct HitRecord {
    gm::IPoint3 point = {};
    gm::IVec3f normal = {};
    const RTMaterial *material = nullptr;
    const Primitives *object = nullptr;
    bool frontFace = false;
    double time = 0;
    bool hitExpanded = false;
};

struct Interval {
    double min, max;
    Interval() : min(+std::numeric_limits<double>::infinity()), max(-std::numeric_limits<double>::infinity()) {}
    Interval(double min, double max) : min(min), max(max) {}
    double size() const { return max - min; }
    bool contains(double x) const { return min <= x && x <= max; }
    bool surrounds(double x) const { return min < x && x < max; }
    double clamp(double x) const { return x < min ? min : (x > max ? max : x); }
    static const Interval empty, universe; // as in the reference (inc/RTGeometry.h:89)
};
inline const Interval Interval::empty = Interval();
inline const Interval Interval::universe =
    Interval(-std::numeric_limits<double>::infinity(), +std::numeric_limits<double>::infinity());

} // namespace rtamd
#endif
