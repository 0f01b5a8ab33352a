// RTGeometry.h — the plain value types of the render path (reference inc/RTGeometry.h:36-90).
#ifndef RTAMD_RTGEOMETRY_H
#define RTAMD_RTGEOMETRY_H

#include <limits>

#include "Vec3.h"

namespace rtamd {

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
};

} // namespace rtamd
#endif
