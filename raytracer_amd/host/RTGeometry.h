// RTGeometry.h — plain value types that cross the render path's host interface
// (reference inc/RTGeometry.h:36-90: Ray, HitRecord, Interval).  The reflection / refraction helpers
// of that header (:18-34) are evaluated on the GPU only (csrc/rt_device.h) and have no host copy.
#ifndef RTAMD_RTGEOMETRY_H
#define RTAMD_RTGEOMETRY_H

#include <limits>

#include "Vec3.h"

namespace rtamd {

struct RTMaterial;
class Primitives;

// Half-line origin + t * direction; the direction need not be unit length
// (SphereObject::hitDetail divides by its squared length, reference inc/RTObjects.h:136).
struct Ray {
    gm::IPoint3 origin;
    gm::IVec3f direction;

    Ray() = default;
    Ray(const gm::IPoint3 &from, const gm::IVec3f &along) : origin(from), direction(along) {}
    gm::IPoint3 at(double t) const { return origin + direction * t; }
};

// Open / closed ranges of the ray parameter.  hitClosest searches the OPEN range (surrounds).
struct Interval {
    double min = +std::numeric_limits<double>::infinity(); // default: empty
    double max = -std::numeric_limits<double>::infinity();

    Interval() = default;
    Interval(double lo, double hi) : min(lo), max(hi) {}

    bool surrounds(double x) const { return x > min && x < max; }
    bool contains(double x) const { return x >= min && x <= max; }
    double clamp(double x) const { return x < min ? min : (x > max ? max : x); }
    double size() const { return max - min; }

    static const Interval empty, universe;
};
inline const Interval Interval::empty{};
inline const Interval Interval::universe{-std::numeric_limits<double>::infinity(),
                                         +std::numeric_limits<double>::infinity()};

// Result of a closest-hit query (reference inc/RTGeometry.h:49-62).  `object` stays null for cubes
// and polygons, as in the reference (inc/RTObjects.h:507,642); `hitExpanded` marks a hit on the
// grown shell of a selected object.
struct HitRecord {
    double time = 0;
    gm::IPoint3 point{};
    gm::IVec3f normal{}; // unit, against the ray
    bool frontFace = false;
    bool hitExpanded = false;
    const RTMaterial *material = nullptr;
    const Primitives *object = nullptr;
};

} // namespace rtamd
#endif
