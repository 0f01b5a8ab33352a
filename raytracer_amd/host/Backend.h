// Backend.h — what the host classes share to reach the HIP library: the error type, and the
// flattening of a SceneManager into the plain arrays of include/rt_amd.h.
#ifndef RTAMD_BACKEND_H
#define RTAMD_BACKEND_H

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace rtamd {

class SceneManager;

// Thrown when a scene holds something the GPU path does not know (a user-defined Primitives
// subclass, a primitive without material), an argument is invalid, or a HIP call fails; status()
// is the rt_status of include/rt_amd.h.  The reference itself only asserts (src/Camera.cpp:76).
class BackendError : public std::runtime_error {
public:
    BackendError(int status, const std::string &what) : std::runtime_error(what), status_(status) {}
    int status() const { return status_; }

private:
    int status_;
};

// A SceneManager as the C ABI takes it.  prims[i] is sceneManager.primitives()[i]: insertion order
// is the object id and the tie-break order of hitClosest (reference src/RayTracer.cpp:77-82).
struct FlatScene {
    std::vector<rt_prim> prims;
    std::vector<float> verts; // xyz triples of the polygons
    std::vector<rt_material> mats; // de-duplicated by RTMaterial pointer
    std::vector<rt_light> lights;
};

FlatScene flattenScene(const SceneManager &sceneManager, bool withLights);

// Byte-wise equality of two flattened scenes: what decides whether the device copy is still current.
bool sameFlatScene(const FlatScene &a, const FlatScene &b);

// rt_upload_scene_ex + rt_upload_lights; throws BackendError on failure.
void uploadScene(rt_ctx *ctx, const FlatScene &flat);

} // namespace rtamd
#endif
