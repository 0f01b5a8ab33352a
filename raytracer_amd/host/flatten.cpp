// Flattening of the host-side scene graph into the arrays of the C ABI (include/rt_amd.h).
#include <cstring>
#include <map>

#include "Backend.h"
#include "RayTracer.h"

namespace rtamd {

namespace {
void put3(float *dst, const gm::IVec3f &v) {
    dst[0] = (float)v.x();
    dst[1] = (float)v.y();
    dst[2] = (float)v.z();
}

rt_material flattenMaterial(const RTMaterial *m) {
    rt_material fm{};
    fm.type = (uint32_t)m->kind();
    gm::IVec3f albedo;
    switch (m->kind()) {
    case RTMaterialKind::Lambertian: albedo = m->diffuse(); break;
    case RTMaterialKind::Metal:
        albedo = m->specular();
        fm.fuzz = (float)static_cast<const RTMetal *>(m)->fuzz();
        break;
    case RTMaterialKind::Dielectric:
        albedo = static_cast<const RTDielectric *>(m)->tint();
        fm.ior = (float)static_cast<const RTDielectric *>(m)->refractionIndex();
        break;
    case RTMaterialKind::Emissive: break;
    }
    put3(fm.albedo, albedo);
    put3(fm.emission, m->emitted());
    put3(fm.diffuse, m->diffuse()); // what Light::getDirectLighting multiplies by (inc/RTObjects.h:251-252)
    return fm;
}
} // namespace

FlatScene flattenScene(const SceneManager &sceneManager, bool withLights) {
    FlatScene flat;
    std::map<const RTMaterial *, uint32_t> seen;
    for (const Primitives *p : sceneManager.primitives()) {
        const RTMaterial *m = p->material();
        if (!m) throw BackendError(RT_ERR_INVALID, "primitive '" + p->typeString() + "' has no material");
        auto it = seen.find(m);
        if (it == seen.end()) {
            it = seen.emplace(m, (uint32_t)flat.mats.size()).first;
            flat.mats.push_back(flattenMaterial(m));
        }
        rt_prim pr{};
        pr.mat = it->second;
        pr.selected = p->selected() ? 1u : 0u;
        if (const auto *s = dynamic_cast<const SphereObject *>(p)) {
            pr.type = RT_PRIM_SPHERE;
            put3(pr.p, s->position());
            pr.p[3] = s->getRadius();
        } else if (const auto *pl = dynamic_cast<const PlaneObject *>(p)) {
            pr.type = RT_PRIM_PLANE;
            put3(pr.p, pl->position());
            put3(pr.p + 3, pl->getNormal());
        } else if (const auto *c = dynamic_cast<const CubeObject *>(p)) {
            pr.type = RT_PRIM_CUBE;
            put3(pr.p, c->position());
            put3(pr.p + 3, c->getHalfSize());
        } else if (const auto *g = dynamic_cast<const PolygonObject *>(p)) {
            // the polygon lives where its vertices are: position() is the centroid, and a position
            // handed to SceneManager::addObject does not move it (reference src/RayTracer.cpp:26)
            pr.type = RT_PRIM_POLYGON;
            put3(pr.p, g->normal());
            put3(pr.p + 3, g->centroid());
            pr.first_vertex = (uint32_t)(flat.verts.size() / 3);
            pr.n_vertices = (uint32_t)g->vertices().size();
            for (const gm::IPoint3 &v : g->vertices()) {
                flat.verts.push_back((float)v.x());
                flat.verts.push_back((float)v.y());
                flat.verts.push_back((float)v.z());
            }
        } else {
            throw BackendError(RT_ERR_UNSUPPORTED,
                               "primitive '" + p->typeString() + "' is not a kind the GPU path intersects");
        }
        flat.prims.push_back(pr);
    }
    if (withLights) {
        for (const Light *l : sceneManager.lights()) {
            rt_light fl{};
            put3(fl.position, l->position());
            put3(fl.ambient, l->ambientIntensity());
            put3(fl.diffuse, l->defuseIntensity());
            put3(fl.specular, l->specularIntensity());
            fl.view_pow = (float)l->viewLightPow();
            flat.lights.push_back(fl);
        }
    }
    return flat;
}

namespace {
template <typename T>
bool sameBytes(const std::vector<T> &a, const std::vector<T> &b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}
} // namespace

bool sameFlatScene(const FlatScene &a, const FlatScene &b) {
    return sameBytes(a.prims, b.prims) && sameBytes(a.verts, b.verts) && sameBytes(a.mats, b.mats) &&
           sameBytes(a.lights, b.lights);
}

void uploadScene(rt_ctx *ctx, const FlatScene &flat) {
    int rc = rt_upload_scene_ex(ctx, (int)flat.prims.size(), flat.prims.data(), (int)(flat.verts.size() / 3),
                                flat.verts.data(), (int)flat.mats.size(), flat.mats.data());
    if (rc != RT_OK) throw BackendError(rc, std::string("rt_upload_scene_ex: ") + rt_last_error(ctx));
    rc = rt_upload_lights(ctx, (int)flat.lights.size(), flat.lights.data());
    if (rc != RT_OK) throw BackendError(rc, std::string("rt_upload_lights: ") + rt_last_error(ctx));
}

} // namespace rtamd
