// rt_prims.h — the primitives beside the sphere (plane, axis-aligned box, planar polygon), the
// selection halo, and the closest-hit merge over a general primitive list.  Used by the
// "full" variants of the kernels; sphere-only scenes never touch this code.
#pragma once
#include "rt_device.h"

namespace rt {

enum { PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_CUBE = 2, PRIM_POLYGON = 3 };

// HitRecord::object codes: a primitive index, or one of these
constexpr int OBJ_NONE = -1; // no hit
constexpr int OBJ_NULL = -2; // a hit whose object pointer is null (cube, polygon: inc/RTObjects.h:507,642)

constexpr float EXPAND_COEF = 1.05f; // Primitives::EXPAND_COEF, inc/RTObjects.h:26

// Device copy of one primitive, 64 bytes; the array index is the object id (insertion order).
//   sphere : p = centre xyz, radius          plane  : p = point xyz, normal xyz
//   cube   : p = centre xyz, half size xyz   polygon: p = normal xyz, centroid xyz + vertex range
struct DPrim {
    uint32_t type, mat, first_vertex, n_vertices;
    float p[8];
    uint32_t selected;
    int32_t sph_local; // spheres: index in the packed scan arrays, else -1
    uint32_t _pad[2];
};

static_assert(sizeof(DPrim) == 64, "load_prim reads a primitive as four 16-byte words");

// One record of a constant primitive list (64 bytes: one s_load_dwordx16 when the index is uniform).
__device__ __forceinline__ DPrim load_prim(cptr<DPrim> list, int idx) {
    cptr<cuint4> w = (cptr<cuint4>)(list + idx);
    const cuint4 a = w[0], b = w[1], c = w[2], d = w[3];
    DPrim r;
    r.type = a.x; r.mat = a.y; r.first_vertex = a.z; r.n_vertices = a.w;
    r.p[0] = __uint_as_float(b.x); r.p[1] = __uint_as_float(b.y); r.p[2] = __uint_as_float(b.z); r.p[3] = __uint_as_float(b.w);
    r.p[4] = __uint_as_float(c.x); r.p[5] = __uint_as_float(c.y); r.p[6] = __uint_as_float(c.z); r.p[7] = __uint_as_float(c.w);
    r.selected = d.x; r.sph_local = (int32_t)d.y; r._pad[0] = d.z; r._pad[1] = d.w;
    return r;
}

struct FullScene {
    // read through the constant address space: loops over these lists run the same index in every lane
    // and get scalar loads
    cptr<DPrim> prims;        // every primitive, in insertion order
    cptr<float> verts;        // polygon vertices, xyz
    cptr<int32_t> others;     // indices of the non-sphere primitives, ascending
    cptr<int32_t> halo;       // indices taking part in the halo pass (selected ones and planes), ascending
    const float4 *halo_bound; // per entry of `halo`: a sphere (centre, radius) around the primitive grown by EXPAND_COEF;
                              // radius -1 marks a plane, which has none
    cptr<int32_t> sph_gid;    // packed sphere index -> primitive index
    int n_others, n_halo;
};

// The closest hit so far, in the reference's units of t.
struct Closest {
    float t;
    int prim;       // primitive index, -1 = none
    float scale;    // 1, or EXPAND_COEF when the hit is on a halo shell
    int box_face;   // cubes: 0..5 = -x,+x,-y,+y,-z,+z of the slab that produced t
    bool expanded;  // HitRecord::hitExpanded
};

// Quotients below are products with v_rcp_f32 (1 ulp): a correctly rounded single-precision quotient is an
// eleven-instruction sequence, and these tests run for every ray and primitive.

// "first primitive in insertion order reaching the smallest t wins" (strict < in
// SceneManager::hitClosest, src/RayTracer.cpp:77-82) as an order-free rule
__device__ __forceinline__ bool beats(float t, int prim, const Closest &c) {
    return t < c.t || (t == c.t && prim < c.prim);
}

// SphereObject::hitDetail's root choice (inc/RTObjects.h:135-149) for one sphere, plain arithmetic.
__device__ __forceinline__ bool sphere_t(V3 o, V3 d, float tmin, float tmax, V3 c, float r, float &t) {
    V3 oc = o - c;
    float a = dot(d, d), hb = dot(oc, d);
    float cc = fmaf(oc.x, oc.x, fmaf(oc.y, oc.y, fmaf(oc.z, oc.z, -(r * r))));
    float disc = fmaf(hb, hb, -(a * cc));
    if (disc < 0.0f) return false;
    float sq = __builtin_amdgcn_sqrtf(disc);
    float q = -(hb + __builtin_copysignf(sq, hb));
    float r0 = q * __builtin_amdgcn_rcpf(a), r1 = cc * __builtin_amdgcn_rcpf(q);
    float t_near = fminf(r0, r1), t_far = fmaxf(r0, r1);
    float cand = (t_near > tmin && t_near < tmax) ? t_near : t_far;
    if (!(cand > tmin && cand < tmax)) return false;
    t = cand;
    return true;
}

// PlaneObject::hit, inc/RTObjects.h:174-177
__device__ __forceinline__ bool plane_t(V3 o, V3 d, float tmin, float tmax, const DPrim &pr, float &t) {
    V3 pos = mk(pr.p[0], pr.p[1], pr.p[2]), n = mk(pr.p[3], pr.p[4], pr.p[5]);
    float time = dot(n, pos - o) * __builtin_amdgcn_rcpf(dot(n, d));
    if (!(time > tmin && time < tmax)) return false;
    t = time;
    return true;
}

// CubeObject::hitBox, inc/RTObjects.h:593-620.  Also reports which slab face produced t: in double
// precision the reference finds that face by testing the hit point against the six faces within
// 1e-6 (:626-631); in single precision the point itself is not that accurate, the slab is.
__device__ __forceinline__ bool box_t(V3 o, V3 d, float tmin_i, float tmax_i, V3 center, V3 half, float &t, int &face) {
    float tmin = -__builtin_inff(), tmax = __builtin_inff();
    int fmin = 0, fmax = 0;
    const float orig[3] = {o.x, o.y, o.z}, dir[3] = {d.x, d.y, d.z};
    const float cen[3] = {center.x, center.y, center.z}, hs[3] = {half.x, half.y, half.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float lo = cen[i] - hs[i], hi = cen[i] + hs[i];
        if (fabsf(dir[i]) < 1e-12f) {
            if (orig[i] < lo || orig[i] > hi) return false;
        } else {
            float inv = __builtin_amdgcn_rcpf(dir[i]);
            float t0 = (lo - orig[i]) * inv, t1 = (hi - orig[i]) * inv;
            int f0 = 2 * i, f1 = 2 * i + 1;
            if (t0 > t1) {
                float tt = t0; t0 = t1; t1 = tt;
                f0 = 2 * i + 1; f1 = 2 * i;
            }
            if (t0 > tmin) { tmin = t0; fmin = f0; }
            if (t1 < tmax) { tmax = t1; fmax = f1; }
            if (tmax < tmin) return false;
        }
    }
    if (tmin > tmin_i && tmin < tmax_i) { t = tmin; face = fmin; return true; }
    if (tmax > tmin_i && tmax < tmax_i) { t = tmax; face = fmax; return true; }
    return false;
}

// PolygonObject::projectTo2D, inc/RTObjects.h:444-453
__device__ __forceinline__ void project2d(V3 n, V3 p, float &u, float &v) {
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax > ay && ax > az) { u = p.y; v = p.z; }
    else if (ay > az) { u = p.x; v = p.z; }
    else { u = p.x; v = p.y; }
}

// PolygonObject::hitDetail + pointInPolygon2D, inc/RTObjects.h:455-497; `scale` pushes the vertices
// away from the centroid for the halo (:375-381).
__device__ __forceinline__ bool polygon_t(V3 o, V3 d, float tmin, float tmax, const DPrim &pr, cptr<float> verts,
                                          float scale, float &t) {
    const int nv = (int)pr.n_vertices;
    if (nv < 3) return false;
    V3 n = mk(pr.p[0], pr.p[1], pr.p[2]), cen = mk(pr.p[3], pr.p[4], pr.p[5]);
    float denom = dot(n, d);
    if (fabsf(denom) < 1e-12f) return false;
    float time = dot(n, cen - o) * __builtin_amdgcn_rcpf(denom);
    if (!(time > tmin && time < tmax)) return false;
    V3 p = madd(o, d, time);
    float px, py;
    project2d(n, p, px, py);
    bool inside = false;
    cptr<float> vv = verts + 3 * pr.first_vertex;
    V3 vj = mk(vv[3 * (nv - 1)], vv[3 * (nv - 1) + 1], vv[3 * (nv - 1) + 2]);
    vj = madd(cen, vj - cen, scale);
    for (int i = 0; i < nv; ++i) {
        V3 vi = mk(vv[3 * i], vv[3 * i + 1], vv[3 * i + 2]);
        vi = madd(cen, vi - cen, scale);
        float xi, yi, xj, yj;
        project2d(n, vi, xi, yi);
        project2d(n, vj, xj, yj);
        bool cross = ((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) * __builtin_amdgcn_rcpf(yj - yi + 1e-20f) + xi);
        inside = inside != cross;
        vj = vi;
    }
    if (!inside) return false;
    t = time;
    return true;
}

// Primitives::hit of one non-sphere primitive against (tmin, c.t]; updates c on success.
__device__ __forceinline__ void other_hit(const FullScene &fs, int idx, V3 o, V3 d, float tmin, Closest &c) {
    const DPrim pr = load_prim(fs.prims, idx);
    float t = 0.f;
    int face = 0;
    bool ok;
    // the interval's upper end is inclusive here and the tie is settled by beats()
    const float tmax = __builtin_inff();
    if (pr.type == PRIM_PLANE) ok = plane_t(o, d, tmin, tmax, pr, t);
    else if (pr.type == PRIM_CUBE)
        ok = box_t(o, d, tmin, tmax, mk(pr.p[0], pr.p[1], pr.p[2]), mk(pr.p[3], pr.p[4], pr.p[5]), t, face);
    else ok = polygon_t(o, d, tmin, tmax, pr, fs.verts, 1.0f, t);
    if (ok && beats(t, idx, c)) {
        c.t = t;
        c.prim = idx;
        c.scale = 1.0f;
        c.box_face = face;
        c.expanded = false;
    }
}

// The selection-halo pass of SceneManager::hitClosest (src/RayTracer.cpp:84-97), run for primary rays:
// every selected primitive (and every plane, whose hitExpanded is its plain hit, inc/RTObjects.h:192-194)
// is tried in insertion order against the shrinking interval, first as itself and, failing that, grown by
// EXPAND_COEF (sphere :96-105, polygon :370-386, cube :560-568).
// In the reference the record persists across primitives, so a plain hit found here keeps an earlier
// halo flag; but such a plain hit never reaches the caller there: the normal pass found the same
// primitive at the same t, and the tie goes to the normal record (src/RayTracer.cpp:95-99).  In single
// precision the two passes compute that t by different formulas and the tie can break either way, so
// here the flag is that of the final record alone and the caller takes this record only when it lies on
// a grown shell.
// `in_view`: bit k clear leaves out entry k of the list (k < 64); the caller knows that the ray misses it, plain and grown.
__device__ __forceinline__ Closest halo_pass(const FullScene &fs, V3 o, V3 d, float tmin, unsigned long long in_view = ~0ull) {
    Closest e;
    e.t = __builtin_inff();
    e.prim = -1;
    e.scale = 1.0f;
    e.box_face = 0;
    e.expanded = false;
    for (int k = 0; k < fs.n_halo; ++k) {
        if (k < 64 && !((in_view >> k) & 1ull)) continue;
        const int idx = fs.halo[k];
        const DPrim pr = load_prim(fs.prims, idx);
        float t = 0.f;
        int face = 0;
        bool ok = false, grown = false;
        if (pr.type == PRIM_PLANE) {
            ok = plane_t(o, d, tmin, e.t, pr, t);
        } else if (pr.type == PRIM_SPHERE) {
            V3 c = mk(pr.p[0], pr.p[1], pr.p[2]);
            ok = sphere_t(o, d, tmin, e.t, c, pr.p[3], t);
            if (!ok) { ok = sphere_t(o, d, tmin, e.t, c, pr.p[3] * EXPAND_COEF, t); grown = ok; }
        } else if (pr.type == PRIM_CUBE) {
            V3 c = mk(pr.p[0], pr.p[1], pr.p[2]), hs = mk(pr.p[3], pr.p[4], pr.p[5]);
            ok = box_t(o, d, tmin, e.t, c, hs, t, face);
            if (!ok) { ok = box_t(o, d, tmin, e.t, c, hs * EXPAND_COEF, t, face); grown = ok; }
        } else {
            ok = polygon_t(o, d, tmin, e.t, pr, fs.verts, 1.0f, t);
            if (!ok) { ok = polygon_t(o, d, tmin, e.t, pr, fs.verts, EXPAND_COEF, t); grown = ok; }
        }
        if (ok) {
            e.t = t;
            e.prim = idx;
            e.scale = grown ? EXPAND_COEF : 1.0f;
            e.box_face = face;
            e.expanded = grown;
        }
    }
    return e;
}

// Hit point, normal, face flag and HitRecord::object code of the winning primitive
// (sphere inc/RTObjects.h:151-157, plane :179-187, cube :622-642, polygon :499-507).
__device__ __forceinline__ Hit finish_any(const FullScene &fs, V3 o, V3 d, const Closest &c, int &object) {
    const DPrim pr = load_prim(fs.prims, c.prim);
    Hit h;
    if (pr.type == PRIM_SPHERE) {
        object = c.prim;
        return finish_hit(o, d, c.t, make_float4(pr.p[0], pr.p[1], pr.p[2], pr.p[3] * c.scale));
    }
    h.p = madd(o, d, c.t);
    V3 outward;
    if (pr.type == PRIM_PLANE) {
        V3 n = mk(pr.p[3], pr.p[4], pr.p[5]);
        outward = dot(n, d) < 0.0f ? n * -1.0f : n; // turned along the ray: a plane is never front-facing
        object = c.prim;
    } else if (pr.type == PRIM_CUBE) {
        float s = (c.box_face & 1) ? 1.0f : -1.0f;
        int axis = c.box_face >> 1;
        outward = mk(axis == 0 ? s : 0.f, axis == 1 ? s : 0.f, axis == 2 ? s : 0.f);
        object = OBJ_NULL;
    } else {
        V3 n = mk(pr.p[0], pr.p[1], pr.p[2]);
        outward = dot(n, d) > 0.0f ? n * -1.0f : n;
        object = OBJ_NULL;
    }
    h.front = dot(d, outward) < 0.0f;
    V3 nn = normalized(outward);
    h.n = h.front ? nn : nn * -1.0f;
    return h;
}

} // namespace rt
