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
    uint32_t first_edge; // polygons: first of its n_vertices records in FullScene::edges
    uint32_t _pad;
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
    r.selected = d.x; r.sph_local = (int32_t)d.y; r.first_edge = d.z; r._pad = d.w;
    return r;
}

struct FullScene {
    // read through the constant address space: loops over these lists run the same index in every lane
    // and get scalar loads
    cptr<DPrim> prims;        // every primitive, in insertion order
    cptr<float> verts;        // polygon vertices, xyz
    cptr<float4> edges;       // per polygon vertex i, as the polygon test reads it: (x_i, y_i, x_j - x_i, y_j - y_i + 1e-20) in the
                              // polygon's 2D projection, j the vertex before i; n_edges records for the polygons as they are
                              // (DPrim::first_edge), then n_edges for the polygons grown by EXPAND_COEF; built on the host
    int n_edges;
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

// CubeObject::hitBox, inc/RTObjects.h:593-620: the slab test, entry and exit times (tn, tf) of the ray's line through the
// box and the faces (0..5 = -x,+x,-y,+y,-z,+z) that produce them; false when the line misses the box.  `scale` grows the
// half size (1, or EXPAND_COEF for the halo, :560-568); the slab bounds centre -+ half * scale are one fused
// multiply-add each, spelled out so that every copy of this arithmetic rounds alike.
__device__ __forceinline__ bool box_slabs(V3 o, V3 d, V3 center, V3 half, float scale, float &tn, float &tf, int &fn, int &ff) {
    float tmin = -__builtin_inff(), tmax = __builtin_inff();
    int fmin = 0, fmax = 0;
    const float orig[3] = {o.x, o.y, o.z}, dir[3] = {d.x, d.y, d.z};
    const float cen[3] = {center.x, center.y, center.z}, hs[3] = {half.x, half.y, half.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float lo = fmaf(-hs[i], scale, cen[i]), hi = fmaf(hs[i], scale, cen[i]);
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
    tn = tmin; tf = tmax; fn = fmin; ff = fmax;
    return true;
}

// The hit of the slab test inside (tmin_i, tmax_i): the entry time if it lies there, else the exit time.
// The closest-hit loops only need t, and for all but a wave that holds a ray parallel to an axis plane the slab
// test is minima and maxima: the entry time is the largest of the three nearer slab times, the exit time the
// smallest of the farther ones, and "exit before entry" at any axis shows in the final pair as well (entry only
// grows, exit only shrinks).  Same operations on the same operands as box_slabs(), so the same t; the face is
// looked up by box_face() for the one box that wins.
__device__ __forceinline__ bool box_t(V3 o, V3 d, float tmin_i, float tmax_i, V3 center, V3 half, float scale, float &t) {
    float tn, tf;
    if (__ballot(fabsf(d.x) < 1e-12f || fabsf(d.y) < 1e-12f || fabsf(d.z) < 1e-12f)) {
        int fn, ff;
        if (!box_slabs(o, d, center, half, scale, tn, tf, fn, ff)) return false;
    } else {
        const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
        const float x0 = (fmaf(-half.x, scale, center.x) - o.x) * ix, x1 = (fmaf(half.x, scale, center.x) - o.x) * ix;
        const float y0 = (fmaf(-half.y, scale, center.y) - o.y) * iy, y1 = (fmaf(half.y, scale, center.y) - o.y) * iy;
        const float z0 = (fmaf(-half.z, scale, center.z) - o.z) * iz, z1 = (fmaf(half.z, scale, center.z) - o.z) * iz;
        tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        if (tf < tn) return false;
    }
    if (tn > tmin_i && tn < tmax_i) { t = tn; return true; }
    if (tf > tmin_i && tf < tmax_i) { t = tf; return true; }
    return false;
}

// The face a box was hit on at time t (a time box_t() returned for this ray and box): in double precision the
// reference finds it by testing the hit point against the six faces within 1e-6 (inc/RTObjects.h:626-631); in
// single precision the point itself is not that accurate, the slab that produced t is.
__device__ __forceinline__ int box_face(V3 o, V3 d, V3 center, V3 half, float scale, float t) {
    float tn = 0.f, tf = 0.f;
    int fn = 0, ff = 0;
    (void)box_slabs(o, d, center, half, scale, tn, tf, fn, ff);
    return t == tn ? fn : ff;
}

// PolygonObject::projectTo2D, inc/RTObjects.h:444-453
__host__ __device__ __forceinline__ void project2d(V3 n, V3 p, float &u, float &v) {
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax > ay && ax > az) { u = p.y; v = p.z; }
    else if (ay > az) { u = p.x; v = p.z; }
    else { u = p.x; v = p.y; }
}

// One record of FullScene::edges: vertex i of a polygon pushed away from the centroid by `scale` (1, or EXPAND_COEF for
// the halo, inc/RTObjects.h:375-381) and projected (projectTo2D), with its difference to the vertex before it.  The
// operations and their order are those of pointInPolygon2D's crossing test (inc/RTObjects.h:455-497).
__host__ __device__ __forceinline__ void polygon_edge(V3 n, V3 cen, V3 vi, V3 vj, float scale, float out[4]) {
    vi = madd(cen, vi - cen, scale);
    vj = madd(cen, vj - cen, scale);
    float xi, yi, xj, yj;
    project2d(n, vi, xi, yi);
    project2d(n, vj, xj, yj);
    out[0] = xi;
    out[1] = yi;
    out[2] = xj - xi;
    const float dy = yj - yi;
    out[3] = dy + 1e-20f;
}

// PolygonObject::hitDetail + pointInPolygon2D, inc/RTObjects.h:455-497, over the edge records of the polygon as it is
// or grown (`grown`).
__device__ __forceinline__ bool polygon_t(V3 o, V3 d, float tmin, float tmax, const DPrim &pr, const FullScene &fs,
                                          bool grown, float &t) {
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
    cptr<float4> ee = fs.edges + ((grown ? fs.n_edges : 0) + (int)pr.first_edge);
    bool above_j = load_const4(ee, nv - 1).y > py; // the vertex before the first is the last
    for (int i = 0; i < nv; ++i) {
        const float4 e = load_const4(ee, i);
        const bool above_i = e.y > py;
        const bool cross = (above_i != above_j) && (px < fmaf(e.z * (py - e.y), __builtin_amdgcn_rcpf(e.w), e.x));
        inside = inside != cross;
        above_j = above_i;
    }
    if (!inside) return false;
    t = time;
    return true;
}

// Primitives::hit of one non-sphere primitive against (tmin, c.t]; updates c on success.
__device__ __forceinline__ void other_hit(const FullScene &fs, int idx, V3 o, V3 d, float tmin, Closest &c) {
    const DPrim pr = load_prim(fs.prims, idx);
    float t = 0.f;
    bool ok;
    // the interval's upper end is inclusive here and the tie is settled by beats()
    const float tmax = __builtin_inff();
    if (pr.type == PRIM_PLANE) ok = plane_t(o, d, tmin, tmax, pr, t);
    else if (pr.type == PRIM_CUBE) ok = box_t(o, d, tmin, tmax, mk(pr.p[0], pr.p[1], pr.p[2]), mk(pr.p[3], pr.p[4], pr.p[5]), 1.0f, t);
    else ok = polygon_t(o, d, tmin, tmax, pr, fs, false, t);
    if (ok && beats(t, idx, c)) {
        c.t = t;
        c.prim = idx;
        c.scale = 1.0f;
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
    e.expanded = false;
    for (int k = 0; k < fs.n_halo; ++k) {
        if (k < 64 && !((in_view >> k) & 1ull)) continue;
        const int idx = fs.halo[k];
        const DPrim pr = load_prim(fs.prims, idx);
        float t = 0.f;
        bool ok = false, grown = false;
        if (pr.type == PRIM_PLANE) {
            ok = plane_t(o, d, tmin, e.t, pr, t);
        } else if (pr.type == PRIM_SPHERE) {
            V3 c = mk(pr.p[0], pr.p[1], pr.p[2]);
            ok = sphere_t(o, d, tmin, e.t, c, pr.p[3], t);
            if (!ok) { ok = sphere_t(o, d, tmin, e.t, c, pr.p[3] * EXPAND_COEF, t); grown = ok; }
        } else if (pr.type == PRIM_CUBE) {
            V3 c = mk(pr.p[0], pr.p[1], pr.p[2]), hs = mk(pr.p[3], pr.p[4], pr.p[5]);
            ok = box_t(o, d, tmin, e.t, c, hs, 1.0f, t);
            if (!ok) { ok = box_t(o, d, tmin, e.t, c, hs, EXPAND_COEF, t); grown = ok; }
        } else {
            ok = polygon_t(o, d, tmin, e.t, pr, fs, false, t);
            if (!ok) { ok = polygon_t(o, d, tmin, e.t, pr, fs, true, t); grown = ok; }
        }
        if (ok) {
            e.t = t;
            e.prim = idx;
            e.scale = grown ? EXPAND_COEF : 1.0f;
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
        const int face = box_face(o, d, mk(pr.p[0], pr.p[1], pr.p[2]), mk(pr.p[3], pr.p[4], pr.p[5]), c.scale, c.t);
        float s = (face & 1) ? 1.0f : -1.0f;
        int axis = face >> 1;
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
