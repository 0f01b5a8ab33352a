/*
 * rt_oracle.c — CPU restatement of the I1Va/RayTracer per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing the product ships may call into this file:
 * it is the checker used by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED.  The reference cannot be built in this environment: it
 * includes Geom.hpp / IVec3f.hpp / GmUtilities.hpp and links a target "GeomLib"
 * (reference CMakeLists.txt:20, inc/RTObjects.h:13-14, inc/RTGeometry.h:4,7)
 * that is not part of the reference tree, carries no version pin, and is not
 * installed here.  The reference also ships no tests, golden images or
 * known-answer vectors.  This file therefore follows the reference's *source
 * text* line by line (citations on every function) and DEFINES the pieces that
 * live in the missing library:
 *   - vec3 arithmetic: plain component-wise IEEE arithmetic in `real`
 *     (double by default), dot = x*x' + y*y' + z*z' summed left to right,
 *     normalized() = v / sqrt(v.v), nearZero() = all |components| < 1e-8;
 *   - the random generator: a counter-based hash (pcg4d, Jarzynski & Olano,
 *     "Hash Functions for GPU Rendering", JCGT 2020) keyed on
 *     (pixel id, sample index, call index, seed); each call yields four 32-bit
 *     words, a word w maps to the uniform (w >> 8) * 2^-24 in [0,1);
 *   - IVec3f::randomUnit(): the analytic map z = 1-2u0, phi = 2*pi*u1.
 * What pins it instead: the hand-derived known-answer vectors in
 * tests/golden/ (closed-form ray/sphere, reflection, refraction, tone-map
 * cases) and the internal recursive-vs-iterative cross-check.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; -DRT_REAL=float gives
 * the single-precision diagnostic variant).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef RT_REAL
#define RT_REAL double
#endif
typedef RT_REAL real;

#define ORC_API __attribute__((visibility("default")))

/* ---- flattened inputs: same layout as include/rt_amd.h -------------------- */
typedef struct { float cx, cy, cz, r; } orc_sphere;
typedef struct {
    uint32_t type;
    float albedo[3];
    float fuzz;
    float ior;
    float emission[3];
    float diffuse[3];
} orc_material;
typedef struct {
    float position[3], ambient[3], diffuse[3], specular[3];
    float view_pow;
    float _pad[3];
} orc_light;
typedef struct { float center[3], upper_left[3], right[3], down[3]; } orc_camera;

/* One primitive of a general scene (same layout as rt_prim in include/rt_amd.h):
 *   sphere : p = centre xyz, radius                    (SphereObject, inc/RTObjects.h:85-160)
 *   plane  : p = point xyz, normal xyz                 (PlaneObject, :162-218)
 *   cube   : p = centre xyz, half size xyz             (CubeObject, :543-645)
 *   polygon: p = normal xyz, centroid xyz; vertices [first_vertex, first_vertex + n_vertices)
 *            of the vertex array, xyz each             (PolygonObject, :323-540) */
enum { PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_CUBE = 2, PRIM_POLYGON = 3 };
typedef struct {
    uint32_t type, mat, first_vertex, n_vertices;
    float p[8];
    uint32_t selected;
    uint32_t _pad[3];
} orc_prim;

enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_EMISSIVE = 3 };

typedef struct { real x, y, z; } vec3;

typedef struct {
    int n;
    const orc_sphere *sph;
    const uint32_t *mat_id;
    int n_mats;
    const orc_material *mats;
    int n_lights;             /* 0 unless direct lighting is enabled */
    const orc_light *lights;
    vec3 cam_center;          /* Camera::center_, read by computeDirectLighting */
    int n_prims;              /* > 0: general scene (the sphere arrays above are then unused) */
    const orc_prim *prims;
    const float *verts;
} scene_t;

#define OBJ_NONE (-1) /* no hit */
#define OBJ_NULL (-2) /* a hit whose HitRecord::object is nullptr (cube, polygon: inc/RTObjects.h:507,642) */

/* HitRecord, reference inc/RTGeometry.h:49-62 (material/object pointers become indices) */
typedef struct {
    vec3 point, normal;
    int object;      /* HitRecord::object: primitive index, OBJ_NONE or OBJ_NULL */
    int prim;        /* primitive index of the hit (material lookup) */
    int front;
    real time;
    int hit_expanded; /* HitRecord::hitExpanded */
} hitrec;

static real r_sqrt(real x) { return (real)sqrt((double)x); }

/* ---- the vec3 operations the reference takes from GeomLib (defined here) --- */
static inline vec3 v3(real x, real y, real z) { vec3 v = {x, y, z}; return v; }
static inline vec3 vadd(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 vsub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 vmuls(vec3 a, real s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 vdivs(vec3 a, real s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline vec3 vmul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline real vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 vcross(vec3 a, vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline real vlen2(vec3 a) { return vdot(a, a); }
static inline vec3 vnormalized(vec3 a) { return vdivs(a, r_sqrt(vlen2(a))); }
static inline int vnear_zero(vec3 a) {
    const real e = (real)1e-8;
    return (real)fabs((double)a.x) < e && (real)fabs((double)a.y) < e && (real)fabs((double)a.z) < e;
}
static inline vec3 vf(const float *p) { return v3((real)p[0], (real)p[1], (real)p[2]); }

/* ---- random numbers (defined here; the reference's live in GeomLib) -------- */
ORC_API void orc_rng4(uint32_t pixel, uint32_t sample, uint32_t call, uint32_t seed, uint32_t out[4]) {
    uint32_t x = pixel, y = sample, z = call, w = seed;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}

static inline real u01(uint32_t w) { return (real)(w >> 8) * (real)(1.0 / 16777216.0); }

/* IVec3f::randomUnit() stand-in: uniform direction from two uniforms. */
static vec3 random_unit(real u0, real u1) {
    real z = (real)1 - (real)2 * u0;
    real r2 = (real)1 - z * z;
    real r = r_sqrt(r2 > 0 ? r2 : 0);
    double phi = 6.283185307179586476925286766559 * (double)u1;
    return v3(r * (real)cos(phi), r * (real)sin(phi), z);
}

/* per-(pixel,sample) random stream: call index advances by one per bounce-level draw */
typedef struct { uint32_t pixel, sample, call, seed; } rng_t;

static void rng_next(rng_t *g, real u[4]) {
    uint32_t w[4];
    orc_rng4(g->pixel, g->sample, g->call, g->seed, w);
    g->call++;
    for (int i = 0; i < 4; i++) u[i] = u01(w[i]);
}

/* ---- geometry helpers, reference inc/RTGeometry.h -------------------------- */

/* reflect(), reference inc/RTGeometry.h:18-20:  v - n*2*dot(v,n) */
static vec3 reflect(vec3 v, vec3 n) { return vsub(v, vmuls(vmuls(n, 2), vdot(v, n))); }

static long g_refract_clamped; /* times the assert at inc/RTGeometry.h:30 would have fired */

/* refract(), reference inc/RTGeometry.h:22-34 (including the 1/eta^2 rescale at :28).
 * Where the reference asserts 1-|perp|^2 > 0 (:30) this restatement clamps at 0 and
 * counts the event; tests require the count to stay 0. */
static vec3 refract(vec3 uv, vec3 n, real eta) {
    vec3 unit = vnormalized(uv);
    real cos_t = (real)fmin((double)vdot(vmuls(unit, -1), n), 1.0);
    vec3 perp = vmuls(vadd(unit, vmuls(n, cos_t)), eta);
    if ((real)1 - vlen2(perp) < 0) perp = vmuls(perp, (real)1 / (eta * eta));
    real k = (real)1 - vlen2(perp);
    if (!(k > 0)) {
#pragma omp atomic
        g_refract_clamped++;
        k = 0;
    }
    vec3 par = vmuls(n, -r_sqrt(k));
    return vadd(perp, par);
}

/* Interval::surrounds, reference inc/RTGeometry.h:79-81 (open interval) */
static inline int surrounds(real lo, real hi, real x) { return lo < x && x < hi; }

/* SphereObject::hitDetail + HitRecord::setFaceNormal,
 * reference inc/RTObjects.h:129-159 and inc/RTGeometry.h:58-61.
 * rec is written only on success. */
static int hit_sphere(vec3 O, vec3 D, real tmin, real tmax, vec3 C, real radius, int id, hitrec *rec) {
    vec3 oc = vsub(O, C);
    real a = vdot(D, D);
    real half_b = vdot(oc, D);
    real c = vdot(oc, oc) - radius * radius;
    real disc = half_b * half_b - a * c;
    if (disc < 0) return 0;
    real sq = r_sqrt(disc);
    real root = (-half_b - sq) / a;
    if (!surrounds(tmin, tmax, root)) {
        root = (-half_b + sq) / a;
        if (!surrounds(tmin, tmax, root)) return 0;
    }
    rec->time = root;
    rec->point = vadd(O, vmuls(D, root));
    vec3 outward = vdivs(vsub(rec->point, C), radius);
    rec->front = vdot(D, outward) < 0;
    vec3 nn = vnormalized(outward);
    rec->normal = rec->front ? nn : vmuls(nn, -1);
    rec->object = id;
    rec->prim = id;
    return 1;
}

/* HitRecord::setFaceNormal, reference inc/RTGeometry.h:58-61 */
static void set_face_normal(hitrec *rec, vec3 D, vec3 outward) {
    rec->front = vdot(D, outward) < 0;
    vec3 nn = vnormalized(outward);
    rec->normal = rec->front ? nn : vmuls(nn, -1);
}

/* PlaneObject::hit, reference inc/RTObjects.h:174-190.  The "outward" normal is turned along the ray
 * before setFaceNormal, so a plane always reports frontFace == false. */
static int hit_plane(vec3 O, vec3 D, real tmin, real tmax, const orc_prim *pr, int id, hitrec *rec) {
    vec3 pos = vf(pr->p), n = vf(pr->p + 3);
    real time = vdot(n, vsub(pos, O)) * ((real)1.0 / vdot(n, D));
    if (!surrounds(tmin, tmax, time)) return 0;
    rec->object = id;
    rec->prim = id;
    rec->time = time;
    rec->point = vadd(O, vmuls(D, time));
    vec3 outward = n;
    if (vdot(n, D) < 0) outward = vmuls(n, -1);
    set_face_normal(rec, D, outward);
    return 1;
}

/* CubeObject::hitBox, reference inc/RTObjects.h:586-644 (axis-aligned slab test; the normal is picked
 * by which face the point lies within 1e-6 of, else by the dominant local axis; object stays null). */
static int hit_box(vec3 O, vec3 D, real tmin_i, real tmax_i, vec3 center, vec3 half, int id, hitrec *rec) {
    const real EPS = (real)1e-6;
    real tmin = (real)-INFINITY, tmax = (real)INFINITY;
    real orig[3] = {O.x, O.y, O.z}, dir[3] = {D.x, D.y, D.z};
    real minB[3] = {center.x - half.x, center.y - half.y, center.z - half.z};
    real maxB[3] = {center.x + half.x, center.y + half.y, center.z + half.z};
    for (int i = 0; i < 3; ++i) {
        if ((real)fabs((double)dir[i]) < (real)1e-12) {
            if (orig[i] < minB[i] || orig[i] > maxB[i]) return 0;
        } else {
            real invD = (real)1.0 / dir[i];
            real t0 = (minB[i] - orig[i]) * invD, t1 = (maxB[i] - orig[i]) * invD;
            if (t0 > t1) { real tt = t0; t0 = t1; t1 = tt; }
            tmin = tmin > t0 ? tmin : t0;
            tmax = tmax < t1 ? tmax : t1;
            if (tmax < tmin) return 0;
        }
    }
    real t = tmin;
    if (!surrounds(tmin_i, tmax_i, t)) {
        t = tmax;
        if (!surrounds(tmin_i, tmax_i, t)) return 0;
    }
    rec->time = t;
    rec->point = vadd(O, vmuls(D, t));
    vec3 P = rec->point, out = v3(0, 0, 0);
    if ((real)fabs((double)(P.x - minB[0])) < EPS) out = v3(-1, 0, 0);
    else if ((real)fabs((double)(P.x - maxB[0])) < EPS) out = v3(1, 0, 0);
    else if ((real)fabs((double)(P.y - minB[1])) < EPS) out = v3(0, -1, 0);
    else if ((real)fabs((double)(P.y - maxB[1])) < EPS) out = v3(0, 1, 0);
    else if ((real)fabs((double)(P.z - minB[2])) < EPS) out = v3(0, 0, -1);
    else if ((real)fabs((double)(P.z - maxB[2])) < EPS) out = v3(0, 0, 1);
    else {
        vec3 l = vsub(P, center);
        real ax = (real)fabs((double)l.x), ay = (real)fabs((double)l.y), az = (real)fabs((double)l.z);
        if (ax > ay && ax > az) out = v3(l.x > 0 ? 1 : -1, 0, 0);
        else if (ay > az) out = v3(0, l.y > 0 ? 1 : -1, 0);
        else out = v3(0, 0, l.z > 0 ? 1 : -1);
    }
    set_face_normal(rec, D, out);
    rec->object = OBJ_NULL;
    rec->prim = id;
    return 1;
}

/* PolygonObject::projectTo2D, reference inc/RTObjects.h:444-453: drop the dominant axis of the normal */
static void project2d(vec3 n, vec3 p, real *u, real *v) {
    real ax = (real)fabs((double)n.x), ay = (real)fabs((double)n.y), az = (real)fabs((double)n.z);
    if (ax > ay && ax > az) { *u = p.y; *v = p.z; }
    else if (ay > az) { *u = p.x; *v = p.z; }
    else { *u = p.x; *v = p.y; }
}

/* PolygonObject::hitDetail + pointInPolygon2D, reference inc/RTObjects.h:455-512: plane through the
 * centroid, even-odd crossing test in the projected plane.  `scale` = 1, or EXPAND_COEF for the halo
 * (vertices pushed away from the centroid, :375-381). */
static int hit_polygon(vec3 O, vec3 D, real tmin, real tmax, const orc_prim *pr, const float *verts, real scale,
                       int id, hitrec *rec) {
    int nv = (int)pr->n_vertices;
    if (nv < 3) return 0;
    vec3 n = vf(pr->p), centroid = vf(pr->p + 3);
    real denom = vdot(n, D);
    if ((real)fabs((double)denom) < (real)1e-12) return 0;
    real t = vdot(n, vsub(centroid, O)) / denom;
    if (!surrounds(tmin, tmax, t)) return 0;
    vec3 P = vadd(O, vmuls(D, t));
    real px, py;
    project2d(n, P, &px, &py);
    int inside = 0;
    const float *vv = verts + 3 * pr->first_vertex;
    for (int i = 0, j = nv - 1; i < nv; j = i++) {
        vec3 vi = vf(vv + 3 * i), vj = vf(vv + 3 * j);
        if (scale != (real)1) {
            vec3 si = vmuls(vsub(vi, centroid), scale), sj = vmuls(vsub(vj, centroid), scale);
            vi = v3(centroid.x + si.x, centroid.y + si.y, centroid.z + si.z);
            vj = v3(centroid.x + sj.x, centroid.y + sj.y, centroid.z + sj.z);
        }
        real xi, yi, xj, yj;
        project2d(n, vi, &xi, &yi);
        project2d(n, vj, &xj, &yj);
        int intersect = ((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi + (real)1e-20) + xi);
        if (intersect) inside = !inside;
    }
    if (!inside) return 0;
    rec->time = t;
    rec->point = P;
    vec3 outward = n;
    if (vdot(n, D) > 0) outward = vmuls(n, -1);
    set_face_normal(rec, D, outward);
    rec->object = OBJ_NULL;
    rec->prim = id;
    return 1;
}

static const real EXPAND_COEF = (real)1.05; /* Primitives::EXPAND_COEF, reference inc/RTObjects.h:26 */

/* Primitives::hit for a general primitive (virtual dispatch of src/RayTracer.cpp:78) */
static int prim_hit(const scene_t *s, int i, vec3 O, vec3 D, real tmin, real tmax, hitrec *rec) {
    const orc_prim *pr = &s->prims[i];
    switch (pr->type) {
    case PRIM_SPHERE: return hit_sphere(O, D, tmin, tmax, vf(pr->p), (real)pr->p[3], i, rec);
    case PRIM_PLANE: return hit_plane(O, D, tmin, tmax, pr, i, rec);
    case PRIM_CUBE: return hit_box(O, D, tmin, tmax, vf(pr->p), vf(pr->p + 3), i, rec);
    default: return hit_polygon(O, D, tmin, tmax, pr, s->verts, (real)1, i, rec);
    }
}

/* Primitives::hitExpanded (selection halo): sphere inc/RTObjects.h:96-105, plane :192-194 (a plane answers
 * with its plain hit whether selected or not), polygon :370-386, cube :560-568. */
static int prim_hit_expanded(const scene_t *s, int i, vec3 O, vec3 D, real tmin, real tmax, hitrec *rec) {
    const orc_prim *pr = &s->prims[i];
    if (pr->type == PRIM_PLANE) return hit_plane(O, D, tmin, tmax, pr, i, rec);
    if (!pr->selected) return 0;
    if (prim_hit(s, i, O, D, tmin, tmax, rec)) return 1;
    int res;
    if (pr->type == PRIM_SPHERE)
        res = hit_sphere(O, D, tmin, tmax, vf(pr->p), (real)pr->p[3] * EXPAND_COEF, i, rec);
    else if (pr->type == PRIM_CUBE)
        res = hit_box(O, D, tmin, tmax, vf(pr->p), vmuls(vf(pr->p + 3), EXPAND_COEF), i, rec);
    else
        res = hit_polygon(O, D, tmin, tmax, pr, s->verts, EXPAND_COEF, i, rec);
    if (res) rec->hit_expanded = 1;
    return res;
}

/* SceneManager::hitClosest, reference src/RayTracer.cpp:68-100.  `expanded_state` is the
 * hitExpandedState argument (true for primary rays only, src/Camera.cpp:141): it adds the
 * selection-halo pass :84-97, whose record persists across objects exactly as in the reference
 * (a plain hit found through hitExpanded does not clear an earlier hitExpanded flag). */
static int hit_closest_ex(const scene_t *s, vec3 O, vec3 D, real tmin, real tmax, int expanded_state, hitrec *rec) {
    hitrec tmp, exp;
    memset(&tmp, 0, sizeof tmp);
    tmp.object = OBJ_NONE;
    tmp.prim = -1;
    exp = tmp;
    real closest = tmax, closest_exp = tmax;
    int any = 0;
    if (s->n_prims > 0) {
        for (int i = 0; i < s->n_prims; i++) {
            if (prim_hit(s, i, O, D, tmin, closest, &tmp)) {
                any = 1;
                closest = tmp.time;
            }
        }
        if (expanded_state) {
            for (int i = 0; i < s->n_prims; i++) {
                if (prim_hit_expanded(s, i, O, D, tmin, closest_exp, &exp)) {
                    any = 1;
                    closest_exp = exp.time;
                }
            }
        }
    } else {
        for (int i = 0; i < s->n; i++) {
            const orc_sphere *sp = &s->sph[i];
            if (hit_sphere(O, D, tmin, closest, v3((real)sp->cx, (real)sp->cy, (real)sp->cz), (real)sp->r, i, &tmp)) {
                any = 1;
                closest = tmp.time;
            }
        }
    }
    *rec = closest_exp < closest ? exp : tmp;
    return any;
}

static int hit_closest(const scene_t *s, vec3 O, vec3 D, real tmin, real tmax, hitrec *rec) {
    return hit_closest_ex(s, O, D, tmin, tmax, 0, rec);
}

/* RTDielectric::reflectance, reference inc/RTMaterial.h:193-197 */
static real reflectance(real cosine, real ri) {
    real r0 = ((real)1 - ri) / ((real)1 + ri);
    r0 = r0 * r0;
    return r0 + ((real)1 - r0) * (real)pow((double)((real)1 - cosine), 5.0);
}

/* RTMaterial::scatter for the four kinds, reference inc/RTMaterial.h:
 *   Lambertian 67-80, Metal 105-115 (+reflect), Dielectric 142-170 (+refract), Emissive 205-207.
 * u[0],u[1] feed randomUnit(), u[2] feeds randomDouble(). */
static int scatter(const orc_material *m, vec3 in_dir, const hitrec *rec, const real u[4],
                   vec3 *att, vec3 *out_org, vec3 *out_dir) {
    switch (m->type) {
    case MAT_LAMBERTIAN: {
        vec3 dir = vadd(rec->normal, random_unit(u[0], u[1]));
        if (vnear_zero(dir)) dir = rec->normal;
        *out_org = rec->point;
        *out_dir = dir;
        *att = vf(m->albedo);
        return 1;
    }
    case MAT_METAL: {
        vec3 refl = reflect(in_dir, rec->normal);
        refl = vadd(vnormalized(refl), vmuls(random_unit(u[0], u[1]), (real)m->fuzz));
        *out_org = rec->point;
        *out_dir = refl;
        *att = vf(m->albedo);
        return 1;
    }
    case MAT_DIELECTRIC: {
        *att = vf(m->albedo);
        real ior = (real)m->ior;
        real eta = rec->front ? ((real)1 / ior) : ior;
        vec3 unit = vnormalized(in_dir);
        real cos_t = (real)fmin((double)vdot(vmuls(unit, -1), rec->normal), 1.0);
        real s2 = (real)1 - cos_t * cos_t;
        real sin_t = r_sqrt(s2 > 0 ? s2 : 0);
        int cannot = eta * sin_t > (real)1;
        vec3 dir;
        /* short-circuit: the uniform is consumed only when refraction is possible (:158);
         * with a counter-based stream "consumed" just means "looked at". */
        if (cannot || reflectance(cos_t, eta) > u[2]) {
            dir = reflect(unit, rec->normal);
        } else {
            dir = vnormalized(refract(unit, rec->normal, eta));
        }
        const real eps = (real)1e-4;
        *out_org = vadd(rec->point, vmuls(rec->normal, rec->front ? eps : -eps));
        *out_dir = vnormalized(dir);
        return 1;
    }
    default: /* MAT_EMISSIVE */
        return 0;
    }
}

static const real TMIN = (real)0.001; /* CLOSEST_HIT_MIN_T, reference src/Camera.cpp:10 */

/* Light::getDirectLighting with defuseLightIntensity (Lambert) and specularLightIntesity
 * (Phong), reference inc/RTObjects.h:247-257, 300-320.  getOrtogonal(a, b) is the part of a
 * orthogonal to b, so toLight - 2*getOrtogonal(toLight, n) is toLight mirrored about n. */
static vec3 light_direct(const orc_light *lt, vec3 to_view, const hitrec *rec, const orc_material *m, int hitted) {
    vec3 to_light = vnormalized(vsub(vf(lt->position), rec->point));
    vec3 mdiff = vf(m->diffuse);
    vec3 ambient = vmul(vf(lt->ambient), mdiff);
    real cos_l = vdot(vnormalized(to_light), vnormalized(rec->normal));
    vec3 diffuse = cos_l > 0 ? vmul(vmuls(vf(lt->diffuse), cos_l), mdiff) : v3(0, 0, 0);
    vec3 nn = rec->normal;
    vec3 ortho = vsub(to_light, vmuls(nn, vdot(to_light, nn) / vdot(nn, nn)));
    vec3 refl = vsub(to_light, vmuls(ortho, 2));
    real cos_v = vdot(vnormalized(refl), vnormalized(vnormalized(to_view)));
    vec3 specular = v3(0, 0, 0);
    if (cos_v > 0) {
        real k = (real)pow((double)cos_v, (double)lt->view_pow);
        specular = v3(k, k, k);
    }
    real shadow = hitted ? (real)0.0 : (real)1.0;
    return vadd(ambient, vmuls(vadd(diffuse, specular), shadow));
}

/* Camera::computeDirectLighting, reference src/Camera.cpp:161-178: one closest-hit query per light
 * along the (unnormalised, unbounded) ray towards it; a hit on the shaded object itself does not count. */
static vec3 direct_lighting(const scene_t *s, const hitrec *rec, const orc_material *m, uint64_t *rays) {
    vec3 sum = v3(0, 0, 0);
    vec3 to_view = vsub(s->cam_center, rec->point);
    for (int i = 0; i < s->n_lights; i++) {
        const orc_light *lt = &s->lights[i];
        hitrec tmp;
        (*rays)++;
        int hitted = hit_closest(s, rec->point, vsub(vf(lt->position), rec->point), (real)0.001, (real)INFINITY, &tmp);
        if (hitted && tmp.object == rec->object) hitted = 0;
        sum = vadd(sum, light_direct(lt, to_view, rec, m, hitted));
    }
    return sum;
}

typedef struct {
    const scene_t *s;
    int samples_per_scatter;
    uint64_t rays;
    int max_depth; /* renderProperties.maxRayDepth: the halo pass runs when depth == max_depth */
} trace_ctx;

static const orc_material *material_of(const scene_t *s, const hitrec *rec) {
    return &s->mats[s->n_prims > 0 ? s->prims[rec->prim].mat : s->mat_id[rec->prim]];
}

/* sky colour on a miss, reference src/Camera.cpp:155-156 (direction used as is) */
static vec3 sky(vec3 d) {
    real a = (real)0.5 * (d.y + (real)1.0);
    return vadd(vmuls(v3(1, 1, 1), (real)1.0 - a), vmuls(v3((real)0.5, (real)0.7, (real)1.0), a));
}

/* Camera::getRayColor + computeMultipleScatterLInderect, reference src/Camera.cpp:137-157,180-195,
 * recursive exactly as written (the :150 term is direct_lighting when lights are enabled). */
static vec3 ray_color(trace_ctx *tc, rng_t *g, vec3 O, vec3 D, int depth) {
    if (depth == 0) return v3(0, 0, 0);
    hitrec rec;
    tc->rays++;
    if (hit_closest_ex(tc->s, O, D, TMIN, (real)INFINITY, depth == tc->max_depth, &rec)) {
        const orc_material *m = material_of(tc->s, &rec);
        vec3 emitted = vf(m->emission);
        if (rec.hit_expanded) emitted = v3(1, 0, 0); /* selection colour, src/Camera.cpp:144-147 */
        vec3 lind = v3(0, 0, 0);
        for (int i = 0; i < tc->samples_per_scatter; i++) {
            real u[4];
            rng_next(g, u);
            vec3 att, so, sd;
            if (scatter(m, D, &rec, u, &att, &so, &sd))
                lind = vadd(lind, vmul(att, ray_color(tc, g, so, sd, depth - 1)));
        }
        lind = vmuls(lind, (real)1.0 / (real)tc->samples_per_scatter);
        vec3 ldirect = tc->s->n_lights ? direct_lighting(tc->s, &rec, m, &tc->rays) : v3(0, 0, 0);
        return vadd(vadd(emitted, lind), ldirect);
    }
    return sky(D);
}

/* The same radiance with samplesPerScatter == 1, unrolled into the loop a GPU runs:
 * L += T*emitted; T *= attenuation; a miss adds T*sky; running out of depth adds nothing. */
static vec3 ray_color_iter(trace_ctx *tc, rng_t *g, vec3 O, vec3 D, int max_depth) {
    vec3 L = v3(0, 0, 0), T = v3(1, 1, 1);
    for (int depth = max_depth; depth > 0; depth--) {
        hitrec rec;
        tc->rays++;
        if (!hit_closest_ex(tc->s, O, D, TMIN, (real)INFINITY, depth == max_depth, &rec)) {
            L = vadd(L, vmul(T, sky(D)));
            break;
        }
        const orc_material *m = material_of(tc->s, &rec);
        L = vadd(L, vmul(T, rec.hit_expanded ? v3(1, 0, 0) : vf(m->emission)));
        if (tc->s->n_lights) L = vadd(L, vmul(T, direct_lighting(tc->s, &rec, m, &tc->rays)));
        real u[4];
        rng_next(g, u);
        vec3 att, so, sd;
        if (!scatter(m, D, &rec, u, &att, &so, &sd)) break;
        T = vmul(T, att);
        O = so;
        D = sd;
    }
    return L;
}

/* Camera::genRay, reference src/Camera.cpp:124-135.  The two randomDouble(0,1) calls
 * are unsequenced in the reference; here the x jitter is word 0, the y jitter word 1. */
static void gen_ray(const orc_camera *cam, int px, int py, int W, int H, const real u[4], vec3 *O, vec3 *D) {
    real dw = (real)1 / (real)W, dh = (real)1 / (real)H;
    vec3 p = vadd(vadd(vf(cam->upper_left), vmuls(vmuls(vf(cam->right), (real)px + u[0]), dw)),
                  vmuls(vmuls(vf(cam->down), (real)py + u[1]), dh));
    vec3 c = vf(cam->center);
    *O = c;
    *D = vnormalized(vsub(p, c));
}

/* linearToGamma + convertRTColor, reference src/Camera.cpp:11-34 */
static uint8_t to_byte(real x) {
    real g = x > 0 ? r_sqrt(x) : 0;
    if (g < (real)0.000) g = (real)0.000;
    if (g > (real)0.999) g = (real)0.999;
    return (uint8_t)((real)256 * g);
}

ORC_API void orc_convert_color(const double rgb[3], uint8_t out[4]) {
    out[0] = to_byte((real)rgb[0]);
    out[1] = to_byte((real)rgb[1]);
    out[2] = to_byte((real)rgb[2]);
    out[3] = 255;
}

/* Camera::renderParallel + renderPixelColor, reference src/Camera.cpp:69-88,105-122.
 * gm::setThreadSeed(pixelId) (:84) becomes "the stream is keyed on pixelId".
 *   iterative != 0 selects ray_color_iter (requires samples_per_scatter == 1).
 *   n_prims > 0 renders the general primitive list `prims` (+ polygon vertices `verts`) instead of the
 *   sphere arrays; selected primitives get the halo on primary rays.
 *   n_lights > 0 enables Camera::computeDirectLighting (enableLDirect) with those lights.
 *   rows [y0, y1) are rendered; rgba/accum are full W*H buffers (accum: W*H*3 doubles,
 *   per-pixel sample SUM; may be NULL).  sample_offset shifts the sample index.
 * Returns the number of hitClosest calls. */
ORC_API uint64_t orc_render(int n, const orc_sphere *sph, const uint32_t *mat_id, int n_mats,
                            const orc_material *mats, int n_lights, const orc_light *lights,
                            int n_prims, const orc_prim *prims, const float *verts,
                            const orc_camera *cam, int W, int H, int spp,
                            int samples_per_scatter, int max_depth, uint32_t seed, int sample_offset,
                            int iterative, int y0, int y1, uint8_t *rgba, double *accum) {
    scene_t s = {n, sph, mat_id, n_mats, mats, n_lights, lights, vf(cam->center), n_prims, prims, verts};
    uint64_t total = 0;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
    for (int pid = y0 * W; pid < y1 * W; pid++) {
        int px = pid % W, py = pid / W;
        trace_ctx tc = {&s, samples_per_scatter, 0, max_depth};
        vec3 sum = v3(0, 0, 0);
        for (int smp = 0; smp < spp; smp++) {
            rng_t g = {(uint32_t)pid, (uint32_t)(smp + sample_offset), 0, seed};
            real u[4];
            rng_next(&g, u);
            vec3 O, D;
            gen_ray(cam, px, py, W, H, u, &O, &D);
            vec3 col = iterative ? ray_color_iter(&tc, &g, O, D, max_depth)
                                 : ray_color(&tc, &g, O, D, max_depth);
            sum = vadd(sum, col);
        }
        if (accum) {
            accum[3 * (size_t)pid + 0] = (double)sum.x;
            accum[3 * (size_t)pid + 1] = (double)sum.y;
            accum[3 * (size_t)pid + 2] = (double)sum.z;
        }
        if (rgba) {
            vec3 mean = vdivs(vmuls(sum, (real)1.0), (real)spp);
            rgba[4 * (size_t)pid + 0] = to_byte(mean.x);
            rgba[4 * (size_t)pid + 1] = to_byte(mean.y);
            rgba[4 * (size_t)pid + 2] = to_byte(mean.z);
            rgba[4 * (size_t)pid + 3] = 255;
        }
        total += tc.rays;
    }
    return total;
}

/* SceneManager::hitClosest(ray, Interval(tmin, +inf), rec, false) for a batch of rays
 * (reference src/RayTracer.cpp:68-82).  Outputs double precision; `margin` (optional)
 * receives a measure of how close the decision was to flipping:
 * min over spheres of |disc|/(hb^2 + |a*c|), and over hit spheres of the relative
 * distance of a root to tmin or to the winning t — small values mark rays whose id a
 * lower-precision evaluation may legitimately get wrong. */
ORC_API void orc_first_hit(int n, const orc_sphere *sph, int n_prims, const orc_prim *prims, const float *verts,
                           int expanded_state, int n_rays, const float *org, const float *dir,
                           double tmin, int32_t *id, double *t, double *normal, uint8_t *front,
                           uint8_t *hit_expanded, double *margin) {
    scene_t s = {n, sph, NULL, 0, NULL, 0, NULL, {0, 0, 0}, n_prims, prims, verts};
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rays; i++) {
        vec3 O = vf(org + 3 * i), D = vf(dir + 3 * i);
        hitrec rec;
        int any = hit_closest_ex(&s, O, D, (real)tmin, (real)INFINITY, expanded_state, &rec);
        id[i] = any ? rec.prim : -1;
        if (hit_expanded) hit_expanded[i] = any ? (uint8_t)rec.hit_expanded : 0;
        if (t) t[i] = any ? (double)rec.time : 0.0;
        if (normal) {
            normal[3 * i + 0] = any ? (double)rec.normal.x : 0.0;
            normal[3 * i + 1] = any ? (double)rec.normal.y : 0.0;
            normal[3 * i + 2] = any ? (double)rec.normal.z : 0.0;
        }
        if (front) front[i] = any ? (uint8_t)rec.front : 0;
        if (margin && n_prims == 0) {
            double best = any ? (double)rec.time : INFINITY;
            double m = INFINITY;
            for (int k = 0; k < n; k++) {
                double cx = sph[k].cx, cy = sph[k].cy, cz = sph[k].cz, r = sph[k].r;
                double ox = (double)O.x - cx, oy = (double)O.y - cy, oz = (double)O.z - cz;
                double a = (double)vdot(D, D);
                double hb = ox * (double)D.x + oy * (double)D.y + oz * (double)D.z;
                double c = ox * ox + oy * oy + oz * oz - r * r;
                double disc = hb * hb - a * c;
                double scale = hb * hb + fabs(a * c);
                double rel = scale > 0 ? fabs(disc) / scale : 0.0;
                if (rel < m) m = rel;
                if (disc >= 0) {
                    double sq = sqrt(disc);
                    double roots[2] = {(-hb - sq) / a, (-hb + sq) / a};
                    for (int q = 0; q < 2; q++) {
                        double x = roots[q];
                        double d1 = fabs(x - tmin) / fmax(fabs(x), tmin);
                        if (d1 < m) m = d1;
                        if (k != id[i] && isfinite(best)) {
                            double d2 = fabs(x - best) / fmax(fabs(best), 1e-30);
                            if (d2 < m) m = d2;
                        }
                    }
                }
            }
            margin[i] = m;
        }
    }
}

/* RTMaterial::scatter for a batch of hits with explicit uniforms (KATs / GPU parity).
 * All vectors n x 3 doubles; u is n x 3 (randomUnit u0,u1; randomDouble u2). */
ORC_API void orc_scatter(int n, const orc_material *mats, const uint32_t *mat, const double *in_dir,
                         const double *point, const double *normal, const uint8_t *front,
                         const double *u, double *att, double *out_org, double *out_dir,
                         uint8_t *scattered) {
    for (int i = 0; i < n; i++) {
        hitrec rec;
        memset(&rec, 0, sizeof rec);
        rec.point = v3((real)point[3 * i], (real)point[3 * i + 1], (real)point[3 * i + 2]);
        rec.normal = v3((real)normal[3 * i], (real)normal[3 * i + 1], (real)normal[3 * i + 2]);
        rec.front = front[i];
        rec.object = 0;
        rec.time = 0;
        real uu[4] = {(real)u[3 * i], (real)u[3 * i + 1], (real)u[3 * i + 2], 0};
        vec3 a = v3(0, 0, 0), so = v3(0, 0, 0), sd = v3(0, 0, 0);
        vec3 d = v3((real)in_dir[3 * i], (real)in_dir[3 * i + 1], (real)in_dir[3 * i + 2]);
        int ok = scatter(&mats[mat[i]], d, &rec, uu, &a, &so, &sd);
        scattered[i] = (uint8_t)ok;
        att[3 * i] = (double)a.x; att[3 * i + 1] = (double)a.y; att[3 * i + 2] = (double)a.z;
        out_org[3 * i] = (double)so.x; out_org[3 * i + 1] = (double)so.y; out_org[3 * i + 2] = (double)so.z;
        out_dir[3 * i] = (double)sd.x; out_dir[3 * i + 1] = (double)sd.y; out_dir[3 * i + 2] = (double)sd.z;
    }
}

/* One sample's radiance, both formulations (cross-check of the loop unrolling). */
ORC_API void orc_sample_radiance(int n, const orc_sphere *sph, const uint32_t *mat_id, int n_mats,
                                 const orc_material *mats, const orc_camera *cam, int W, int H, int pid,
                                 int smp, int samples_per_scatter, int max_depth, uint32_t seed,
                                 int iterative, double out[3]) {
    scene_t s = {n, sph, mat_id, n_mats, mats, 0, NULL, vf(cam->center), 0, NULL, NULL};
    trace_ctx tc = {&s, samples_per_scatter, 0, max_depth};
    rng_t g = {(uint32_t)pid, (uint32_t)smp, 0, seed};
    real u[4];
    rng_next(&g, u);
    vec3 O, D;
    gen_ray(cam, pid % W, pid / W, W, H, u, &O, &D);
    vec3 c = iterative ? ray_color_iter(&tc, &g, O, D, max_depth) : ray_color(&tc, &g, O, D, max_depth);
    out[0] = (double)c.x; out[1] = (double)c.y; out[2] = (double)c.z;
}

/* Primary rays as genRay makes them (for first-hit tests at jittered or centred positions).
 * jitter == 0 puts the ray through the pixel centre. */
ORC_API void orc_primary_rays(const orc_camera *cam, int W, int H, uint32_t seed, int smp, int jitter,
                              float *org, float *dir) {
    for (int pid = 0; pid < W * H; pid++) {
        real u[4] = {(real)0.5, (real)0.5, 0, 0};
        if (jitter) {
            rng_t g = {(uint32_t)pid, (uint32_t)smp, 0, seed};
            rng_next(&g, u);
        }
        vec3 O, D;
        gen_ray(cam, pid % W, pid / W, W, H, u, &O, &D);
        org[3 * pid] = (float)O.x; org[3 * pid + 1] = (float)O.y; org[3 * pid + 2] = (float)O.z;
        dir[3 * pid] = (float)D.x; dir[3 * pid + 1] = (float)D.y; dir[3 * pid + 2] = (float)D.z;
    }
}

/* Camera::updateViewPort, reference src/Camera.cpp:206-221 (direction is stored
 * normalised by setDirection, :233-236). */
ORC_API void orc_update_viewport(const double center[3], const double direction[3], orc_camera *out) {
    vec3 c = v3((real)center[0], (real)center[1], (real)center[2]);
    vec3 d = vnormalized(v3((real)direction[0], (real)direction[1], (real)direction[2]));
    vec3 up = v3(0, 0, 1);
    if (fabs((double)vdot(d, up)) > 0.999) up = v3(0, 1, 0);
    vec3 right = vnormalized(vcross(d, up));
    vec3 down = vnormalized(vcross(d, right));
    vec3 ul = vsub(vsub(vadd(c, vmuls(d, 1)), vmuls(vmuls(right, 1), (real)0.5)), vmuls(vmuls(down, 1), (real)0.5));
    out->center[0] = (float)c.x; out->center[1] = (float)c.y; out->center[2] = (float)c.z;
    out->upper_left[0] = (float)ul.x; out->upper_left[1] = (float)ul.y; out->upper_left[2] = (float)ul.z;
    out->right[0] = (float)right.x; out->right[1] = (float)right.y; out->right[2] = (float)right.z;
    out->down[0] = (float)down.x; out->down[1] = (float)down.y; out->down[2] = (float)down.z;
}

ORC_API long orc_refract_clamped(void) { return g_refract_clamped; }
ORC_API int orc_real_bytes(void) { return (int)sizeof(real); }
