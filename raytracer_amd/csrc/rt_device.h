// rt_device.h — device-side building blocks of the gfx950 path tracer:
// counter-based RNG, vec3 helpers, the ray/sphere test, the four scatter
// functions and the tone-map.  Each block names the reference lines it follows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct V3 {
    float x, y, z;
};

__host__ __device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__host__ __device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__host__ __device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__host__ __device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
// a + b*s
__host__ __device__ __forceinline__ V3 madd(V3 a, V3 b, float s) {
    return mk(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z));
}
__device__ __forceinline__ V3 normalized(V3 a) { return a * __frsqrt_rn(dot(a, a)); }
__host__ __device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Scene data never changes while a kernel runs.  Reading it through the constant address space tells
// the compiler so: wherever the address is the same in every lane (loops over the primitive list, the
// lights, a small sphere list) it then uses scalar loads, which return into SGPRs through the scalar
// cache and cost neither vector registers nor vector-memory latency.
template <class T>
using cptr = const T __attribute__((address_space(4))) *;
template <class T>
__host__ __device__ __forceinline__ cptr<T> as_const(const T *p) {
    return (cptr<T>)(uintptr_t)p;
}

typedef float cfloat4 __attribute__((ext_vector_type(4)));
typedef uint32_t cuint4 __attribute__((ext_vector_type(4)));
// float4 number i of a constant array (class types cannot be copied out of another address space; the
// compiler's own vector types can)
__device__ __forceinline__ float4 load_const4(cptr<float4> p, int i) {
    const cfloat4 v = ((cptr<cfloat4>)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}

// Device copy of one material: three 16-byte words.
struct DMat {
    float4 albedo_type; // albedo rgb, type in the bits of .w
    float4 emission_fuzz; // emission rgb, fuzz
    float4 ior_diffuse;   // ior, RTMaterial::diffuse_ rgb (read by direct lighting only)
};

// Device copy of one Phong point light (class Light, inc/RTObjects.h:220-226).
struct DLight {
    float4 pos_pow;  // position, viewLightPow_
    float4 ambient;  // ambientIntensity_
    float4 diffuse;  // defuseIntensity_
};

// ---- random numbers ------------------------------------------------------
// pcg4d (Jarzynski & Olano, JCGT 2020) keyed on (pixel, sample, call, seed): one
// evaluation yields the four words a bounce can consume.  Stands in for the
// per-thread generator the reference seeds per pixel (src/Camera.cpp:84).
constexpr uint32_t RNG_MUL = 1664525u, RNG_ADD = 1013904223u;
// first step of the generator: a key word through one linear congruential step.  A path keeps its
// pixel and sample words in this form, and its call word too: the next call's is this plus RNG_MUL.
__host__ __device__ __forceinline__ uint32_t rng_key(uint32_t v) { return v * RNG_MUL + RNG_ADD; }

// the mixing rounds on key words that have been through rng_key()
__host__ __device__ __forceinline__ void rng4_keyed(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t &o0,
                                                    uint32_t &o1, uint32_t &o2, uint32_t &o3) {
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    o0 = x; o1 = y; o2 = z; o3 = w;
}

__host__ __device__ __forceinline__ void rng4(uint32_t pixel, uint32_t sample, uint32_t call, uint32_t seed,
                                              uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
    uint32_t x = rng_key(pixel);
    uint32_t y = rng_key(sample);
    uint32_t z = rng_key(call);
    uint32_t w = rng_key(seed);
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    o0 = x; o1 = y; o2 = z; o3 = w;
}

// 24-bit uniform in [0,1), exactly representable in float and double.
__host__ __device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// Uniform direction on the unit sphere from two uniforms (the role of
// IVec3f::randomUnit(), called at inc/RTMaterial.h:72,111).
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so u1 feeds them directly.
__device__ __forceinline__ V3 random_unit(float u0, float u1) {
    float z = fmaf(-2.0f, u0, 1.0f);
    float r = __fsqrt_rn(fmaxf(fmaf(-z, z, 1.0f), 0.0f));
    float s = __builtin_amdgcn_sinf(u1);
    float c = __builtin_amdgcn_cosf(u1);
    return mk(r * c, r * s, z);
}

// ---- ray / sphere ----------------------------------------------------------
// The scan works on a unit-length copy of the ray direction: with dh = d/|d| the
// quadratic of SphereObject::hitDetail (inc/RTObjects.h:135-149) has a = 1 and its
// roots are the reference's roots times |d|, so the interval (tmin, closest) is
// scaled by |d| on entry and the winning root by 1/|d| on exit.
//
// The scan itself is a FILTER.  With |dh| = 1 the discriminant is the squared radius
// minus the squared distance from the centre to the ray's line,
//   disc = r^2 - |C_perp - O_perp|^2          (x_perp = x - dh (x.dh))
//        = (r^2 - |C|^2) + (C.dh)^2 + 2 C.O_perp - |O_perp|^2,
// i.e. a per-sphere constant q, two dot products with per-ray vectors, and a per-ray
// constant: seven fused multiply-adds per sphere and one comparison against |O_perp|^2.
// The terms are of the size of |C|^2 and cancel, so this form is only trusted up to a
// margin: q is stored enlarged by FILTER_MARGIN*(|C|^2 + r^2), the per-ray threshold is
// lowered by FILTER_MARGIN*|O|^2 (about four times the worst rounding error of the
// chain), and every sphere that passes is decided by sphere_root(), which forms the
// discriminant from O - C, where rounding errors enter linearly in the coordinates
// (on the spot in the LDS-staged scan, from the candidate queue in the scalar-load scan).
//
// Spheres are stored in PAIRS so that one packed-FP32 instruction (v_pk_mul_f32 /
// v_pk_fma_f32) serves two spheres:
//   word 2j   = (x0, x1, y0, y1)      word 2j+1 = (z0, z1, q0, q1)
// and their squared radii, which only the LDS-staged scan's decision reads, in a separate
// array of one float4 per group of four spheres.
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr float FILTER_MARGIN = 1.9073486328125e-6f; // 2^-19

// Ray operands of the scan, replicated into both halves of a register pair: the unit
// direction dh, twice the part of the origin perpendicular to it, and the threshold.
struct RayK {
    f2 dx, dy, dz, kx, ky, kz;
    float thr;
};

__device__ __forceinline__ RayK make_rayk(V3 o, V3 dh) {
    RayK r;
    float od = dot(o, dh);
    V3 op = madd(o, dh, -od); // O_perp
    r.dx = (f2){dh.x, dh.x}; r.dy = (f2){dh.y, dh.y}; r.dz = (f2){dh.z, dh.z};
    r.kx = (f2){2.0f * op.x, 2.0f * op.x}; r.ky = (f2){2.0f * op.y, 2.0f * op.y}; r.kz = (f2){2.0f * op.z, 2.0f * op.z};
    r.thr = fmaf(-FILTER_MARGIN, dot(o, o), dot(op, op));
    return r;
}

// Filter value of two spheres in 7 packed instructions; the sphere can only be hit if
// the value reaches RayK::thr.  One chain, so that every instruction reads a single
// scalar register pair (the sphere words arrive in SGPRs).
__device__ __forceinline__ f2 pair_filter(const RayK &r, float4 A, float4 B) {
    f2 cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, q = {B.z, B.w};
    f2 b = __builtin_elementwise_fma(cz, r.dz, __builtin_elementwise_fma(cy, r.dy, cx * r.dx));
    f2 t = __builtin_elementwise_fma(b, b, q);
    return __builtin_elementwise_fma(cz, r.kz, __builtin_elementwise_fma(cy, r.ky, __builtin_elementwise_fma(cx, r.kx, t)));
}

// Tuning build only (-DRT_SECTIONS): wave cycles per part of the render loop, from s_memtime stamps.
#ifdef RT_SECTIONS
__device__ unsigned long long g_sections[24];
struct SecClock {
    unsigned long long t[24];
    unsigned long long last;
    int cur;
    __device__ __forceinline__ void start() {
        for (int k = 0; k < 24; ++k) t[k] = 0;
        last = __builtin_amdgcn_s_memtime();
        cur = 7;
    }
    __device__ __forceinline__ void mark(int k) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        t[cur] += now - last;
        last = now;
        cur = k;
    }
};
#else
struct SecClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
};
#endif
#ifdef RT_STATS
// tuning build only: wave-level event counters
__device__ unsigned long long g_stats[8];
__device__ unsigned long long g_times[8]; // first start, first end, last end, sum of durations, waves, max items
__device__ unsigned long long g_turns[8]; // turns without a scan, their lanes with a decided primary ray, their lanes waiting
                                          // for a scan, primaries decided as miss / as hit / left to the scan
__device__ __forceinline__ void stat(int k) {
    unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) atomicAdd(&g_stats[k], 1ull);
}
__device__ __forceinline__ void stat_lanes(int k) {
    unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) atomicAdd(&g_stats[k], (unsigned long long)__popcll(m));
}
#ifdef RT_STATS_TIMES_ONLY
#define RT_STAT(k)
#define RT_STAT_LANES(k)
#else
#define RT_STAT(k) stat(k)
#define RT_STAT_LANES(k) stat_lanes(k)
#endif
#else
#define RT_STAT(k)
#define RT_STAT_LANES(k)
#endif

// The decision for one sphere that passed the filter: the roots of
// SphereObject::hitDetail (inc/RTObjects.h:135-149) with Interval::surrounds' open
// interval (inc/RTGeometry.h:79-81), nearer root first.
// With oc = O - C: hb = oc.dh, the discriminant is r^2 - |oc - hb dh|^2 (accumulated
// onto r^2 inside the fused multiply-adds) and c = |oc|^2 - r^2 likewise; the roots
// -hb -+ sqrt(disc) are taken as q and c/q with |q| = |hb| + sqrt(disc), so the root
// near the origin does not lose its digits to cancellation.
// `self_id` names the sphere whose surface the ray starts on (-1: none): its constant
// term is exactly zero in real arithmetic but rounding noise of size ulp(|oc|)*r in
// floats, so it is skipped here and handled in closed form by self_hit().
// Returns the root the reference would take, in units of |d| (a NaN or a value outside
// (tmin, .) when there is none).
__device__ __forceinline__ float sphere_root(V3 o, V3 dh, float cx, float cy, float cz, float r2, float tmin) {
    float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
    float hb = fmaf(ocx, dh.x, fmaf(ocy, dh.y, ocz * dh.z));
    float px = fmaf(-hb, dh.x, ocx), py = fmaf(-hb, dh.y, ocy), pz = fmaf(-hb, dh.z, ocz);
    float disc = fmaf(-px, px, fmaf(-py, py, fmaf(-pz, pz, r2)));
    // a negative discriminant needs no branch: its square root is NaN, and no comparison the
    // callers make holds for a NaN
    float c = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -r2)));
    float sq = __builtin_amdgcn_sqrtf(disc);
    float q = -(hb + __builtin_copysignf(sq, hb)); // |q| = |hb| + sq
    float r1 = c * __builtin_amdgcn_rcpf(q);
    float t_near = fminf(q, r1), t_far = fmaxf(q, r1);
    // the reference takes the nearer root if it lies in (tmin, closest), else the farther one if
    // that does; a nearer root >= closest rules the farther one out as well
    return t_near > tmin ? t_near : t_far;
}

__device__ __forceinline__ void resolve_hit(V3 o, V3 dh, float cx, float cy, float cz, float r2, float f, float thr,
                                            float tmin, int idx, int self_id, float &best_t, int &best_id) {
    if (f >= thr) {
        RT_STAT(4);       // sphere bodies entered (per wave)
        RT_STAT_LANES(5); // ... (per lane)
        float cand = sphere_root(o, dh, cx, cy, cz, r2, tmin);
        if (cand > tmin && cand < best_t && idx != self_id) {
            RT_STAT_LANES(6); // closest-hit updates (per lane)
            best_t = cand;
            best_id = idx;
        }
    }
}

// Filter chains of four spheres (two pair-words each), written one step at a time so that two
// groups can be interleaved: consecutive packed instructions that depend on each other cost a
// wait state, and a wave that has a SIMD to itself needs independent work to fill the pipeline.
struct QuadF {
    f2 F0, F1;
};
#define RT_QUAD_STEP(Q, OP0, OP1) Q.F0 = OP0; Q.F1 = OP1

// LDS-staged scan: decide, on the spot, every sphere of a group that passed the filter.
__device__ __forceinline__ void resolve_quad(const RayK &r, V3 o, V3 dh, float4 A0, float4 B0, float4 A1, float4 B1,
                                             float4 R, const QuadF &q, float tmin, int idx, int self_id, float &best_t,
                                             int &best_id) {
    float m = fmaxf(fmaxf(q.F0.x, q.F0.y), fmaxf(q.F1.x, q.F1.y));
    if (m >= r.thr) {
        RT_STAT(2);       // quad bodies entered (per wave)
        RT_STAT_LANES(3); // ... (per lane)
        resolve_hit(o, dh, A0.x, A0.z, B0.x, R.x, q.F0.x, r.thr, tmin, idx, self_id, best_t, best_id);
        resolve_hit(o, dh, A0.y, A0.w, B0.y, R.y, q.F0.y, r.thr, tmin, idx + 1, self_id, best_t, best_id);
        resolve_hit(o, dh, A1.x, A1.z, B1.x, R.z, q.F1.x, r.thr, tmin, idx + 2, self_id, best_t, best_id);
        resolve_hit(o, dh, A1.y, A1.w, B1.y, R.w, q.F1.y, r.thr, tmin, idx + 3, self_id, best_t, best_id);
    }
}

// Four spheres (two pair-words each, and their squared radii R) against one ray.
__device__ __forceinline__ void test_quad(const RayK &r, V3 o, V3 dh, float4 A0, float4 B0, float4 A1, float4 B1,
                                          float4 R, float tmin, int idx, int self_id, float &best_t, int &best_id) {
    f2 x0 = {A0.x, A0.y}, y0 = {A0.z, A0.w}, z0 = {B0.x, B0.y}, q0 = {B0.z, B0.w};
    f2 x1 = {A1.x, A1.y}, y1 = {A1.z, A1.w}, z1 = {B1.x, B1.y}, q1 = {B1.z, B1.w};
    QuadF q;
    RT_QUAD_STEP(q, x0 * r.dx, x1 * r.dx);
    RT_QUAD_STEP(q, __builtin_elementwise_fma(y0, r.dy, q.F0), __builtin_elementwise_fma(y1, r.dy, q.F1));
    RT_QUAD_STEP(q, __builtin_elementwise_fma(z0, r.dz, q.F0), __builtin_elementwise_fma(z1, r.dz, q.F1));
    RT_QUAD_STEP(q, __builtin_elementwise_fma(q.F0, q.F0, q0), __builtin_elementwise_fma(q.F1, q.F1, q1));
    RT_QUAD_STEP(q, __builtin_elementwise_fma(x0, r.kx, q.F0), __builtin_elementwise_fma(x1, r.kx, q.F1));
    RT_QUAD_STEP(q, __builtin_elementwise_fma(y0, r.ky, q.F0), __builtin_elementwise_fma(y1, r.ky, q.F1));
    RT_QUAD_STEP(q, __builtin_elementwise_fma(z0, r.kz, q.F0), __builtin_elementwise_fma(z1, r.kz, q.F1));
    resolve_quad(r, o, dh, A0, B0, A1, B1, R, q, tmin, idx, self_id, best_t, best_id);
}

// Two groups of four spheres at once: the same arithmetic as two test_quad calls, with the
// chains of the two groups interleaved (dependent instructions four apart).
__device__ __forceinline__ void test_quad2(const RayK &r, V3 o, V3 dh, float4 A0, float4 B0, float4 A1, float4 B1,
                                           float4 Ra, float4 C0, float4 D0, float4 C1, float4 D1, float4 Rb, float tmin,
                                           int idx, int self_id, float &best_t, int &best_id) {
    f2 x0 = {A0.x, A0.y}, y0 = {A0.z, A0.w}, z0 = {B0.x, B0.y}, q0 = {B0.z, B0.w};
    f2 x1 = {A1.x, A1.y}, y1 = {A1.z, A1.w}, z1 = {B1.x, B1.y}, q1 = {B1.z, B1.w};
    f2 x2 = {C0.x, C0.y}, y2 = {C0.z, C0.w}, z2 = {D0.x, D0.y}, q2 = {D0.z, D0.w};
    f2 x3 = {C1.x, C1.y}, y3 = {C1.z, C1.w}, z3 = {D1.x, D1.y}, q3 = {D1.z, D1.w};
    QuadF a, b;
    RT_QUAD_STEP(a, x0 * r.dx, x1 * r.dx);
    RT_QUAD_STEP(b, x2 * r.dx, x3 * r.dx);
    RT_QUAD_STEP(a, __builtin_elementwise_fma(y0, r.dy, a.F0), __builtin_elementwise_fma(y1, r.dy, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(y2, r.dy, b.F0), __builtin_elementwise_fma(y3, r.dy, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(z0, r.dz, a.F0), __builtin_elementwise_fma(z1, r.dz, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(z2, r.dz, b.F0), __builtin_elementwise_fma(z3, r.dz, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(a.F0, a.F0, q0), __builtin_elementwise_fma(a.F1, a.F1, q1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(b.F0, b.F0, q2), __builtin_elementwise_fma(b.F1, b.F1, q3));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(x0, r.kx, a.F0), __builtin_elementwise_fma(x1, r.kx, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(x2, r.kx, b.F0), __builtin_elementwise_fma(x3, r.kx, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(y0, r.ky, a.F0), __builtin_elementwise_fma(y1, r.ky, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(y2, r.ky, b.F0), __builtin_elementwise_fma(y3, r.ky, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(z0, r.kz, a.F0), __builtin_elementwise_fma(z1, r.kz, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(z2, r.kz, b.F0), __builtin_elementwise_fma(z3, r.kz, b.F1));
    // one test for both groups first: it keeps the two chains together ahead of the branch
    float m = fmaxf(fmaxf(fmaxf(a.F0.x, a.F0.y), fmaxf(a.F1.x, a.F1.y)), fmaxf(fmaxf(b.F0.x, b.F0.y), fmaxf(b.F1.x, b.F1.y)));
    if (m >= r.thr) {
        RT_STAT(7); // groups of eight entered (per wave)
        resolve_quad(r, o, dh, A0, B0, A1, B1, Ra, a, tmin, idx, self_id, best_t, best_id);
        resolve_quad(r, o, dh, C0, D0, C1, D1, Rb, b, tmin, idx + 4, self_id, best_t, best_id);
    }
}

// ---- deferred decisions: the wave's candidate queue --------------------------------------
// Inside the scan a sphere passes the filter for about one lane in six, so deciding it on the spot
// runs the whole decision with most lanes masked off.  The scalar-load scan therefore only RECORDS
// what passed: every (lane, sphere) pair goes into a queue in LDS (ballot + prefix count), and the
// queue is worked off 64 entries at a time with every lane busy: lane j takes entry j, fetches the
// ray of the entry's lane through ds_bpermute and the sphere from memory, and folds the root into
// the ray's closest hit by a 64-bit minimum in LDS on (t bits << 32 | sphere index): the smallest
// t wins and, on equal t, the lowest index, which is what SceneManager::hitClosest's strict '<' in
// insertion order gives (src/RayTracer.cpp:77-82).  A root is positive (> tmin > 0), so its bits
// order like its value.
constexpr int CAND_CAP = 64 * 9; // a group of eight appends at most 8*64 entries to a queue holding fewer than 64
constexpr unsigned long long NO_HIT_KEY = ~0ull;

struct CandQ {
    unsigned *ent;            // CAND_CAP entries: lane | sphere << 6
    unsigned long long *slot; // per lane of the wave: the closest hit so far, NO_HIT_KEY = none
    // the matrix-core filter numbers its candidates by table row: row -> sphere index (null: the entry holds
    // the sphere index itself, and the array handed to resolve_batch() is the scene's sphere list)
    const int *row_idx;
};

__device__ __forceinline__ int lane_rank(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Must be called by the whole wave (it ballots).
__device__ __forceinline__ void push_cand(const CandQ &cq, int &count, float f, float thr, int idx, int lane) {
    const bool pass = f >= thr;
    const unsigned long long m = __ballot(pass);
    if (m) {
        RT_STAT(4);
        if (pass) {
            RT_STAT_LANES(5);
            cq.ent[count + lane_rank(m)] = (unsigned)lane | ((unsigned)idx << 6);
        }
        count += (int)__popcll(m);
    }
}

__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// Works off the last n (<= 64) entries of the queue.  Whole wave, all lanes enabled: ds_bpermute
// returns nothing from a disabled lane.
__device__ __forceinline__ void resolve_batch(const CandQ &cq, int &count, int n, int lane, V3 o, V3 dh, float tmin,
                                              int self_id, const float4 *__restrict__ sph) {
    const int head = count - n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned e = cq.ent[head + min(lane, n - 1)];
    const int src = (int)(e & 63u) << 2;
    const int row = (int)(e >> 6);
    const float4 sp = sph[row];
    const int s = cq.row_idx ? cq.row_idx[row] : row;
    const V3 ro = mk(lane_fetch(src, o.x), lane_fetch(src, o.y), lane_fetch(src, o.z));
    const V3 rd = mk(lane_fetch(src, dh.x), lane_fetch(src, dh.y), lane_fetch(src, dh.z));
    const float rt = lane_fetch(src, tmin);
    const int rself = __builtin_amdgcn_ds_bpermute(src, self_id);
    const float cand = sphere_root(ro, rd, sp.x, sp.y, sp.z, sp.w * sp.w, rt);
    if (lane < n && cand > rt && cand < __builtin_inff() && s != rself) {
        RT_STAT_LANES(6);
        atomicMin(&cq.slot[e & 63u], ((unsigned long long)__float_as_uint(cand) << 32) | (unsigned long long)(unsigned)s);
    }
    count = head;
}

// Filter of two groups of four spheres (as test_quad2); what passes is queued, not decided.
__device__ __forceinline__ void queue_quad2(const RayK &r, float4 A0, float4 B0, float4 A1, float4 B1, float4 C0,
                                            float4 D0, float4 C1, float4 D1, int idx, int lane, const CandQ &cq,
                                            int &count, V3 o, V3 dh, float tmin, int self_id,
                                            const float4 *__restrict__ sph) {
    f2 x0 = {A0.x, A0.y}, y0 = {A0.z, A0.w}, z0 = {B0.x, B0.y}, q0 = {B0.z, B0.w};
    f2 x1 = {A1.x, A1.y}, y1 = {A1.z, A1.w}, z1 = {B1.x, B1.y}, q1 = {B1.z, B1.w};
    f2 x2 = {C0.x, C0.y}, y2 = {C0.z, C0.w}, z2 = {D0.x, D0.y}, q2 = {D0.z, D0.w};
    f2 x3 = {C1.x, C1.y}, y3 = {C1.z, C1.w}, z3 = {D1.x, D1.y}, q3 = {D1.z, D1.w};
    QuadF a, b;
    RT_QUAD_STEP(a, x0 * r.dx, x1 * r.dx);
    RT_QUAD_STEP(b, x2 * r.dx, x3 * r.dx);
    RT_QUAD_STEP(a, __builtin_elementwise_fma(y0, r.dy, a.F0), __builtin_elementwise_fma(y1, r.dy, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(y2, r.dy, b.F0), __builtin_elementwise_fma(y3, r.dy, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(z0, r.dz, a.F0), __builtin_elementwise_fma(z1, r.dz, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(z2, r.dz, b.F0), __builtin_elementwise_fma(z3, r.dz, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(a.F0, a.F0, q0), __builtin_elementwise_fma(a.F1, a.F1, q1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(b.F0, b.F0, q2), __builtin_elementwise_fma(b.F1, b.F1, q3));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(x0, r.kx, a.F0), __builtin_elementwise_fma(x1, r.kx, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(x2, r.kx, b.F0), __builtin_elementwise_fma(x3, r.kx, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(y0, r.ky, a.F0), __builtin_elementwise_fma(y1, r.ky, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(y2, r.ky, b.F0), __builtin_elementwise_fma(y3, r.ky, b.F1));
    RT_QUAD_STEP(a, __builtin_elementwise_fma(z0, r.kz, a.F0), __builtin_elementwise_fma(z1, r.kz, a.F1));
    RT_QUAD_STEP(b, __builtin_elementwise_fma(z2, r.kz, b.F0), __builtin_elementwise_fma(z3, r.kz, b.F1));
    // four three-way maxima decide for the eight spheres at once
    const float m1 = fmaxf(fmaxf(a.F0.x, a.F0.y), a.F1.x), m2 = fmaxf(fmaxf(a.F1.y, b.F0.x), b.F0.y);
    const float m = fmaxf(fmaxf(fmaxf(m1, m2), b.F1.x), b.F1.y);
    if (__ballot(m >= r.thr)) {
        RT_STAT(7); // groups of eight entered (per wave)
        push_cand(cq, count, a.F0.x, r.thr, idx, lane);
        push_cand(cq, count, a.F0.y, r.thr, idx + 1, lane);
        push_cand(cq, count, a.F1.x, r.thr, idx + 2, lane);
        push_cand(cq, count, a.F1.y, r.thr, idx + 3, lane);
        push_cand(cq, count, b.F0.x, r.thr, idx + 4, lane);
        push_cand(cq, count, b.F0.y, r.thr, idx + 5, lane);
        push_cand(cq, count, b.F1.x, r.thr, idx + 6, lane);
        push_cand(cq, count, b.F1.y, r.thr, idx + 7, lane);
        while (count >= 64) resolve_batch(cq, count, 64, lane, o, dh, tmin, self_id, sph);
    }
}

// The sphere a ray starts on: with the origin exactly on the surface the roots are
// 0 and -2*hb; 0 is outside (tmin, .), so only the second can be a hit.
__device__ __forceinline__ void self_hit(V3 o, V3 dh, float tmin, float4 sph, int self_id, float &best_t,
                                         int &best_id) {
    float hb = fmaf(o.x - sph.x, dh.x, fmaf(o.y - sph.y, dh.y, (o.z - sph.z) * dh.z));
    float t = -2.0f * hb;
    if (t > tmin && t < best_t) {
        best_t = t;
        best_id = self_id;
    }
}

struct Hit {
    V3 p, n;
    bool front;
};

// Hit point, outward normal and face flag for the winning sphere
// (inc/RTObjects.h:151-155, HitRecord::setFaceNormal inc/RTGeometry.h:58-61).
// (P-C)/r normalised is sign(r) * (P-C)/|P-C|, so only the sign of r is needed.
__device__ __forceinline__ Hit finish_hit(V3 o, V3 d, float t, float4 sph) {
    Hit h;
    h.p = madd(o, d, t);
    V3 pc = h.p - mk(sph.x, sph.y, sph.z);
    float k = __frsqrt_rn(dot(pc, pc));
    k = sph.w < 0.0f ? -k : k;
    V3 outward = pc * k;
    h.front = dot(d, outward) < 0.0f;
    h.n = h.front ? outward : outward * -1.0f;
    return h;
}

// ---- materials -------------------------------------------------------------
// reflect(), inc/RTGeometry.h:18-20
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return madd(v, n, -2.0f * dot(v, n)); }

// RTDielectric::reflectance (Schlick), inc/RTMaterial.h:193-197
__device__ __forceinline__ float reflectance(float cosine, float ri) {
    // v_rcp_f32 (1 ulp) instead of a correctly rounded quotient: an eleven-instruction sequence otherwise
    float r0 = (1.0f - ri) * __builtin_amdgcn_rcpf(1.0f + ri);
    r0 = r0 * r0;
    float m = 1.0f - cosine;
    float m2 = m * m;
    return fmaf(1.0f - r0, m2 * m2 * m, r0);
}

// RTMaterial::scatter for the four kinds (inc/RTMaterial.h:67-80, 105-115,
// 142-170, 205-207).  Returns false when the path ends (emissive).
// u0,u1 drive random_unit, u2 is the dielectric's reflect-or-refract draw.
__device__ __forceinline__ bool scatter(uint32_t type, V3 albedo, float fuzz, float ior, V3 d_in, const Hit &h,
                                        float u0, float u1, float u2, V3 &att, V3 &o_out, V3 &d_out) {
    att = albedo;
    if (type <= 1u) {
        // Lambertian: normal + random unit vector, not normalised (:67-80).  Metal: always scatters, even
        // below the surface: normalised mirror direction + fuzz * random unit vector (:105-115).
        // One form serves both, so a wave that holds both kinds runs it once: base + spread * unit.
        V3 base = h.n;
        float spread = 1.0f;
        if (type == 1u) {
            base = normalized(reflect(d_in, h.n));
            spread = fuzz;
        }
        V3 dir = madd(base, random_unit(u0, u1), spread);
        if (type == 0u && fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = h.n;
        o_out = h.p;
        d_out = dir;
        return true;
    }
    if (type == 2u) { // Dielectric
        // reciprocals and square roots are the plain hardware ones (1 ulp): the correctly rounded forms cost
        // 11 and 6 instructions each, issued for every lane of every wave that holds one dielectric hit
        float eta = h.front ? __builtin_amdgcn_rcpf(ior) : ior;
        V3 unit = normalized(d_in);
        float cos_t = fminf(-dot(unit, h.n), 1.0f);
        float sin_t = __builtin_amdgcn_sqrtf(fmaxf(fmaf(-cos_t, cos_t, 1.0f), 0.0f));
        bool cannot = eta * sin_t > 1.0f;
        V3 dir;
        if (cannot || reflectance(cos_t, eta) > u2) {
            dir = reflect(unit, h.n);
        } else {
            // refract(), inc/RTGeometry.h:22-34, including the 1/eta^2 rescale at :28;
            // the assert at :30 becomes a clamp at zero.
            V3 perp = (unit + h.n * cos_t) * eta;
            float k = 1.0f - dot(perp, perp);
            if (k < 0.0f) {
                perp = perp * __builtin_amdgcn_rcpf(eta * eta);
                k = 1.0f - dot(perp, perp);
            }
            dir = madd(perp, h.n, -__builtin_amdgcn_sqrtf(fmaxf(k, 0.0f)));
        }
        // origin nudged along the face normal towards the geometric outside (:165-166)
        o_out = madd(h.p, h.n, h.front ? 1e-4f : -1e-4f);
        d_out = normalized(dir);
        return true;
    }
    return false; // Emissive
}

// Light::getDirectLighting with its Lambert and Phong terms (inc/RTObjects.h:247-257, 300-320):
// ambient*diffuse_ + (max(0, cos)*defuse*diffuse_ + pow(max(0, cos(mirror(toLight), toView)), pow)) * shadow.
// toLight - 2*getOrtogonal(toLight, n) is toLight mirrored about the (unit) normal.
__device__ __forceinline__ V3 light_direct(const DLight lt, V3 cam_center, V3 p, V3 n, V3 mdiff, bool hitted) {
    V3 to_light = normalized(mk(lt.pos_pow.x, lt.pos_pow.y, lt.pos_pow.z) - p);
    V3 to_view = normalized(cam_center - p);
    V3 out = mk(lt.ambient.x, lt.ambient.y, lt.ambient.z) * mdiff;
    if (!hitted) {
        float cos_l = dot(to_light, n);
        if (cos_l > 0.0f) out = out + mk(lt.diffuse.x, lt.diffuse.y, lt.diffuse.z) * mdiff * cos_l;
        V3 refl = madd(to_light * -1.0f, n, 2.0f * cos_l);
        float cos_v = dot(normalized(refl), to_view);
        if (cos_v > 0.0f) {
            // cos_v is in (0, 1]: exp2(pow * log2(cos_v)) with the hardware's v_log_f32 / v_exp_f32 (relative error about
            // 2e-7 * pow) instead of the library's correctly rounded power, thirty instructions per shadow query
            float k = __builtin_amdgcn_exp2f(lt.pos_pow.w * __builtin_amdgcn_logf(cos_v));
            out = out + mk(k, k, k);
        }
    }
    return out;
}

// sky colour on a miss, src/Camera.cpp:155-156 (direction used un-normalised)
__device__ __forceinline__ V3 sky(V3 d) {
    float a = 0.5f * (d.y + 1.0f);
    float b = 1.0f - a;
    return mk(fmaf(0.5f, a, b), fmaf(0.7f, a, b), fmaf(1.0f, a, b));
}

// linearToGamma + clamp + byte conversion, src/Camera.cpp:11-34
__device__ __forceinline__ uint32_t to_byte(float x) {
    float g = x > 0.0f ? __fsqrt_rn(x) : 0.0f;
    g = fminf(fmaxf(g, 0.0f), 0.999f);
    return (uint32_t)(256.0f * g);
}

} // namespace rt
