// rt_amd.hip — HIP kernels (gfx950) and the C ABI declared in include/rt_amd.h.
//
// Render kernel layout
//   * the image is cut into 8-row strips (the unit of the multi-GPU partition) and the
//     strips into small square tiles; a work item is all samples of one tile, and the
//     waves of a persistent grid pull items from one global counter;
//   * a lane carries one path.  Every iteration all lanes of a wave scan the whole
//     sphere list (brute force; from 16 spheres on through the matrix-core filter of
//     rt_mfma.h over a table in LDS, else sphere words arrive through scalar loads, or from
//     LDS in the general variant with a short list); spheres that pass the filter are
//     queued per wave and decided 64 at a time with every lane busy; then shade and
//     bounce.  A lane whose path ended adds its radiance to its tile's fixed-point
//     accumulator in LDS and takes the next (pixel, sample) of the open pool; indices
//     come from a ballot + prefix count;
//   * with the matrix-core filter a new sample's primary ray is decided at once against the
//     few spheres its tile's pyramid of primary rays can meet (listed once per tile), and the
//     wave scans only when its lanes are full of bounced rays;
//   * a wave holds two tiles, so lanes freed near the end of one start on the next at
//     once; a tile whose pool is dry and whose last path has ended is tone-mapped and stored.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <new>
#include <cmath>
#include <atomic>
#include <map>
#include <mutex>

#include "../../include/rt_amd.h"
#include "rt_device.h"
#include "rt_prims.h"
#include "rt_mfma.h"

using namespace rt;

#define RT_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int WG_THREADS = 256;
constexpr int WAVES_PER_WG = 4;
#ifndef RT_ITEM_SAMPLES
#define RT_ITEM_SAMPLES 1024 // a work item is the smallest tile shape that holds this many samples
#endif
constexpr int MAX_TILE_PX = 64;              // largest pixel tile a work item may cover (8 x 8)
constexpr int RESIDENT_MAX = 4096;           // spheres kept resident in LDS (80 KiB)
constexpr int SMALL_SCENE = 8;               // up to this many spheres the scalar-load scan decides every sphere directly
constexpr int MF_MIN_SPHERES = 16;           // scenes get a matrix-core filter table from this many spheres on (measured: the packed
                                             // filter is 8 % ahead at 9 and 12 spheres, even at 16, 7 % behind at 24, 24 % at 64)
constexpr int FULL_SCALAR_MIN_SPHERES = 128;  // the general kernel takes sphere words from LDS below this many spheres
constexpr int STREAM_QUADS = 512;            // quads (4 spheres, 80 B) per streamed LDS tile (40 KiB)
constexpr int SLOT_WORDS = 3 * MAX_TILE_PX;  // one tile slot's accumulators: r, g, b planes of 64-bit sums
constexpr int PRIM_LIST_CAP = 16;             // spheres a tile's primary rays may have to be tested against (more: they take the scan)
constexpr int WAVE_LDS_WORDS = 2 * SLOT_WORDS + 8 + PRIM_LIST_CAP; // two slots, their tile descriptors and their primary-ray lists
constexpr size_t ACC_BYTES = WAVES_PER_WG * WAVE_LDS_WORDS * sizeof(unsigned long long); // per-wave accumulators
// per-wave candidate queue and closest-hit slots of the deferred sphere decisions (rt_device.h)
constexpr size_t CQ_BYTES = WAVES_PER_WG * (CAND_CAP * sizeof(unsigned) + 64 * sizeof(unsigned long long));
constexpr size_t WAVE_STATE_BYTES = ACC_BYTES + CQ_BYTES;

// Where the scan loop takes the sphere words from.
enum { SRC_LDS_RESIDENT = 0, SRC_LDS_STREAMED = 1, SRC_SCALAR = 2, SRC_MFMA = 3, SRC_MFMA_STREAMED = 4 };
constexpr bool is_mfma(int src) { return src == SRC_MFMA || src == SRC_MFMA_STREAMED; }

// The matrix-core variant shares one filter table per workgroup, so its workgroups are as large as
// they come (16 waves: one workgroup per CU at four waves per SIMD), and its queue only ever holds
// the remainder of one batch plus one turn of the recording loop.
constexpr size_t MF_LDS_BYTES = 159 * 1024; // dynamic LDS a workgroup of the matrix-core variants may use: a CU has 160 KiB, and the workgroup-wide vote of the streamed variant keeps a little static LDS
#ifndef RT_MF_STREAM_THREADS
#define RT_MF_STREAM_THREADS 1024
#endif
// The general kernel (lights, other primitive kinds, the halo) carries more state per path: its matrix-core
// variant gets three waves per SIMD (170 registers each) in a workgroup of twelve.
#ifndef RT_MF_FULL_THREADS
#define RT_MF_FULL_THREADS 768
#endif
constexpr int wg_threads_of(int src, bool full = false) {
    return src == SRC_MFMA ? (full ? RT_MF_FULL_THREADS : 1024)
           : src == SRC_MFMA_STREAMED ? (full ? RT_MF_FULL_THREADS : RT_MF_STREAM_THREADS) : WG_THREADS;
}
constexpr int cand_cap_of(int src) { return is_mfma(src) ? 128 : CAND_CAP; }
constexpr size_t acc_bytes_of(int src, bool full = false) {
    return (size_t)(wg_threads_of(src, full) / 64) * WAVE_LDS_WORDS * sizeof(unsigned long long);
}
constexpr size_t wave_state_bytes_of(int src, bool full = false) {
    return acc_bytes_of(src, full) +
           (size_t)(wg_threads_of(src, full) / 64) * (cand_cap_of(src) * sizeof(unsigned) + 64 * sizeof(unsigned long long));
}
static_assert(wave_state_bytes_of(SRC_SCALAR) == WAVE_STATE_BYTES, "the plain variants keep their layout");

// Radiance sums are kept in signed 40.24 fixed point: integer addition is exact, so a pixel's sum
// does not depend on the order in which its samples finish -- which changes with the tile that
// shares the wave, i.e. with timing -- and a frame renders to the same bytes every time and
// under every strip partition.  One sample is clamped to +-2^30 (a NaN counts as 0) and
// resolved to 2^-24; the sum holds 2^39, i.e. 2^26 samples of average radiance 8192.
// |x| in 40.24 fixed point.  For m >= 0 the conversion to unsigned truncates like floor, and
// v_fract_f32 returns m - floor(m) exactly.
__device__ __forceinline__ unsigned long long to_fixed_magnitude(float x) {
    const float m = fminf(fmaxf(fabsf(x), 0.0f), 1073741824.0f);
    const unsigned hi = (unsigned)m;
    const unsigned lo = (unsigned)(__builtin_amdgcn_fractf(m) * 16777216.0f);
    return ((unsigned long long)hi << 24) | (unsigned long long)lo;
}
// The three channels of one sample.  Radiance is hardly ever negative (it takes a negative emission or
// albedo), so the sign is applied on a path of its own that a wave enters only if one of its lanes needs it.
__device__ __forceinline__ void to_fixed3(V3 L, unsigned long long &fx, unsigned long long &fy, unsigned long long &fz) {
    // the clamped magnitudes (a NaN becomes 0)
    const float mx = fminf(fmaxf(fabsf(L.x), 0.0f), 1073741824.0f), my = fminf(fmaxf(fabsf(L.y), 0.0f), 1073741824.0f),
                mz = fminf(fmaxf(fabsf(L.z), 0.0f), 1073741824.0f);
    if (__ballot(fmaxf(fmaxf(mx, my), mz) >= 256.0f) == 0) {
        // the usual case, radiance below 256 in every lane: m * 2^24 is exact in a float and fits 32 bits, and
        // its truncation is the same number to_fixed_magnitude() assembles from integer and fractional part
        fx = (unsigned long long)(unsigned)(mx * 16777216.0f);
        fy = (unsigned long long)(unsigned)(my * 16777216.0f);
        fz = (unsigned long long)(unsigned)(mz * 16777216.0f);
    } else {
        asm volatile(""); // keeps this a branch
        fx = to_fixed_magnitude(L.x);
        fy = to_fixed_magnitude(L.y);
        fz = to_fixed_magnitude(L.z);
    }
    if (__ballot(fminf(fminf(L.x, L.y), L.z) < 0.0f)) {
        asm volatile(""); // keeps this a branch: folded into selects it would run for every sample
        fx = L.x < 0.0f ? 0ull - fx : fx;
        fy = L.y < 0.0f ? 0ull - fy : fy;
        fz = L.z < 0.0f ? 0ull - fz : fz;
    }
}
__device__ __forceinline__ float from_fixed(unsigned long long v) {
    return (float)((double)(long long)v * (1.0 / 16777216.0));
}

constexpr float TMIN = 0.001f; // CLOSEST_HIT_MIN_T, src/Camera.cpp:10

static_assert(RT_STRIP_ROWS == 8, "tile shapes below assume 8-row strips");

struct KParams {
    int width, height;
    int spp, max_depth, sample_offset;
    uint32_t seed, flags;
    int strip_first, strip_stride;
    int tile_w_log2, tile_h_log2; // work-item tile: 1x1 ... 8x8 pixels
    int tiles_x;         // tiles across the image
    int tiles_per_strip; // tiles_x * (8 / tile height)
    int n_items;         // tiles this call renders
    int n_quads;     // sphere quads to scan (padding included)
    int n_spheres;   // spheres of the scene
    int n_lights;    // lights shaded at every hit (0 unless RT_FLAG_DIRECT_LIGHT)
    int prim_min;    // primary-ray lists: a scan runs when at least this many lanes need one (0: no lists, every ray takes the scan)
    MfmaFrame mf;    // frame of the matrix-core filter table (SRC_MFMA)
    float inv_w, inv_h;
    float cam[12];   // center, upper_left, right, down
};

struct DeviceScene {
    const float4 *sph;    // cx, cy, cz, r
    const float4 *pairs;  // pair-interleaved (centre, q) words, see rt_device.h; padded with
                          // never-hit spheres to a multiple of 8 plus four spare quads
    const float4 *r2q;    // squared radii, one float4 per group of four spheres, padded alike
    const uint32_t *mat_id;
    const DMat *sph_mats; // the material of every sphere, by sphere index: one load instead of two dependent ones
    const DMat *mats;
    const DLight *lights;
    const uint4v *mf_table; // matrix-core filter table, MF_BLOCK_WORDS 16-byte words per 32 spheres
    const float4 *mf_rows;  // the sphere (cx, cy, cz, r) behind every table row ...
    const int *mf_row_idx;  // ... and its index in `sph` (rows are in the order of a space-filling curve)
    FullScene full; // general primitive list (used by the FULL kernel variants only)
};

// Everything a render launch is given, as the kernel's one argument.
struct KArgs {
    KParams p;
    DeviceScene sc;
    uchar4 *rgba;
    float4 *accum;                // may be null
    unsigned long long *rays_out; // may be null
    unsigned long long *scans_out; // may be null: receives the number of scans the waves ran (64 lanes each)
    unsigned *queue;              // work counter, zero at launch
};

// Scan `nquads` groups of four spheres held in LDS (every lane reads the same address:
// a broadcast read) for one ray.
__device__ __forceinline__ void scan_lds(const float4 *lds, const float4 *lds_r2, int nquads, int base, const RayK &r,
                                         V3 o, V3 dh, float tmin, int self_id, float &best_t, int &best_id) {
#pragma unroll 1
    for (int q = 0; q < nquads; ++q) {
        float4 A0 = lds[4 * q], B0 = lds[4 * q + 1], A1 = lds[4 * q + 2], B1 = lds[4 * q + 3];
        test_quad(r, o, dh, A0, B0, A1, B1, lds_r2[q], tmin, base + 4 * q, self_id, best_t, best_id);
    }
}

// Same scan with the sphere words fetched by scalar loads (s_load_dwordx16, one quad =
// 64 bytes per instruction) straight from global memory through the scalar cache: the
// words arrive in SGPRs and feed the packed instructions as scalar operands, so the scan
// issues no LDS read and spends no VGPR on sphere data.  Scalar loads return out of
// order, so the only safe wait is lgkmcnt(0); the loop keeps exactly one load in
// flight: wait for quad q, request quad q+1, test quad q.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

template <int BYTE_OFF>
__device__ __forceinline__ u32x16 sload_quad(const float4 *p) {
    u32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "i"(BYTE_OFF) : "memory");
    return v;
}
__device__ __forceinline__ void swait_quads(u32x16 &a, u32x16 &b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)::"memory");
}
__device__ __forceinline__ float4 word(const u32x16 &v, int k) {
    return make_float4(__uint_as_float(v[4 * k]), __uint_as_float(v[4 * k + 1]), __uint_as_float(v[4 * k + 2]),
                       __uint_as_float(v[4 * k + 3]));
}
// nquads is even (the list is padded with never-hit spheres) and four spare quads follow
// it, so the read-ahead never leaves the arrays.  Two quads are requested while two are
// tested.
// Called by the whole wave, lanes without a ray included (their threshold is +inf): what passes
// the filter is queued and decided in full batches (rt_device.h), which needs every lane.
__device__ __forceinline__ void scan_scalar(const float4 *__restrict__ g, int nquads, const RayK &r, V3 o, V3 dh,
                                            float tmin, int self_id, int lane, const CandQ &cq,
                                            const float4 *__restrict__ sph, int n_spheres, float &best_t,
                                            int &best_id) {
    if (n_spheres <= SMALL_SCENE) {
        // a handful of spheres: every lane decides every sphere outright, in insertion order with the
        // shrinking interval of hitClosest.  Filter, queue and batches would cost more than they save,
        // and the result is the same: the filter only ever drops spheres this decision rejects.
        const cptr<float4> csph = as_const(sph);
        for (int i = 0; i < n_spheres; ++i) {
            const float4 sp = load_const4(csph, i); // same address in every lane: a scalar load
            const float cand = sphere_root(o, dh, sp.x, sp.y, sp.z, sp.w * sp.w, tmin);
            if (cand > tmin && cand < best_t && i != self_id) {
                best_t = cand;
                best_id = i;
            }
        }
        return;
    }
    cq.slot[lane] = NO_HIT_KEY;
    int count = 0; // queue fill, wave-uniform
    u32x16 A0 = sload_quad<0>(g), A1 = sload_quad<64>(g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int q = 0; q < nquads; q += 4) {
        swait_quads(A0, A1);
        u32x16 B0 = sload_quad<128>(g), B1 = sload_quad<192>(g);
        __builtin_amdgcn_sched_barrier(0); // keep the requests ahead of the arithmetic they overlap with
        queue_quad2(r, word(A0, 0), word(A0, 1), word(A0, 2), word(A0, 3), word(A1, 0), word(A1, 1), word(A1, 2),
                    word(A1, 3), 4 * q, lane, cq, count, o, dh, tmin, self_id, sph);
        swait_quads(B0, B1);
        g += 16; // four quads
        A0 = sload_quad<0>(g);
        A1 = sload_quad<64>(g);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < nquads)
            queue_quad2(r, word(B0, 0), word(B0, 1), word(B0, 2), word(B0, 3), word(B1, 0), word(B1, 1), word(B1, 2),
                        word(B1, 3), 4 * q + 8, lane, cq, count, o, dh, tmin, self_id, sph);
    }
    swait_quads(A0, A1); // drain the read-ahead before the registers are reused
    while (count > 0) resolve_batch(cq, count, min(count, 64), lane, o, dh, tmin, self_id, sph);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned long long key = cq.slot[lane];
    if (key != NO_HIT_KEY) {
        best_t = __uint_as_float((unsigned)(key >> 32));
        best_id = (int)(unsigned)key;
    }
}

// Camera::genRay, src/Camera.cpp:124-135: jittered point on the 1x1 viewport.
template <class P>
__device__ __forceinline__ void gen_ray(const P &p, int px, int py, float u0, float u1, V3 &o, V3 &d) {
    V3 c = mk(p.cam[0], p.cam[1], p.cam[2]);
    V3 ul = mk(p.cam[3], p.cam[4], p.cam[5]);
    V3 right = mk(p.cam[6], p.cam[7], p.cam[8]);
    V3 down = mk(p.cam[9], p.cam[10], p.cam[11]);
    float fx = ((float)px + u0) * p.inv_w;
    float fy = ((float)py + u1) * p.inv_h;
    V3 pt = madd(madd(ul, right, fx), down, fy);
    o = c;
    d = normalized(pt - c);
}

// One path's state, carried in registers.
struct Path {
    V3 o, d;
    V3 T, L;      // throughput, radiance so far
    V3 dh;        // d / |d|
    float len, inv_len; // |d| and its reciprocal
    int pix;      // pixel index inside the wave tile, -1 = lane has no work
    // RNG key words, each already through rng_key(): global pixel id, sample index, and call index
    // (0 = ray generation, k = k-th bounce; the next call's word is this one plus RNG_MUL)
    uint32_t gpix;
    uint32_t sample;
    uint32_t call;
    int depth;    // closest-hit queries left
    int self_id;  // sphere whose surface the ray starts on exactly (diffuse / metal bounce), else -1
};

// Extra per-path state of the direct-lighting variant: while the shadow queries of a hit run, the
// lane's ray registers hold the shadow ray and the already-scattered bounce waits here.
struct LightState {
    int light_i;   // -1: the lane traces its path ray; k >= 0: the shadow ray towards light k
    V3 hp, hn;     // the shaded hit (HitRecord::point / normal)
    int hobj, hmat; // its HitRecord::object code and material index
    int hself;      // packed index of the sphere the hit lies on exactly, else -1
    V3 nxt_o, nxt_d, att; // the bounce to resume with
    int nxt_self;
    bool nxt_valid; // false: the path ends after the shadow queries
};

// Per-ray setup of the scan: unit direction, its length, and the packed operands.
__device__ __forceinline__ void set_dir(Path &pt, V3 d) {
    pt.d = d;
    float a = dot(d, d);
    pt.inv_len = __frsqrt_rn(a);
    pt.len = a * pt.inv_len;
    pt.dh = d * pt.inv_len;
}

// The tile of pixels one work item covers, decoded from the item index.
struct Tile {
    int x0, y0;   // top-left pixel
    int vw, vh;   // valid width / height inside the image (<= 0: nothing to do)
    int row0;     // output row of y0
};

template <class P>
__device__ __forceinline__ Tile decode_item(const P &p, unsigned item) {
    Tile t;
    int strip_local = (int)(item / (unsigned)p.tiles_per_strip);
    int rem = (int)(item - (unsigned)strip_local * (unsigned)p.tiles_per_strip);
    int sub_row = rem / p.tiles_x;
    int tile_x = rem - sub_row * p.tiles_x;
    int strip = p.strip_first + strip_local * p.strip_stride;
    t.x0 = tile_x << p.tile_w_log2;
    int dy = sub_row << p.tile_h_log2;
    t.y0 = strip * RT_STRIP_ROWS + dy;
    t.vw = min(1 << p.tile_w_log2, p.width - t.x0);
    t.vh = min(1 << p.tile_h_log2, p.height - t.y0);
    t.row0 = p.strip_stride == 1 ? t.y0 : strip_local * RT_STRIP_ROWS + dy;
    return t;
}

// The spheres a tile's primary rays can meet (Camera::genRay, src/Camera.cpp:124-135: every ray starts at the camera
// centre and passes through the tile's patch of the viewport): those rays fill a pyramid with its apex at the camera,
// and a sphere that meets it has its centre within |r| of the inner side of each of the pyramid's four side planes.
// A sphere is listed when that holds with |r| + 0.1 % + 10^-4 of its (L1) distance from the camera; one that contains
// the camera always does.  The test errs towards listing (a NaN passes); what is listed is decided exactly.
// Whole wave; returns the length of the list, or -1 when it would be longer than PRIM_LIST_CAP (the tile's primary
// rays then take the scan).
template <class P>
__device__ __forceinline__ int list_primary_candidates(const P &p, const Tile &t, const float4 *__restrict__ sph, int n,
                                                       int *list, int lane) {
    const V3 ul = mk(p.cam[3], p.cam[4], p.cam[5]), right = mk(p.cam[6], p.cam[7], p.cam[8]), down = mk(p.cam[9], p.cam[10], p.cam[11]);
    const V3 c = mk(p.cam[0], p.cam[1], p.cam[2]);
    const float fx0 = (float)t.x0 * p.inv_w, fx1 = (float)(t.x0 + t.vw) * p.inv_w;
    const float fy0 = (float)t.y0 * p.inv_h, fy1 = (float)(t.y0 + t.vh) * p.inv_h;
    const V3 base = ul - c;
    // outward unit normals of the four side planes: each holds the apex and one edge line of the patch, i.e. the
    // edge's direction (down or right) and any vector from the apex to the line -- two vectors at about a right
    // angle, so the cross product is well conditioned however thin the pyramid is (two neighbouring corner
    // directions would be nearly parallel); the side a normal must point to is known by construction
    V3 nrm[4] = {cross(down, madd(base, right, fx0)), cross(down, madd(base, right, fx1)),
                 cross(right, madd(base, down, fy0)), cross(right, madd(base, down, fy1))};
    const V3 out[4] = {right * -1.0f, right, down * -1.0f, down};
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float l2 = dot(nrm[k], nrm[k]);
        ok = ok && l2 > 0.0f && l2 < __builtin_inff();
        nrm[k] = nrm[k] * __frsqrt_rn(l2);
        const float side = dot(nrm[k], out[k]);
        ok = ok && side != 0.0f;
        if (side < 0.0f) nrm[k] = nrm[k] * -1.0f;
    }
    if (!ok) return -1; // a degenerate camera gets no list
    int count = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool pass = false;
        if (i < n) {
            const float4 sp = sph[i];
            const V3 v = mk(sp.x - c.x, sp.y - c.y, sp.z - c.z);
            const float reach = fabsf(sp.w) * 1.001f + 1.0e-4f * (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
            // anything not provably outside is listed: a NaN fails the comparisons and passes
            pass = !(dot(v, nrm[0]) > reach) && !(dot(v, nrm[1]) > reach) && !(dot(v, nrm[2]) > reach) && !(dot(v, nrm[3]) > reach);
        }
        const unsigned long long m = __ballot(pass);
        if (pass) {
            const int at = count + lane_rank(m);
            if (at < PRIM_LIST_CAP) list[at] = i;
        }
        count += (int)__popcll(m);
        if (count > PRIM_LIST_CAP) return -1;
    }
    return count;
}

// Persistent render kernel.  Work items are pixel tiles (all samples of 1 to 256 pixels,
// a couple of thousand samples each); every wave pulls items from one global counter until
// it runs dry, so the grid stays busy however uneven the image is.  A wave holds up to TWO
// tiles: when the pool of one runs dry, lanes that finish their paths start on the next
// tile at once while the long paths of the old one run out, so small tiles cost no idle
// lanes, and small tiles keep the longest item (the kernel cannot end before it does)
// short.  Inside a tile the wave runs the path loop described at the top of this file.
template <int SRC, bool FULL>
// the sphere-only variants are held to the register budget of five waves per SIMD
#ifndef RT_SPHERE_MIN_WAVES
#define RT_SPHERE_MIN_WAVES 5
#endif
#ifndef RT_FULL_MIN_WAVES
#define RT_FULL_MIN_WAVES 1
#endif
__global__ __launch_bounds__(wg_threads_of(SRC, FULL), is_mfma(SRC) ? (FULL ? 3 : 4) : (FULL ? RT_FULL_MIN_WAVES : RT_SPHERE_MIN_WAVES)) void render_kernel(KArgs args) {
    constexpr int WGT = wg_threads_of(SRC, FULL), NW = WGT / 64;
    // The scan needs 64 SGPRs for sphere words, so whatever else lives in SGPRs across it is spilled
    // into VGPR lanes and comes back through v_readlane, which costs VALU issue slots, the kernel's
    // bottleneck.  The launch parameters are therefore read from the kernel-argument segment (constant
    // memory, scalar loads; the single argument sits at its start) and re-read after every scan
    // instead of being kept.
    cptr<KArgs> kargs = (cptr<KArgs>)__builtin_amdgcn_kernarg_segment_ptr();
    (void)args;
#define KP (kargs->p)
#define KS (kargs->sc)
#define rgba (kargs->rgba)
#define accum (kargs->accum)
#define rays_out (kargs->rays_out)
#define scans_out (kargs->scans_out)
#define queue (kargs->queue)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per-wave state first (two accumulator slots of 3 channels x 64 pixels and their tile
    // descriptors), then the sphere words
    unsigned long long *acc_all = reinterpret_cast<unsigned long long *>(smem);
    float4 *lds_sph = reinterpret_cast<float4 *>(smem + wave_state_bytes_of(SRC, FULL));
    uint4v *lds_tab = reinterpret_cast<uint4v *>(smem + wave_state_bytes_of(SRC, FULL)); // the matrix-core variants' table / tiles
    // the squared radii follow the pair words of the resident list / of one streamed tile
    float4 *lds_r2 = lds_sph + 4 * (SRC == SRC_LDS_STREAMED ? STREAM_QUADS : KP.n_quads);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned long long *acc = acc_all + wave * WAVE_LDS_WORDS;
    int *desc = reinterpret_cast<int *>(acc + 2 * SLOT_WORDS); // slot s: desc[8 s ...] = x0, y0, vw, vh, row0, length of its primary-ray list
    int *plist = desc + 16;                                    // slot s: plist[PRIM_LIST_CAP s ...] = sphere indices
    // Primary rays of a tile all start at the camera and pass through the tile's patch of the viewport: the
    // spheres that meet the cone around that pyramid are listed once per tile, and a new sample's primary ray
    // is decided against the list at once, by the very decision the scan's candidates get.  The scan is then left to
    // bounced rays, and runs only when enough lanes hold one; meanwhile the wave shades primary hits.
    constexpr bool PRIM = is_mfma(SRC);
    CandQ cq;
    cq.slot = reinterpret_cast<unsigned long long *>(smem + acc_bytes_of(SRC, FULL)) + wave * 64;
    cq.ent = reinterpret_cast<unsigned *>(smem + acc_bytes_of(SRC, FULL) + NW * 64 * sizeof(unsigned long long)) + wave * cand_cap_of(SRC);
    cq.row_idx = is_mfma(SRC) ? KS.mf_row_idx : nullptr;

    const int nquads = KP.n_quads; // groups of four spheres, padding included
    if (SRC == SRC_LDS_RESIDENT) {
        for (int i = threadIdx.x; i < 4 * nquads; i += WGT) lds_sph[i] = KS.pairs[i];
        for (int i = threadIdx.x; i < nquads; i += WGT) lds_r2[i] = KS.r2q[i];
    }
    if (SRC == SRC_MFMA) {
        uint4v *dst = reinterpret_cast<uint4v *>(smem + wave_state_bytes_of(SRC, FULL));
        for (int i = threadIdx.x; i < (KP.mf.n_blocks + 1) * MF_BLOCK_WORDS; i += WGT) dst[i] = KS.mf_table[i];
    }
    __syncthreads();

    // tuning build (-DRT_SECTIONS): shader cycles this wave spends in each part of the loop body.  Not in the
    // scalar-load variant: a time stamp lands in SGPRs that its loads in flight own.
    SecClock sec;
    sec.start();
#define RT_SEC(k)                          \
    do {                                   \
        if (SRC != SRC_SCALAR) sec.mark(k); \
    } while (0)
#ifdef RT_STATS
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_fetch[2] = {t_start, t_start};
    unsigned long long cur_item[2] = {0, 0};
    int items_done = 0;
#endif
    Path pt;
    pt.pix = -1;
    pt.o = pt.d = pt.dh = pt.T = pt.L = mk(0, 0, 0);
    pt.len = pt.inv_len = 1.f;
    pt.gpix = pt.sample = pt.call = 0;
    pt.depth = 0;
    pt.self_id = -1;
    unsigned long long my_rays = 0;
    unsigned n_scans = 0; // wave-uniform
    float pre_t = 0.f; // closest hit of the lane's primary ray, decided from its tile's list ...
    int pre_id = -2;   // ... and its sphere (-1: none); -2: the lane's ray takes the scan
    LightState ls;
    ls.light_i = -1;
    ls.hp = ls.hn = ls.nxt_o = ls.nxt_d = ls.att = mk(0, 0, 0);
    ls.hobj = ls.hmat = 0;
    ls.hself = -1;
    ls.nxt_self = -1;
    ls.nxt_valid = false;
    const V3 cam_center = mk(KP.cam[0], KP.cam[1], KP.cam[2]);

    // wave-uniform state: two tile slots.  At most one of them hands out samples (its pool is
    // open); the other may still have paths in flight from before its pool ran dry.
    bool queue_empty = false, pool_open = false;
    bool have0 = false, have1 = false; // slot holds a tile whose result is not stored yet
    int fill = 0;                      // the slot whose pool is open
    Tile tile = {0, 0, 0, 0, 0};       // its tile
    int nvalid = 0;
    int cur_q = 0, cur_smp = 0;           // next (pixel, sample) of the open pool, pixels fastest
    float inv_nvalid = 0.f, inv_vw = 0.f; // reciprocals for the small index divisions below
    const int tile_px = 1 << (KP.tile_w_log2 + KP.tile_h_log2);

    for (;;) {
        const bool wave_live = have0 || have1 || !queue_empty;
        if (SRC == SRC_LDS_STREAMED || SRC == SRC_MFMA_STREAMED) {
            if (!__syncthreads_or(wave_live)) break;
        } else {
            if (!wave_live) break;
        }

        RT_SEC(0); // ray set-up and the scan
        const bool in_shadow = FULL && ls.light_i >= 0;
        bool active = pt.pix >= 0 && (pt.depth > 0 || in_shadow);
        // a path with no depth left (max_depth == 0) contributes black and ends at once
        bool finished = pt.pix >= 0 && pt.depth <= 0 && !in_shadow;
        // lanes whose primary hit is known need no scan; while fewer than prim_min lanes need one and some
        // primary hit waits to be shaded, the wave shades those and the other rays wait a turn
        const bool pre = PRIM && active && pre_id != -2;
        bool skip_scan = false;
        if (PRIM && KP.prim_min > 0) {
            const unsigned long long need = __ballot(active && !pre);
            // (and a wave none of whose lanes holds a ray to scan for -- the first turn, the last ones -- runs no scan)
            skip_scan = need == 0 || (__ballot(pre) != 0 && (int)__popcll(need) < KP.prim_min);
            // the waves of a workgroup that streams its table scan in step: they skip together or not at all
            // (a wave without any ray goes along with either)
            if (SRC == SRC_MFMA_STREAMED) skip_scan = __syncthreads_and(skip_scan || __ballot(active) == 0);
            if (skip_scan) active = pre;
        }

#if defined(RT_STATS) && !defined(RT_STATS_TIMES_ONLY)
        {
            unsigned long long am = __ballot(active);
            if (lane == 0) {
                atomicAdd(&g_stats[0], 1ull);                            // scan iterations (per wave)
                atomicAdd(&g_stats[1], (unsigned long long)__popcll(am)); // lanes with a ray in them
            }
        }
#endif
        // ---- closest hit over all spheres (SceneManager::hitClosest, src/RayTracer.cpp:77-82) ----
        float best_t = __builtin_inff(); // in units of |d|
        int best_id = -1;
        if (!skip_scan) n_scans++;
        const float tmin = TMIN * pt.len;
        RayK rk = make_rayk(pt.o, pt.dh);
        if constexpr (SRC == SRC_LDS_STREAMED) {
            for (int base = 0; base < nquads; base += STREAM_QUADS) {
                int cnt = min(STREAM_QUADS, nquads - base);
                __syncthreads();
                for (int i = threadIdx.x; i < 4 * cnt; i += WGT) lds_sph[i] = KS.pairs[4 * base + i];
                for (int i = threadIdx.x; i < cnt; i += WGT) lds_r2[i] = KS.r2q[base + i];
                __syncthreads();
                if (active)
                    scan_lds(lds_sph, lds_r2, cnt, 4 * base, rk, pt.o, pt.dh, tmin, pt.self_id, best_t, best_id);
            }
        } else if constexpr (SRC == SRC_LDS_RESIDENT) {
            if (active) scan_lds(lds_sph, lds_r2, nquads, 0, rk, pt.o, pt.dh, tmin, pt.self_id, best_t, best_id);
        } else if constexpr (SRC == SRC_MFMA) {
            if (!skip_scan)
                scan_mfma(lds_tab, KP.mf, pt.o, pt.dh, active && !pre, tmin, pt.self_id, lane, cq, KS.sph, KS.mf_rows, best_t,
                          best_id, sec);
            if (pre) {
                best_t = pre_t;
                best_id = pre_id;
                pre_id = -2;
            }
        } else if constexpr (SRC == SRC_MFMA_STREAMED) {
            if (!skip_scan)
                scan_mfma_streamed<WGT>(lds_tab, KS.mf_table, KP.mf, pt.o, pt.dh, active && !pre, tmin, pt.self_id, lane, cq,
                                        KS.sph, KS.mf_rows, best_t, best_id, sec);
            if (pre) {
                best_t = pre_t;
                best_id = pre_id;
                pre_id = -2;
            }
        } else {
            // the whole wave scans: a lane without a ray passes no filter
            rk.thr = active ? rk.thr : __builtin_inff();
            scan_scalar(KS.pairs, nquads, rk, pt.o, pt.dh, tmin, pt.self_id, lane, cq, KS.sph, KP.n_spheres, best_t, best_id);
        }

        // ---- shade (Camera::getRayColor, src/Camera.cpp:137-157) ----
        asm volatile("" : "+s"(kargs)); // launch parameters are fetched again from here on, not held across the scan
        RT_SEC(1);
        bool bounce = false;
        Hit h;
        h.p = h.n = mk(0, 0, 0);
        h.front = false;
        float4 at = make_float4(0, 0, 0, 0), ef = make_float4(0, 0, 0, 0);
        float ior = 1.f;
        int hit_mat = 0;
        int next_self = -1; // packed index of the sphere a diffuse / metal bounce starts on
        const bool shadow_ray = FULL && ls.light_i >= 0;
        if (active) {
            my_rays++;
            if (pt.self_id >= 0) self_hit(pt.o, pt.dh, tmin, KS.sph[pt.self_id], pt.self_id, best_t, best_id);
            // the winner of the scan; FULL: merged with the other primitive kinds and the selection halo
            bool any_hit = best_id >= 0;
            int hit_obj = best_id; // HitRecord::object code
            int hit_self = best_id;
            bool halo = false;
            Closest c;
            FullScene full; // a member-wise copy: class types cannot be copied out of the constant address space
            if (FULL) {
                full.prims = KS.full.prims;
                full.verts = KS.full.verts;
                full.others = KS.full.others;
                full.halo = KS.full.halo;
                full.sph_gid = KS.full.sph_gid;
                full.n_others = KS.full.n_others;
                full.n_halo = KS.full.n_halo;
                c.t = any_hit ? best_t * pt.inv_len : __builtin_inff();
                c.prim = any_hit ? full.sph_gid[best_id] : -1;
                c.scale = 1.0f;
                c.box_face = 0;
                c.expanded = false;
                for (int k = 0; k < full.n_others; ++k) other_hit(full, full.others[k], pt.o, pt.d, TMIN, c);
                // the halo pass runs for primary rays only (hitExpandedState, src/Camera.cpp:141)
                if (!shadow_ray && full.n_halo > 0 && pt.depth == KP.max_depth) {
                    Closest e = halo_pass(full, pt.o, pt.d, TMIN);
                    if (e.expanded && e.t < c.t) c = e; // a plain halo-pass hit is the normal pass's own hit
                }
                any_hit = c.prim >= 0;
            }
            if (!shadow_ray) {
                pt.depth--;
                if (!any_hit) {
                    pt.L = pt.L + pt.T * sky(pt.d);
                    finished = true;
                } else {
                    if (FULL) {
                        h = finish_any(full, pt.o, pt.d, c, hit_obj);
                        const DPrim pr = load_prim(full.prims, c.prim);
                        hit_mat = (int)pr.mat;
                        hit_self = (pr.type == PRIM_SPHERE && c.scale == 1.0f) ? pr.sph_local : -1;
                        halo = c.expanded;
                    } else {
                        h = finish_hit(pt.o, pt.d, best_t * pt.inv_len, KS.sph[best_id]);
                    }
                    const DMat &m = FULL ? KS.mats[hit_mat] : KS.sph_mats[best_id];
                    at = m.albedo_type;
                    ef = m.emission_fuzz;
                    ior = m.ior_diffuse.x;
                    // a halo hit glows in the selection colour instead (src/Camera.cpp:144-147)
                    pt.L = pt.L + pt.T * (halo ? mk(1.0f, 0.0f, 0.0f) : mk(ef.x, ef.y, ef.z));
                    // an emissive surface ends the path (RTEmissive::scatter, inc/RTMaterial.h:205-207); so does
                    // running out of depth: the next getRayColor would return black (src/Camera.cpp:138)
                    if (__float_as_uint(at.w) == 3u || pt.depth == 0) finished = true;
                    else bounce = true;
                    // diffuse and metal bounces start exactly on the sphere; the dielectric nudges its origin
                    next_self = __float_as_uint(at.w) <= 1u ? hit_self : -1;
                    if (FULL && KP.n_lights > 0) {
                        // Camera::computeDirectLighting (src/Camera.cpp:161-178): the hit is kept, the shadow
                        // queries run in the following iterations, then the path resumes or ends
                        ls.hp = h.p;
                        ls.hn = h.n;
                        ls.hobj = hit_obj;
                        ls.hself = hit_self;
                        ls.hmat = hit_mat;
                        ls.nxt_valid = bounce;
                        ls.light_i = 0;
                        finished = false;
                    }
                }
            } else {
                // the closest hit along the ray towards the light; a hit on the shaded object itself does
                // not count, and cubes / polygons all carry the same null object (src/Camera.cpp:169-172)
                int obj = OBJ_NONE;
                if (any_hit) {
                    const uint32_t ty = full.prims[c.prim].type;
                    obj = (ty == PRIM_CUBE || ty == PRIM_POLYGON) ? OBJ_NULL : c.prim;
                }
                const bool hitted = any_hit && obj != ls.hobj;
                float4 idf = KS.mats[ls.hmat].ior_diffuse;
                pt.L = pt.L + pt.T * light_direct(KS.lights[ls.light_i], cam_center, ls.hp, ls.hn,
                                                  mk(idf.y, idf.z, idf.w), hitted);
                ls.light_i++;
                if (ls.light_i >= KP.n_lights) {
                    ls.light_i = -1;
                    if (ls.nxt_valid) {
                        pt.T = pt.T * ls.att;
                        pt.o = ls.nxt_o;
                        set_dir(pt, ls.nxt_d);
                        pt.self_id = ls.nxt_self;
                    } else {
                        finished = true;
                    }
                }
            }
        }
        RT_SEC(2); // radiance of finished paths into the tile's sums
        if (finished) {
            unsigned long long fx, fy, fz;
            to_fixed3(pt.L, fx, fy, fz);
            atomicAdd(&acc[pt.pix], fx);
            atomicAdd(&acc[MAX_TILE_PX + pt.pix], fy);
            atomicAdd(&acc[2 * MAX_TILE_PX + pt.pix], fz);
            pt.pix = -1;
        }

        // ---- a slot whose pool has run dry and whose last path has ended: tone-map and store
        //      (convertRTColor, src/Camera.cpp:19-34) ----
        RT_SEC(3); // tiles that are done: tone-map and store
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool occupied = s ? have1 : have0;
            if (!occupied || (pool_open && fill == s)) continue;
            if (__ballot(pt.pix >= 0 && (pt.pix >= SLOT_WORDS) == (s == 1))) continue;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const unsigned long long *a = acc + s * SLOT_WORDS;
            const int *d = desc + 8 * s;
            const int x0 = d[0], vw = d[2], vh = d[3], row0 = d[4];
            for (int q = lane; q < tile_px; q += 64) {
                int lx = q & ((1 << KP.tile_w_log2) - 1), ly = q >> KP.tile_w_log2;
                if (lx < vw && ly < vh) {
                    size_t o = (size_t)(row0 + ly) * (size_t)KP.width + (size_t)(x0 + lx);
                    float r = from_fixed(a[q]), g = from_fixed(a[MAX_TILE_PX + q]), b = from_fixed(a[2 * MAX_TILE_PX + q]);
                    float cnt = (float)KP.spp;
                    if (accum) {
                        if (KP.flags & RT_FLAG_ACCUMULATE) {
                            float4 prev = accum[o];
                            r += prev.x; g += prev.y; b += prev.z; cnt += prev.w;
                        }
                        accum[o] = make_float4(r, g, b, cnt);
                    }
                    float inv = 1.0f / cnt;
                    uchar4 px8;
                    px8.x = (unsigned char)to_byte(r * inv);
                    px8.y = (unsigned char)to_byte(g * inv);
                    px8.z = (unsigned char)to_byte(b * inv);
                    px8.w = 255;
                    rgba[o] = px8;
                }
            }
            if (s) have1 = false; else have0 = false;
#ifdef RT_STATS
            items_done++;
            if (lane == 0) atomicMax(&g_times[6], ((wall_clock64() - t_fetch[s]) << 32) | cur_item[s]);
#endif
        }

        // ---- hand the next (pixel, sample) of the open pool to every lane without a path
        //      (ballot + prefix count); when the pool runs dry, open the next tile's in the free
        //      slot and go on, so the wave stays full until the queue is empty ----
        RT_SEC(4); // hand-out of new samples
        bool fresh = false;
        int px = 0, py = 0, new_smp = 0;
        for (unsigned long long m = __ballot(pt.pix < 0); m != 0; m = __ballot(pt.pix < 0)) {
            if (!pool_open) {
                if (queue_empty || (have0 && have1)) break;
                unsigned it = 0;
                if (lane == 0) it = atomicAdd(queue, 1u);
                it = (unsigned)__builtin_amdgcn_readfirstlane((int)it);
                if (it >= (unsigned)KP.n_items) {
                    queue_empty = true;
                    break;
                }
                tile = decode_item(KP, it);
                nvalid = (tile.vw > 0 && tile.vh > 0) ? tile.vw * tile.vh : 0;
                if (nvalid == 0) continue; // a tile outside the image: nothing to render or store
                fill = have0 ? 1 : 0;
                if (fill) have1 = true; else have0 = true;
                pool_open = true;
                inv_nvalid = 1.0f / (float)nvalid;
                inv_vw = 1.0f / (float)tile.vw;
                cur_q = 0;
                cur_smp = 0;
                unsigned long long *a = acc + fill * SLOT_WORDS;
                for (int q = lane; q < tile_px; q += 64) {
                    a[q] = 0ull;
                    a[MAX_TILE_PX + q] = 0ull;
                    a[2 * MAX_TILE_PX + q] = 0ull;
                }
                if (lane == 0) {
                    int *d = desc + 8 * fill;
                    d[0] = tile.x0; d[1] = tile.y0; d[2] = tile.vw; d[3] = tile.vh; d[4] = tile.row0;
                }
                if (PRIM && KP.prim_min > 0) {
                    const int n_listed = list_primary_candidates(KP, tile, KS.sph, KP.n_spheres, plist + PRIM_LIST_CAP * fill, lane);
                    if (lane == 0) desc[8 * fill + 5] = n_listed;
                }
#ifdef RT_STATS
                t_fetch[fill] = wall_clock64();
                cur_item[fill] = it;
#endif
            }
            const int asking = (int)__popcll(m);
            if (pt.pix < 0) {
                // idx < nvalid + 64 <= 128 and q < 64: the float reciprocals divide exactly
                int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                int idx = cur_q + rank;
                int wraps = (int)(((float)idx + 0.5f) * inv_nvalid);
                int q = idx - wraps * nvalid;
                int smp = cur_smp + wraps;
                if (smp < KP.spp) {
                    int ly = (int)(((float)q + 0.5f) * inv_vw);
                    int lx = q - ly * tile.vw;
                    pt.pix = fill * SLOT_WORDS + (ly << KP.tile_w_log2) + lx;
                    px = tile.x0 + lx;
                    py = tile.y0 + ly;
                    new_smp = smp;
                    fresh = true;
                }
            }
            {
                int idx = cur_q + asking;
                int wraps = (int)(((float)idx + 0.5f) * inv_nvalid);
                cur_q = idx - wraps * nvalid;
                cur_smp += wraps;
                if (cur_smp >= KP.spp) pool_open = false;
            }
        }

        // ---- the next ray of every lane that has one to make: a primary ray (Camera::genRay,
        //      src/Camera.cpp:124-135) or a bounce (RTMaterial::scatter); one random draw serves both ----
        RT_SEC(5); // random words, primary ray or scatter
        if (fresh) {
            // the path state of a new sample is set here, outside the hand-out loop: inside it every field
            // would be carried around the loop's exits in register copies
            pt.gpix = rng_key((uint32_t)(py * KP.width + px));
            pt.sample = rng_key((uint32_t)(new_smp + KP.sample_offset));
            pt.call = rng_key(0u);
            pt.T = mk(1, 1, 1);
            pt.L = mk(0, 0, 0);
            pt.depth = KP.max_depth;
            pt.self_id = -1;
        }
        if (fresh || bounce) {
            uint32_t w0, w1, w2, w3;
            rng4_keyed(pt.gpix, pt.sample, pt.call, rng_key(KP.seed), w0, w1, w2, w3);
            pt.call += RNG_MUL;
            V3 nd;
            if (fresh) {
                gen_ray(KP, px, py, u01(w0), u01(w1), pt.o, nd);
                if (FULL) ls.light_i = -1;
                set_dir(pt, nd);
                if (PRIM && KP.prim_min > 0 && pt.depth > 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the list was written by other lanes
                    const int slot = pt.pix >= SLOT_WORDS ? 1 : 0;
                    const int n_listed = desc[8 * slot + 5];
                    if (n_listed >= 0) {
                        const int *mine = plist + PRIM_LIST_CAP * slot;
                        const float tmin0 = TMIN * pt.len;
                        unsigned long long key = NO_HIT_KEY;
                        for (int k = 0; k < n_listed; ++k) {
                            const int sidx = mine[k];
                            const float4 sp = KS.sph[sidx];
                            const float cand = sphere_root(pt.o, pt.dh, sp.x, sp.y, sp.z, sp.w * sp.w, tmin0);
                            if (cand > tmin0 && cand < __builtin_inff()) {
                                const unsigned long long kk = ((unsigned long long)__float_as_uint(cand) << 32) | (unsigned long long)(unsigned)sidx;
                                key = kk < key ? kk : key;
                            }
                        }
                        pre_t = key != NO_HIT_KEY ? __uint_as_float((unsigned)(key >> 32)) : __builtin_inff();
                        pre_id = key != NO_HIT_KEY ? (int)(unsigned)key : -1;
                    }
                }
            } else {
                V3 att, no;
                scatter(__float_as_uint(at.w), mk(at.x, at.y, at.z), ef.w, ior, pt.d, h, u01(w0), u01(w1), u01(w2), att,
                        no, nd);
                const int nself = next_self;
                if (FULL && KP.n_lights > 0) {
                    ls.nxt_o = no;
                    ls.nxt_d = nd;
                    ls.att = att;
                    ls.nxt_self = nself;
                } else {
                    pt.T = pt.T * att;
                    pt.o = no;
                    pt.self_id = nself;
                    set_dir(pt, nd);
                }
            }
        }
        if (FULL && ls.light_i >= 0 && pt.pix >= 0) {
            // (next) shadow ray: from the hit point towards the light, direction unnormalised as in
            // the reference (src/Camera.cpp:166)
            float4 lp = KS.lights[ls.light_i].pos_pow;
            pt.o = ls.hp;
            set_dir(pt, mk(lp.x, lp.y, lp.z) - ls.hp);
            pt.self_id = ls.hself;
        }

    }
#ifdef RT_SECTIONS
    RT_SEC(7);
    if (lane == 0 && SRC != SRC_SCALAR)
        for (int k = 0; k < 16; ++k) atomicAdd(&g_sections[k], sec.t[k]);
#endif
#ifdef RT_STATS
    if (lane == 0 && items_done > 0) {
        const unsigned long long t_end = wall_clock64();
        atomicMin(&g_times[0], t_start);
        atomicMin(&g_times[1], t_end);
        atomicMax(&g_times[2], t_end);
        atomicAdd(&g_times[3], t_end - t_start);
        atomicAdd(&g_times[4], 1ull);
        atomicMax(&g_times[5], (unsigned long long)items_done);
    }
#endif

    if (rays_out) {
        // wave-level sum, one atomic per wave
        for (int off = 32; off > 0; off >>= 1) my_rays += __shfl_down(my_rays, off, 64);
        if (lane == 0 && my_rays) atomicAdd(rays_out, my_rays);
    }
    if (scans_out && lane == 0 && n_scans) atomicAdd(scans_out, (unsigned long long)n_scans);
#undef KP
#undef KS
#undef rgba
#undef accum
#undef rays_out
#undef scans_out
#undef queue
}

// Filter values of a batch of rays against every table row, through the device functions of the render
// kernel's scan (test hook).  One wave per 64 rays; the table words come from global memory here.
__global__ __launch_bounds__(64) void filter_probe_kernel(int n_rays, const float *__restrict__ org,
                                                          const float *__restrict__ dir, const uint4v *__restrict__ tab,
                                                          const int *__restrict__ row_idx, int n_cols, MfmaFrame fr,
                                                          float *__restrict__ out) {
    const int lane = threadIdx.x, base = blockIdx.x * 64, i = base + lane;
    const bool live = i < n_rays;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (live) {
        o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    }
    const V3 dh = d * __frsqrt_rn(dot(d, d)); // as set_dir() does
    const RayOperand op = mfma_ray_operand(o, dh, live, fr.cx, fr.cy, fr.cz, fr.scale);
    for (int blk = 0; blk < fr.n_blocks; ++blk) {
        const half8 a0 = __builtin_bit_cast(half8, tab[(2 * blk) * 64 + lane]);
        const half8 a1 = __builtin_bit_cast(half8, tab[(2 * blk + 1) * 64 + lane]);
        const float16v acc0 = mfma_tile(a0, a1, op.b[0][0], op.b[0][1]);
        const float16v acc1 = mfma_tile(a0, a1, op.b[1][0], op.b[1][1]);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = MF_BLOCK * blk + v + 16 * (lane >> 5); // the table row behind register v (rt_mfma.h)
            const int col = row_idx[row];                          // the sphere it holds; -1: padding
            const int r0 = base + (lane & 31), r1 = r0 + 32;
            if (col >= 0 && r0 < n_rays) out[(size_t)r0 * n_cols + col] = acc0[v] * (1.0f / (MF_SCALE * MF_SCALE));
            if (col >= 0 && r1 < n_rays) out[(size_t)r1 * n_cols + col] = acc1[v] * (1.0f / (MF_SCALE * MF_SCALE));
        }
    }
}

// ---------------------------------------------------------------------------
// First-hit kernel: SceneManager::hitClosest for a batch of rays
// (src/RayTracer.cpp:68-82 with SphereObject::hit, inc/RTObjects.h:129-159).
// ---------------------------------------------------------------------------

// Double-precision restatement of hitDetail's decision for one sphere, evaluated
// with separately rounded operations (no contraction) so that it reproduces a
// host double evaluation bit for bit.
#pragma clang fp contract(off)
__device__ __noinline__ bool hit64(double ox, double oy, double oz, double dx, double dy, double dz, double tmin,
                                   double tmax, double cx, double cy, double cz, double r, double &t_out) {
    double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
    double a = dx * dx + dy * dy + dz * dz;
    double hb = ocx * dx + ocy * dy + ocz * dz;
    double c = (ocx * ocx + ocy * ocy + ocz * ocz) - r * r;
    double disc = hb * hb - a * c;
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double root = (-hb - sq) / a;
    if (!(tmin < root && root < tmax)) {
        root = (-hb + sq) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    t_out = root;
    return true;
}

__device__ __noinline__ void finish64(double ox, double oy, double oz, double dx, double dy, double dz, double t,
                                      double cx, double cy, double cz, double r, float *n_out, uint8_t *front) {
    double px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;
    double wx = (px - cx) / r, wy = (py - cy) / r, wz = (pz - cz) / r;
    bool f = (dx * wx + dy * wy + dz * wz) < 0.0;
    double len = sqrt(wx * wx + wy * wy + wz * wz);
    double nx = wx / len, ny = wy / len, nz = wz / len;
    if (!f) { nx = -nx; ny = -ny; nz = -nz; }
    if (n_out) { n_out[0] = (float)nx; n_out[1] = (float)ny; n_out[2] = (float)nz; }
    if (front) *front = f ? 1 : 0;
}

// Double-precision closest hit over a general primitive list: the same decisions as
// SceneManager::hitClosest with Primitives::hit / hitExpanded of every kind
// (src/RayTracer.cpp:68-100; inc/RTObjects.h:129-159 sphere, :174-190 plane, :469-512 polygon,
// :586-644 cube), evaluated operation by operation as a host double evaluation would.
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 dsub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 dat(D3 o, D3 d, double t) { return d3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t); }

struct Rec64 {
    double t;
    int prim;
    D3 n;
    bool front, expanded;
};

// HitRecord::setFaceNormal, inc/RTGeometry.h:58-61
__device__ __forceinline__ void face64(Rec64 &rec, D3 d, D3 outward) {
    rec.front = ddot(d, outward) < 0.0;
    double len = sqrt(ddot(outward, outward));
    D3 nn = d3(outward.x / len, outward.y / len, outward.z / len);
    rec.n = rec.front ? nn : d3(-nn.x, -nn.y, -nn.z);
}

__device__ __noinline__ bool prim_hit64(const FullScene &fs, int idx, D3 o, D3 d, double tmin, double tmax, double scale,
                                        Rec64 &rec) {
    const DPrim pr = load_prim(fs.prims, idx);
    if (pr.type == PRIM_SPHERE) {
        double r = (double)pr.p[3] * scale, t;
        if (!hit64(o.x, o.y, o.z, d.x, d.y, d.z, tmin, tmax, pr.p[0], pr.p[1], pr.p[2], r, t)) return false;
        D3 pt = dat(o, d, t);
        D3 w = d3((pt.x - pr.p[0]) / r, (pt.y - pr.p[1]) / r, (pt.z - pr.p[2]) / r);
        rec.t = t;
        face64(rec, d, w);
    } else if (pr.type == PRIM_PLANE) {
        D3 pos = d3(pr.p[0], pr.p[1], pr.p[2]), n = d3(pr.p[3], pr.p[4], pr.p[5]);
        double t = ddot(n, dsub(pos, o)) * (1.0 / ddot(n, d));
        if (!(tmin < t && t < tmax)) return false;
        rec.t = t;
        face64(rec, d, ddot(n, d) < 0.0 ? d3(-n.x, -n.y, -n.z) : n);
    } else if (pr.type == PRIM_CUBE) {
        double lo_t = -__builtin_inf(), hi_t = __builtin_inf();
        const double orig[3] = {o.x, o.y, o.z}, dir[3] = {d.x, d.y, d.z};
        double lo[3], hi[3];
        for (int i = 0; i < 3; ++i) {
            double hs = (double)pr.p[3 + i] * scale;
            lo[i] = (double)pr.p[i] - hs;
            hi[i] = (double)pr.p[i] + hs;
        }
        for (int i = 0; i < 3; ++i) {
            if (fabs(dir[i]) < 1e-12) {
                if (orig[i] < lo[i] || orig[i] > hi[i]) return false;
            } else {
                double inv = 1.0 / dir[i];
                double t0 = (lo[i] - orig[i]) * inv, t1 = (hi[i] - orig[i]) * inv;
                if (t0 > t1) { double tt = t0; t0 = t1; t1 = tt; }
                lo_t = lo_t > t0 ? lo_t : t0;
                hi_t = hi_t < t1 ? hi_t : t1;
                if (hi_t < lo_t) return false;
            }
        }
        double t = lo_t;
        if (!(tmin < t && t < tmax)) {
            t = hi_t;
            if (!(tmin < t && t < tmax)) return false;
        }
        D3 pt = dat(o, d, t), out = d3(0, 0, 0);
        const double EPS = 1e-6;
        if (fabs(pt.x - lo[0]) < EPS) out = d3(-1, 0, 0);
        else if (fabs(pt.x - hi[0]) < EPS) out = d3(1, 0, 0);
        else if (fabs(pt.y - lo[1]) < EPS) out = d3(0, -1, 0);
        else if (fabs(pt.y - hi[1]) < EPS) out = d3(0, 1, 0);
        else if (fabs(pt.z - lo[2]) < EPS) out = d3(0, 0, -1);
        else if (fabs(pt.z - hi[2]) < EPS) out = d3(0, 0, 1);
        else {
            D3 l = d3(pt.x - pr.p[0], pt.y - pr.p[1], pt.z - pr.p[2]);
            double ax = fabs(l.x), ay = fabs(l.y), az = fabs(l.z);
            if (ax > ay && ax > az) out = d3(l.x > 0 ? 1 : -1, 0, 0);
            else if (ay > az) out = d3(0, l.y > 0 ? 1 : -1, 0);
            else out = d3(0, 0, l.z > 0 ? 1 : -1);
        }
        rec.t = t;
        face64(rec, d, out);
    } else {
        const int nv = (int)pr.n_vertices;
        if (nv < 3) return false;
        D3 n = d3(pr.p[0], pr.p[1], pr.p[2]), cen = d3(pr.p[3], pr.p[4], pr.p[5]);
        double denom = ddot(n, d);
        if (fabs(denom) < 1e-12) return false;
        double t = ddot(n, dsub(cen, o)) / denom;
        if (!(tmin < t && t < tmax)) return false;
        D3 pt = dat(o, d, t);
        // PolygonObject::projectTo2D: drop the dominant axis of the normal
        double ax = fabs(n.x), ay = fabs(n.y), az = fabs(n.z);
        const int drop = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
        auto uv = [&](D3 q, double &u, double &v) {
            if (drop == 0) { u = q.y; v = q.z; }
            else if (drop == 1) { u = q.x; v = q.z; }
            else { u = q.x; v = q.y; }
        };
        auto vertex = [&](int k) {
            cptr<float> vv = fs.verts + 3 * ((size_t)pr.first_vertex + (size_t)k);
            D3 v = d3(vv[0], vv[1], vv[2]);
            if (scale != 1.0) {
                D3 sd = d3((v.x - cen.x) * scale, (v.y - cen.y) * scale, (v.z - cen.z) * scale);
                v = d3(cen.x + sd.x, cen.y + sd.y, cen.z + sd.z);
            }
            return v;
        };
        double px, py;
        uv(pt, px, py);
        bool inside = false;
        for (int i = 0, j = nv - 1; i < nv; j = i++) {
            double xi, yi, xj, yj;
            uv(vertex(i), xi, yi);
            uv(vertex(j), xj, yj);
            bool cross = ((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi + 1e-20) + xi);
            if (cross) inside = !inside;
        }
        if (!inside) return false;
        rec.t = t;
        face64(rec, d, ddot(n, d) > 0.0 ? d3(-n.x, -n.y, -n.z) : n);
    }
    rec.prim = idx;
    return true;
}

__device__ __noinline__ Rec64 closest64_general(const FullScene &fs, int n_prims, D3 o, D3 d, double tmin,
                                                bool expanded_state) {
    Rec64 rec, ex;
    rec.t = 0.0;
    rec.prim = -1;
    rec.n = d3(0, 0, 0);
    rec.front = rec.expanded = false;
    ex = rec;
    double closest = __builtin_inf(), closest_ex = __builtin_inf();
    for (int i = 0; i < n_prims; ++i)
        if (prim_hit64(fs, i, o, d, tmin, closest, 1.0, rec)) closest = rec.t;
    if (expanded_state) {
        for (int k = 0; k < fs.n_halo; ++k) {
            const int i = fs.halo[k];
            const uint32_t ty = fs.prims[i].type;
            bool ok = prim_hit64(fs, i, o, d, tmin, closest_ex, 1.0, ex);
            if (!ok && ty != PRIM_PLANE) {
                ok = prim_hit64(fs, i, o, d, tmin, closest_ex, (double)1.05, ex);
                if (ok) ex.expanded = true; // never cleared by a later plain hit, as in the reference
            }
            if (ok) closest_ex = ex.t;
        }
    }
    return closest_ex < closest ? ex : rec;
}
#pragma clang fp contract(fast)

template <bool REFINE>
__global__ __launch_bounds__(WG_THREADS) void first_hit_kernel(int n_rays, const float *__restrict__ org,
                                                               const float *__restrict__ dir, float tmin,
                                                               DeviceScene sc, int n_spheres, int n_prims, int general,
                                                               int expanded_state, int32_t *__restrict__ id,
                                                               float *__restrict__ t, float *__restrict__ normal,
                                                               uint8_t *__restrict__ front,
                                                               uint8_t *__restrict__ expanded) {
    // general: the scene holds other primitives than spheres, or the halo pass is wanted; ids are then
    // indices into the primitive list
    if (REFINE && general) {
        const int i = blockIdx.x * WG_THREADS + threadIdx.x;
        if (i >= n_rays) return;
        Rec64 r = closest64_general(sc.full, n_prims, d3(org[3 * i], org[3 * i + 1], org[3 * i + 2]),
                                    d3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), (double)tmin, expanded_state != 0);
        id[i] = r.prim;
        const bool hit = r.prim >= 0;
        if (t) t[i] = hit ? (float)r.t : 0.f;
        if (normal) {
            normal[3 * i] = hit ? (float)r.n.x : 0.f;
            normal[3 * i + 1] = hit ? (float)r.n.y : 0.f;
            normal[3 * i + 2] = hit ? (float)r.n.z : 0.f;
        }
        if (front) front[i] = hit && r.front ? 1 : 0;
        if (expanded) expanded[i] = hit && r.expanded ? 1 : 0;
        return;
    }
    __shared__ float4 lds_sph[5 * STREAM_QUADS];
    float4 *lds_r2 = lds_sph + 4 * STREAM_QUADS;
    const int i = blockIdx.x * WG_THREADS + threadIdx.x;
    const bool live = i < n_rays;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (live) {
        o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    }
    const float a = dot(d, d);
    const float inv_len = __frsqrt_rn(a), len = a * inv_len, inv_a = 1.0f / a;
    const V3 dh = d * inv_len;
    const RayK rk = make_rayk(o, dh);
    float best_t = __builtin_inff(); // in units of |d| during the scan
    int best_id = -1;
    const int nquads = ((n_spheres + 7) / 8) * 2; // as padded by rt_upload_scene
    for (int base = 0; base < nquads; base += STREAM_QUADS) {
        int cnt = min(STREAM_QUADS, nquads - base);
        __syncthreads();
        for (int k = threadIdx.x; k < 4 * cnt; k += WG_THREADS) lds_sph[k] = sc.pairs[4 * base + k];
        for (int k = threadIdx.x; k < cnt; k += WG_THREADS) lds_r2[k] = sc.r2q[base + k];
        __syncthreads();
        if (live) scan_lds(lds_sph, lds_r2, cnt, 4 * base, rk, o, dh, tmin * len, -1, best_t, best_id);
    }
    best_t *= inv_len;
    if (live && expanded) expanded[i] = 0;
    if (!REFINE && general) {
        if (!live) return;
        Closest c;
        c.t = best_id >= 0 ? best_t : __builtin_inff();
        c.prim = best_id >= 0 ? sc.full.sph_gid[best_id] : -1;
        c.scale = 1.0f;
        c.box_face = 0;
        c.expanded = false;
        for (int k = 0; k < sc.full.n_others; ++k) other_hit(sc.full, sc.full.others[k], o, d, tmin, c);
        if (expanded_state && sc.full.n_halo > 0) {
            Closest e = halo_pass(sc.full, o, d, tmin);
            if (e.expanded && e.t < c.t) c = e; // a plain halo-pass hit is the normal pass's own hit
        }
        id[i] = c.prim;
        Hit h;
        h.p = h.n = mk(0, 0, 0);
        h.front = false;
        int object = OBJ_NONE;
        if (c.prim >= 0) h = finish_any(sc.full, o, d, c, object);
        if (t) t[i] = c.prim >= 0 ? c.t : 0.f;
        if (normal) { normal[3 * i] = h.n.x; normal[3 * i + 1] = h.n.y; normal[3 * i + 2] = h.n.z; }
        if (front) front[i] = h.front ? 1 : 0;
        if (expanded) expanded[i] = c.prim >= 0 && c.expanded ? 1 : 0;
        return;
    }
    if (!REFINE) {
        if (!live) return;
        id[i] = best_id;
        if (best_id >= 0) {
            Hit h = finish_hit(o, d, best_t, sc.sph[best_id]);
            if (t) t[i] = best_t;
            if (normal) { normal[3 * i] = h.n.x; normal[3 * i + 1] = h.n.y; normal[3 * i + 2] = h.n.z; }
            if (front) front[i] = h.front ? 1 : 0;
        } else {
            if (t) t[i] = 0.f;
            if (normal) { normal[3 * i] = 0.f; normal[3 * i + 1] = 0.f; normal[3 * i + 2] = 0.f; }
            if (front) front[i] = 0;
        }
        return;
    }
    // Refinement: every sphere whose single-precision evaluation cannot rule it out
    // (discriminant not clearly negative, nearer root not clearly beyond the
    // single-precision winner) is re-tested in double precision, in index order with
    // the shrinking interval of hitClosest.  If that finds no hit inside the window
    // (the single-precision winner was a false positive) the window is dropped and
    // every sphere with a non-negative-looking discriminant is re-tested.
    const float EPS = 2e-3f;
    float t_lim = best_id >= 0 ? best_t * (1.0f + EPS) + 1e-4f : __builtin_inff();
    double closest = __builtin_inf();
    int id64 = -1;
    if (live) {
        for (int pass = 0; pass < 2; ++pass) {
            closest = __builtin_inf();
            id64 = -1;
            for (int k = 0; k < n_spheres; ++k) {
                float4 s = sc.sph[k];
                float ocx = o.x - s.x, ocy = o.y - s.y, ocz = o.z - s.z;
                float hb = fmaf(ocx, d.x, fmaf(ocy, d.y, ocz * d.z));
                float r2 = s.w * s.w;
                float oc2 = fmaf(ocx, ocx, fmaf(ocy, ocy, ocz * ocz));
                float c = oc2 - r2;
                float disc = fmaf(hb, hb, -(a * c));
                float scale = fmaf(hb, hb, a * (oc2 + r2));
                if (disc < -EPS * scale) continue;
                float sq = __fsqrt_rn(fmaxf(disc, 0.f) + EPS * scale);
                float near_root = (-hb - sq) * inv_a; // a lower bound of the nearer root
                if (near_root > t_lim) continue;
                double tt;
                if (hit64(o.x, o.y, o.z, d.x, d.y, d.z, (double)tmin, closest, s.x, s.y, s.z, s.w, tt)) {
                    closest = tt;
                    id64 = k;
                }
            }
            if (closest <= (double)t_lim) break; // a verified hit inside the window: nothing outside can beat it
            if (!(t_lim < __builtin_inff())) break; // the window was already unbounded
            t_lim = __builtin_inff();
        }
        id[i] = id64;
        if (id64 >= 0) {
            float4 s = sc.sph[id64];
            if (t) t[i] = (float)closest;
            finish64(o.x, o.y, o.z, d.x, d.y, d.z, closest, s.x, s.y, s.z, s.w, normal ? normal + 3 * i : nullptr,
                     front ? front + i : nullptr);
        } else {
            if (t) t[i] = 0.f;
            if (normal) { normal[3 * i] = 0.f; normal[3 * i + 1] = 0.f; normal[3 * i + 2] = 0.f; }
            if (front) front[i] = 0;
        }
    }
}

// First hits of a sphere-only scene through the scan the render kernel runs on it: matrix-core filter over the
// scene's table (read from global memory here), candidate queue, exact decisions (rt_mfma.h).  One ray per
// lane; every wave runs whole, lanes past the last ray included.
__global__ __launch_bounds__(WG_THREADS) void first_hit_mfma_kernel(int n_rays, const float *__restrict__ org,
                                                                    const float *__restrict__ dir, float tmin,
                                                                    DeviceScene sc, MfmaFrame fr, int32_t *__restrict__ id,
                                                                    float *__restrict__ t, float *__restrict__ normal,
                                                                    uint8_t *__restrict__ front,
                                                                    uint8_t *__restrict__ expanded) {
    __shared__ unsigned long long slots[WAVES_PER_WG * 64];
    __shared__ unsigned ents[WAVES_PER_WG * cand_cap_of(SRC_MFMA)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * WG_THREADS + threadIdx.x;
    const bool live = i < n_rays;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (live) {
        o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    }
    const float a = dot(d, d);
    const float inv_len = __frsqrt_rn(a), len = a * inv_len; // as set_dir() does
    const V3 dh = d * inv_len;
    CandQ cq;
    cq.slot = slots + wave * 64;
    cq.ent = ents + wave * cand_cap_of(SRC_MFMA);
    cq.row_idx = sc.mf_row_idx;
    float best_t = __builtin_inff(); // in units of |d| during the scan
    int best_id = -1;
    SecClock sec;
    sec.start();
    scan_mfma(sc.mf_table, fr, o, dh, live, tmin * len, -1, lane, cq, sc.sph, sc.mf_rows, best_t, best_id, sec);
    if (!live) return;
    best_t *= inv_len;
    if (expanded) expanded[i] = 0;
    id[i] = best_id;
    if (best_id >= 0) {
        Hit h = finish_hit(o, d, best_t, sc.sph[best_id]);
        if (t) t[i] = best_t;
        if (normal) { normal[3 * i] = h.n.x; normal[3 * i + 1] = h.n.y; normal[3 * i + 2] = h.n.z; }
        if (front) front[i] = h.front ? 1 : 0;
    } else {
        if (t) t[i] = 0.f;
        if (normal) { normal[3 * i] = 0.f; normal[3 * i + 1] = 0.f; normal[3 * i + 2] = 0.f; }
        if (front) front[i] = 0;
    }
}

// Batch scatter: the bounce step of the render kernel on explicit inputs.
__global__ void scatter_kernel(int n, const DMat *__restrict__ mats, const uint32_t *__restrict__ mat,
                               const float *__restrict__ in_dir, const float *__restrict__ point,
                               const float *__restrict__ nrm, const uint8_t *__restrict__ front,
                               const float *__restrict__ u, float *__restrict__ att, float *__restrict__ out_org,
                               float *__restrict__ out_dir, uint8_t *__restrict__ flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DMat &m = mats[mat[i]];
    float4 at = m.albedo_type, ef = m.emission_fuzz;
    Hit h;
    h.p = mk(point[3 * i], point[3 * i + 1], point[3 * i + 2]);
    h.n = mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    h.front = front[i] != 0;
    V3 a = mk(0, 0, 0), o = mk(0, 0, 0), d = mk(0, 0, 0);
    bool ok = scatter(__float_as_uint(at.w), mk(at.x, at.y, at.z), ef.w, m.ior_diffuse.x,
                      mk(in_dir[3 * i], in_dir[3 * i + 1], in_dir[3 * i + 2]), h, u[3 * i], u[3 * i + 1], u[3 * i + 2],
                      a, o, d);
    flag[i] = ok ? 1 : 0;
    if (!ok) { a = mk(0, 0, 0); o = mk(0, 0, 0); d = mk(0, 0, 0); }
    att[3 * i] = a.x; att[3 * i + 1] = a.y; att[3 * i + 2] = a.z;
    out_org[3 * i] = o.x; out_org[3 * i + 1] = o.y; out_org[3 * i + 2] = o.z;
    out_dir[3 * i] = d.x; out_dir[3 * i + 1] = d.y; out_dir[3 * i + 2] = d.z;
}

// The generator as the render kernel evaluates it, on explicit keys (pixel, sample, call, seed).
__global__ void rng_kernel(int n, const uint32_t *__restrict__ keys, uint32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w0, w1, w2, w3;
    rng4(keys[4 * i], keys[4 * i + 1], keys[4 * i + 2], keys[4 * i + 3], w0, w1, w2, w3);
    out[4 * i] = w0; out[4 * i + 1] = w1; out[4 * i + 2] = w2; out[4 * i + 3] = w3;
}

} // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

// The persistent render kernel pulls its work items from a 32-bit counter in device memory.  A launch
// takes the next of QUEUE_SLOTS counters (256 bytes apart), so renders issued back to back on
// different streams do not share one.
constexpr int QUEUE_SLOTS = 16;
constexpr size_t QUEUE_SLOT_BYTES = 256;

// Counters of rt_render_noalloc: a device array of the code object instead of a hipMalloc'ed buffer.
// Every GPU holds its own copy at its own address; `scratch` itself is only a host-side handle, the
// device address comes from hipGetSymbolAddress (scratch_ptr below) and is what kernels and
// hipMemsetAsync receive.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static_assert(RT_NOALLOC_SCRATCH_BYTES == QUEUE_SLOTS * QUEUE_SLOT_BYTES, "scratch holds one ring of work counters");

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[MAX_DEVICES]; // per device: address of its copy of `scratch`, null until looked up
std::atomic<unsigned> g_scratch_next[MAX_DEVICES];

// Address of `scratch` on `device`, looked up once per device with that device current.
hipError_t scratch_ptr(int device, unsigned char **out) {
    *out = nullptr;
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
} // namespace

struct rt_ctx {
    int device = 0;
    std::string err;
    float4 *d_sph = nullptr;
    float4 *d_pairs = nullptr;
    float4 *d_r2q = nullptr;
    int n_quads = 0;
    bool scalar_source = true;
    // matrix-core filter table (rt_mfma.h): built with every scene it can serve
    uint4v *d_mf_table = nullptr;
    float4 *d_mf_rows = nullptr;
    int *d_mf_row_idx = nullptr;
    int mf_cols = 0; // columns of rt_filter_values_host: one per sphere index, padded
    MfmaFrame mf{};
    bool mf_usable = false;  // the scene has a table and it fits in LDS
    bool mf_wanted = true;   // RT_AMD_SPHERE_SOURCE did not ask for another source
    bool mf_full_ok = true;  // the device accepted the LDS budget of the general kernel's matrix-core variant
    int mf_resident = 0;     // workgroups of the scene's matrix-core variant a CU holds (queried at upload)
    int mf_full_min = 24; // sphere count from which the general kernel takes the table (measured: even at 16, 5 % ahead at 32, 26 % at 64; tuning: RT_AMD_MF_FULL_MIN)
    int mf_resident_full = 0; // the same for the general kernel's variant (0: its table does not fit, or no such device)
    bool mf_full_streamed_ok = true;
    int prim_min = 64; // primary-ray lists of the sphere-only matrix-core kernel: lanes that must need a scan for one to run (0: off; tuning: RT_AMD_PRIM_MIN)
    int prim_min_full = 48; // the general kernel's paths are longer, fewer lanes start a sample per turn: it scans sooner (485 spheres on a floor plane 8.23 / 7.58 / 7.14 / 7.35 ms at 0 / 32 / 48 / 64; with two lights 15.1 / 15.0 / 15.0 / 15.9)
    int prim_min_streamed = 64; // the same for the variants that stream their table (their workgroups decide as one; lit cloud of 5000 spheres through the general kernel: 42.0 / 39.7 / 38.4 ms at 0 / 48 / 64)
    int last_variant = -1; // sphere source of the last render launch, + 8 for the general kernel
    int mf_streamed_full = 0; // the same for the general kernel's variant with the table streamed through LDS
    unsigned char *d_queue = nullptr; // ring of work-item counters of the persistent render kernel
    unsigned queue_next = 0;          // next slot of the ring
    unsigned char *d_scratch = nullptr; // this device's copy of `scratch` (rt_render_noalloc)
    int n_cus = 256;
    int wgs_per_cu = 4; // workgroups of four waves per CU in the persistent grid (measured: 4 / 5 / 8 give 4.18 / 4.22 / 4.27 ms on the four-sphere scene, no difference elsewhere)
    int force_tile = -1; // tuning: 0 = 1x1, 1 = 2x2, 2 = 4x4, 3 = 8x8 work-item tiles
    uint32_t *d_mat_id = nullptr;
    DMat *d_mats = nullptr;
    DMat *d_sph_mats = nullptr;
    DLight *d_lights = nullptr;
    int n_lights = 0;
    // the general primitive list (every scene has one; sphere-only scenes only use it with lights)
    DPrim *d_prims = nullptr;
    float *d_verts = nullptr;
    int32_t *d_others = nullptr, *d_halo = nullptr, *d_sph_gid = nullptr;
    int n_prims = 0, n_others = 0, n_halo = 0;
    int n_spheres = 0;
    int n_mats = 0;
    bool have_scene = false;
    bool have_camera = false;
    rt_camera cam{};
    // Device buffers are kept and only ever grown: re-uploading a scene of the same size and
    // rendering a frame of the same size again allocates nothing.
    std::map<void **, size_t> capacity; // bytes behind each pointer field above and below
    uint64_t n_allocs = 0;              // hipMalloc calls made for this context
    // rt_render_host's own stream, outputs and timing
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    float last_kernel_ms = 0.f;
    void *frame_rgba = nullptr, *frame_accum = nullptr;
    uint64_t *frame_rays = nullptr; // closest-hit queries of the frame being rendered for the host, then its scans
    bool count_scans = false;        // the next launch reports its scans in frame_rays[1]
    uint64_t last_scans = 0;
    // rt_first_hit_ex_host's staging buffers
    void *fh_org = nullptr, *fh_dir = nullptr, *fh_id = nullptr, *fh_t = nullptr, *fh_nrm = nullptr, *fh_front = nullptr,
         *fh_exp = nullptr;
};

namespace {

int fail(rt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(ctx, RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// Makes *field point at >= bytes of device memory on the context's device (which must be current),
// keeping the allocation it has when that is large enough.
hipError_t reserve(rt_ctx *ctx, void **field, size_t bytes) {
    if (bytes == 0) bytes = 16; // kernels always get valid pointers
    size_t &cap = ctx->capacity[field];
    if (*field && cap >= bytes) return hipSuccess;
    if (*field) (void)hipFree(*field);
    *field = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(field, bytes);
    if (e != hipSuccess) return e;
    ctx->n_allocs++;
    cap = bytes;
    return hipSuccess;
}

void free_all(rt_ctx *ctx) {
    for (auto &kv : ctx->capacity)
        if (*kv.first) {
            (void)hipFree(*kv.first);
            *kv.first = nullptr;
        }
    ctx->capacity.clear();
    ctx->have_scene = false;
}

int check_params(rt_ctx *ctx, const rt_params *p) {
    if (!ctx) return RT_ERR_INVALID;
    if (!p) return fail(ctx, RT_ERR_INVALID, "params is null");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (!ctx->have_camera) return fail(ctx, RT_ERR_STATE, "no camera set");
    if (p->width <= 0 || p->height <= 0) return fail(ctx, RT_ERR_INVALID, "width/height must be positive");
    if ((long long)p->width * (long long)p->height > 0x7fffffffLL)
        return fail(ctx, RT_ERR_INVALID, "width*height exceeds 2^31-1");
    if (p->samples_per_pixel <= 0) return fail(ctx, RT_ERR_INVALID, "samples_per_pixel must be positive");
    if (p->max_depth < 0) return fail(ctx, RT_ERR_INVALID, "max_depth must be non-negative");
    if (p->sample_offset < 0) return fail(ctx, RT_ERR_INVALID, "sample_offset must be non-negative");
    if (p->strip_stride <= 0 || p->strip_first < 0 || p->strip_first >= p->strip_stride)
        return fail(ctx, RT_ERR_INVALID, "need 0 <= strip_first < strip_stride");
    if ((long long)p->samples_per_pixel * 64LL > 0xffffffffLL)
        return fail(ctx, RT_ERR_INVALID, "samples_per_pixel too large for one call (max 2^26-1)");
    return RT_OK;
}

int local_strips(const rt_params *p) {
    int strips = (p->height + RT_STRIP_ROWS - 1) / RT_STRIP_ROWS;
    if (p->strip_first >= strips) return 0;
    return (strips - p->strip_first + p->strip_stride - 1) / p->strip_stride;
}

} // namespace

#ifdef RT_SECTIONS
extern "C" __attribute__((visibility("default"))) int rt_debug_sections(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sections), sizeof(g_sections)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sections), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef RT_STATS
extern "C" __attribute__((visibility("default"))) int rt_debug_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(g_stats)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_times), sizeof(g_times)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof z) != hipSuccess) return -1;
        unsigned long long t[8] = {~0ull, ~0ull, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_times), t, sizeof t) != hipSuccess) return -1;
    }
    return 0;
}
#endif

RT_API int rt_version(void) { return 104; }

RT_API void rt_rng4(uint32_t pixel, uint32_t sample, uint32_t call, uint32_t seed, uint32_t out[4]) {
    rng4(pixel, sample, call, seed, out[0], out[1], out[2], out[3]);
}

RT_API int rt_rng4_device_host(rt_ctx *ctx, int n, const uint32_t *h_keys, uint32_t *h_out) {
    if (!ctx) return RT_ERR_INVALID;
    if (n < 0 || (n > 0 && (!h_keys || !h_out))) return fail(ctx, RT_ERR_INVALID, "bad key / output array");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n * 16;
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_org, bytes));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_dir, bytes));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->fh_org, h_keys, bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(rng_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       (const uint32_t *)ctx->fh_org, (uint32_t *)ctx->fh_dir);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(h_out, ctx->fh_dir, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

RT_API int rt_filter_describe(rt_ctx *ctx, rt_filter_info *out) {
    if (!ctx) return RT_ERR_INVALID;
    if (!out) return fail(ctx, RT_ERR_INVALID, "out is null");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    memset(out, 0, sizeof *out);
    if (!ctx->mf_usable) return RT_OK;
    out->in_use = ctx->mf_wanted ? 1 : 0;
    out->center[0] = ctx->mf.cx;
    out->center[1] = ctx->mf.cy;
    out->center[2] = ctx->mf.cz;
    out->radius = MF_SCALE / ctx->mf.scale;
    out->margin = MF_MARGIN;
    out->margin_far = MF_MARGIN_FAR;
    out->n_rows = ctx->mf_cols;
    out->n_table_rows = ctx->mf.n_blocks * MF_BLOCK;
    out->n_big = ctx->mf.n_big;
    for (int i = 0; i < ctx->mf.n_big; ++i) out->big[i] = ctx->mf.big[i];
    return RT_OK;
}

RT_API int rt_filter_values_host(rt_ctx *ctx, int n_rays, const float *h_org, const float *h_dir, float *h_values) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (!ctx->mf_usable) return fail(ctx, RT_ERR_UNSUPPORTED, "the uploaded scene has no filter table");
    if (n_rays < 0 || (n_rays > 0 && (!h_org || !h_dir || !h_values))) return fail(ctx, RT_ERR_INVALID, "bad ray / value array");
    if (n_rays == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t rb = (size_t)n_rays * 12, vb = (size_t)n_rays * (size_t)ctx->mf_cols * 4;
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_org, rb));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_dir, rb));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_nrm, vb));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->fh_org, h_org, rb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->fh_dir, h_dir, rb, hipMemcpyHostToDevice, ctx->stream));
    // columns of spheres that have no row (decided outside the table) and the padding read +1: they never pass
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->fh_nrm, 0x3f800000, vb / 4, ctx->stream));
    hipLaunchKernelGGL(filter_probe_kernel, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, ctx->stream, n_rays,
                       (const float *)ctx->fh_org, (const float *)ctx->fh_dir, (const uint4v *)ctx->d_mf_table,
                       (const int *)ctx->d_mf_row_idx, ctx->mf_cols, ctx->mf, (float *)ctx->fh_nrm);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(h_values, ctx->fh_nrm, vb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

RT_API int rt_output_rows(const rt_params *p) {
    if (!p || p->strip_stride <= 0) return 0;
    if (p->strip_stride == 1) return p->height;
    return local_strips(p) * RT_STRIP_ROWS;
}

RT_API int rt_create(int device, rt_ctx **out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RT_ERR_HIP;
    if (device < 0 || device >= count) return RT_ERR_INVALID;
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return RT_ERR_INVALID;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        return RT_ERR_HIP;
    }
    // allow the render kernels their dynamic LDS budget
    size_t lds_max = WAVE_STATE_BYTES + (size_t)RESIDENT_MAX / 4 * 5 * sizeof(float4);
    const void *lds_kernels[] = {reinterpret_cast<const void *>(&render_kernel<SRC_LDS_RESIDENT, false>),
                                 reinterpret_cast<const void *>(&render_kernel<SRC_LDS_STREAMED, false>),
                                 reinterpret_cast<const void *>(&render_kernel<SRC_LDS_RESIDENT, true>),
                                 reinterpret_cast<const void *>(&render_kernel<SRC_LDS_STREAMED, true>)};
    for (const void *k : lds_kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) != hipSuccess) {
            delete ctx;
            return RT_ERR_HIP;
        }
    // the matrix-core variant wants the whole LDS of a CU; a device that will not give it runs the other variants
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<SRC_MFMA, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)MF_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        ctx->mf_wanted = false;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<SRC_MFMA, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)MF_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        ctx->mf_full_ok = false;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<SRC_MFMA_STREAMED, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)MF_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        ctx->mf_wanted = false;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<SRC_MFMA_STREAMED, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)MF_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        ctx->mf_full_streamed_ok = false;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cus = prop.multiProcessorCount;
    if (scratch_ptr(device, &ctx->d_scratch) != hipSuccess ||
        reserve(ctx, (void **)&ctx->d_queue, QUEUE_SLOTS * QUEUE_SLOT_BYTES) != hipSuccess ||
        reserve(ctx, (void **)&ctx->frame_rays, 16) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    if (const char *w = getenv("RT_AMD_WGS_PER_CU")) { int v = atoi(w); if (v >= 1 && v <= 64) ctx->wgs_per_cu = v; }
    // where the scan takes sphere data from: scalar loads (default, fastest measured) or LDS
    // broadcast reads ("lds": resident copy for small scenes, streamed tiles for large ones)
    if (const char *t = getenv("RT_AMD_MF_FULL_MIN")) { int v = atoi(t); if (v >= MF_MIN_SPHERES) ctx->mf_full_min = v; }
    if (const char *t = getenv("RT_AMD_PRIM_MIN")) { int v = atoi(t); if (*t >= '0' && *t <= '9' && v <= 64) ctx->prim_min = ctx->prim_min_full = ctx->prim_min_streamed = v; }
    if (const char *t = getenv("RT_AMD_TILE")) { int v = atoi(t); if (*t >= '0' && *t <= '9' && v <= 6) ctx->force_tile = v; }
    const char *src = getenv("RT_AMD_SPHERE_SOURCE");
    ctx->scalar_source = !(src && strcmp(src, "lds") == 0);
    ctx->mf_wanted = ctx->mf_wanted && !(src && (strcmp(src, "lds") == 0 || strcmp(src, "scalar") == 0));
    *out = ctx;
    return RT_OK;
}

RT_API void rt_destroy(rt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_all(ctx);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

RT_API const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

namespace {

DeviceScene device_scene(const rt_ctx *ctx) {
    DeviceScene sc;
    sc.sph = ctx->d_sph;
    sc.pairs = ctx->d_pairs;
    sc.r2q = ctx->d_r2q;
    sc.mat_id = ctx->d_mat_id;
    sc.mats = ctx->d_mats;
    sc.sph_mats = ctx->d_sph_mats;
    sc.lights = ctx->d_lights;
    sc.mf_table = ctx->d_mf_table;
    sc.mf_rows = ctx->d_mf_rows;
    sc.mf_row_idx = ctx->d_mf_row_idx;
    sc.full.prims = as_const(ctx->d_prims);
    sc.full.verts = as_const(ctx->d_verts);
    sc.full.others = as_const(ctx->d_others);
    sc.full.halo = as_const(ctx->d_halo);
    sc.full.sph_gid = as_const(ctx->d_sph_gid);
    sc.full.n_others = ctx->n_others;
    sc.full.n_halo = ctx->n_halo;
    return sc;
}

template <typename T>
int to_device(rt_ctx *ctx, T **dst, const std::vector<T> &src) {
    HIP_TRY(ctx, reserve(ctx, (void **)dst, src.size() * sizeof(T)));
    if (!src.empty()) HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

// float -> f16 bits, round to nearest even (values beyond the f16 range clamp to its largest)
uint16_t half_rn(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int e = (int)((x >> 23) & 0xffu) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (e >= 31) return (uint16_t)(sign | 0x7bffu);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = sign | ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++; // a carry into the exponent is the right result
    return (uint16_t)((h & 0x7fffu) >= 0x7c00u ? (sign | 0x7bffu) : h);
}
double half_value(uint16_t h) {
    const int e = (h >> 10) & 0x1f, m = h & 0x3ff;
    const double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp((double)(m | 0x400), e - 25);
    return (h & 0x8000) ? -v : v;
}

// The per-sphere side of the matrix-core filter (rt_mfma.h): frame, big spheres, and the table in
// the lane layout of the 32x32x16 A operand.  `ok` is false when the scene cannot use it.
struct MfTable {
    std::vector<uint32_t> words;
    std::vector<float4> rows;    // the sphere of every table row (padding: zeros) ...
    std::vector<int32_t> row_idx; // ... and its index in the scene's sphere list (padding: -1)
    MfmaFrame fr;
    bool ok = false;
};

// LDS the matrix-core variants need: the whole table (plus the block the read-ahead touches), or two tiles
size_t mf_lds_resident(int n_blocks, bool full = false) {
    return wave_state_bytes_of(SRC_MFMA, full) + (size_t)(n_blocks + 1) * MF_BLOCK_WORDS * 16;
}
constexpr size_t mf_lds_streamed(bool full = false) {
    return wave_state_bytes_of(SRC_MFMA_STREAMED, full) + 2 * (size_t)MF_TILE_WORDS * 16;
}
static_assert(mf_lds_streamed(false) <= MF_LDS_BYTES && mf_lds_streamed(true) <= MF_LDS_BYTES,
              "two tiles and the wave state fit in LDS");
// a table is padded to whole tiles when the sphere-only kernel cannot keep it resident; the general kernel, whose
// workgroup is smaller, then streams it too or holds all of it
static_assert(wave_state_bytes_of(SRC_MFMA, true) <= wave_state_bytes_of(SRC_MFMA, false),
              "whatever the general kernel must stream is padded to whole tiles");

MfTable build_mf_table(const std::vector<float4> &sph) {
    MfTable T;
    memset(&T.fr, 0, sizeof T.fr);
    const int n = (int)sph.size();
    if (n < MF_MIN_SPHERES) return T;
    for (const float4 &s : sph)
        if (!std::isfinite(s.x) || !std::isfinite(s.y) || !std::isfinite(s.z) || !std::isfinite(s.w)) return T;
    std::vector<char> big((size_t)n, 0);
    double c0[3] = {0, 0, 0}, Rs = 0;
    // bounding-box centre of the spheres in the table, their reach from it, and the largest of them
    auto bound = [&](double c[3], int &far_i) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            if (big[i]) continue;
            const double cc[3] = {sph[i].x, sph[i].y, sph[i].z}, r = std::fabs((double)sph[i].w);
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], cc[k] - r);
                hi[k] = std::max(hi[k], cc[k] + r);
            }
            ++cnt;
        }
        far_i = -1;
        if (cnt == 0) return 0.0;
        for (int k = 0; k < 3; ++k) c[k] = 0.5 * (lo[k] + hi[k]);
        double reach = 0, largest = -1;
        for (int i = 0; i < n; ++i) {
            if (big[i]) continue;
            const double dx = sph[i].x - c[0], dy = sph[i].y - c[1], dz = sph[i].z - c[2], r = std::fabs((double)sph[i].w);
            reach = std::max(reach, std::sqrt(dx * dx + dy * dy + dz * dz) + r);
            if (r > largest) { largest = r; far_i = i; }
        }
        return reach;
    };
    int n_big = 0, far_i = -1;
    Rs = bound(c0, far_i);
    // the largest sphere is decided outside the table if the cloud is less than half as wide without it
    while (n_big < MF_MAX_BIG && n - n_big > 1 && far_i >= 0) {
        big[far_i] = 1;
        double c1[3];
        int far1 = -1;
        const double R1 = bound(c1, far1);
        if (!(R1 < 0.5 * Rs)) { big[far_i] = 0; break; }
        T.fr.big[n_big++] = far_i;
        memcpy(c0, c1, sizeof c0);
        Rs = R1;
        far_i = far1;
    }
    if (!(Rs > 1e-30) || !std::isfinite(Rs)) return T;
    const float scale = MF_SCALE / (float)Rs;
    if (!std::isfinite(scale) || !(scale > 0.0f)) return T;
    // the kernel computes O' = (O - c0) * scale in float: the table follows those floats exactly
    const float c0f[3] = {(float)c0[0], (float)c0[1], (float)c0[2]};
    const double inv_rs = (double)scale / (double)MF_SCALE;
    // Rows: the spheres of the table along a Morton curve through the cloud (10 bits per axis of the frame's cube),
    // ties and equal cells in scene order, so that a block of 32 rows is a compact patch.
    std::vector<int32_t> order;
    order.reserve((size_t)n);
    for (int i = 0; i < n; ++i)
        if (!big[i]) order.push_back(i);
    {
        auto spread = [](uint32_t v) { // 10 bits -> every third bit
            uint64_t r = 0;
            for (int b = 0; b < 10; ++b) r |= (uint64_t)((v >> b) & 1u) << (3 * b);
            return r;
        };
        std::vector<uint64_t> key((size_t)n, 0);
        for (int i : order) {
            uint64_t k = 0;
            const double cc[3] = {sph[i].x, sph[i].y, sph[i].z};
            for (int a = 0; a < 3; ++a) {
                double u = ((cc[a] - c0[a]) / Rs + 1.0) * 0.5; // the cloud lies in [-1, 1]^3 of the frame
                u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
                k |= spread((uint32_t)(u * 1023.0)) << a;
            }
            key[i] = k;
        }
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
    }
    const int n_table = (int)order.size();
    int n_blocks = (n_table + MF_BLOCK - 1) / MF_BLOCK;
    // a table that does not fit in LDS is streamed through it in whole tiles
    if (mf_lds_resident(n_blocks) > MF_LDS_BYTES) n_blocks = (n_blocks + MF_TILE_BLOCKS - 1) / MF_TILE_BLOCKS * MF_TILE_BLOCKS;
    T.fr.cx = c0f[0]; T.fr.cy = c0f[1]; T.fr.cz = c0f[2];
    T.fr.scale = scale;
    T.fr.n_blocks = n_blocks;
    T.fr.n_big = n_big;
    // one spare block of rows that never pass follows the list: the scan reads one block ahead
    const int n_rows = (n_blocks + 1) * MF_BLOCK;
    T.words.assign((size_t)n_rows * MF_BLOCK_WORDS / MF_BLOCK * 4, 0u);
    T.rows.assign((size_t)n_rows, make_float4(0, 0, 0, 0));
    T.row_idx.assign((size_t)n_rows, -1);
    for (int row = 0; row < n_rows; ++row) {
        uint16_t hi[9] = {0}, lo[9] = {0}, q_hi = half_rn(MF_NEVER), q_lo = 0;
        const uint16_t unit = half_rn(-MF_SCALE);
        if (row < n_table) {
            const int si = order[(size_t)row];
            T.rows[(size_t)row] = sph[si];
            T.row_idx[(size_t)row] = si;
            const double x = ((double)sph[si].x - c0f[0]) * inv_rs, y = ((double)sph[si].y - c0f[1]) * inv_rs,
                         z = ((double)sph[si].z - c0f[2]) * inv_rs, r = (double)sph[si].w * inv_rs;
            const double feat[9] = {x * x, y * y, z * z, 2 * x * y, 2 * x * z, 2 * y * z, 2 * x, 2 * y, 2 * z};
            for (int f = 0; f < 9; ++f) {
                const double S = -(double)MF_SCALE * feat[f];
                hi[f] = half_rn((float)S);
                lo[f] = half_rn((float)(S - half_value(hi[f])));
            }
            const double Q = -(double)MF_SCALE * (r * r - (x * x + y * y + z * z));
            q_hi = half_rn((float)Q);
            q_lo = half_rn((float)(Q - half_value(q_hi)));
        }
        uint32_t w[16];
        auto pack = [](uint16_t a, uint16_t b) { return (uint32_t)a | ((uint32_t)b << 16); };
        for (int f = 0; f < 9; ++f) w[f] = pack(hi[f], hi[f]);
        for (int g = 0; g < 4; ++g) w[9 + g] = pack(lo[2 * g], lo[2 * g + 1]);
        w[13] = pack(lo[8], q_hi);
        w[14] = pack(q_lo, 0);
        w[15] = pack(unit, unit);
        const int blk = row / MF_BLOCK, r = mf_row_of_sphere(row % MF_BLOCK);
        for (int step = 0; step < 2; ++step)
            for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 4; ++i)
                    T.words[(((size_t)blk * 2 + step) * 64 + (size_t)(32 * h + r)) * 4 + i] = w[8 * step + 4 * h + i];
    }
    T.ok = true;
    return T;
}

// Builds every device array of a scene from the general primitive list.
int upload_prims(rt_ctx *ctx, int n_prims, const rt_prim *h_prims, int n_vertices, const float *h_vertices,
                 int n_mats, const rt_material *h_mats) {
    if (n_prims < 0 || n_mats < 0 || n_vertices < 0) return fail(ctx, RT_ERR_INVALID, "negative count");
    if (n_prims > 0 && (!h_prims || !h_mats || n_mats == 0)) return fail(ctx, RT_ERR_INVALID, "null scene array");
    if (n_vertices > 0 && !h_vertices) return fail(ctx, RT_ERR_INVALID, "null vertex array");
    for (int i = 0; i < n_prims; ++i) {
        const rt_prim &pr = h_prims[i];
        if (pr.type > RT_PRIM_POLYGON) return fail(ctx, RT_ERR_UNSUPPORTED, "unknown primitive type");
        if (pr.mat >= (uint32_t)n_mats) return fail(ctx, RT_ERR_INVALID, "mat_id out of range");
        if (pr.type == RT_PRIM_POLYGON &&
            ((unsigned long long)pr.first_vertex + pr.n_vertices > (unsigned long long)n_vertices))
            return fail(ctx, RT_ERR_INVALID, "polygon vertex range outside the vertex array");
    }
    for (int i = 0; i < n_mats; ++i)
        if (h_mats[i].type > RT_MAT_EMISSIVE) return fail(ctx, RT_ERR_UNSUPPORTED, "unknown material type");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // a render still in flight on the context's own stream reads the arrays about to be overwritten
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_scene = false;

    std::vector<DPrim> prims((size_t)n_prims);
    std::vector<float4> sph;
    std::vector<uint32_t> mid;
    std::vector<int32_t> others, halo, gid;
    bool any_selected = false;
    for (int i = 0; i < n_prims; ++i) any_selected = any_selected || h_prims[i].selected != 0;
    for (int i = 0; i < n_prims; ++i) {
        const rt_prim &h = h_prims[i];
        DPrim &d = prims[i];
        memset(&d, 0, sizeof d);
        d.type = h.type;
        d.mat = h.mat;
        d.first_vertex = h.first_vertex;
        d.n_vertices = h.n_vertices;
        memcpy(d.p, h.p, sizeof d.p);
        d.selected = h.selected ? 1u : 0u;
        d.sph_local = -1;
        if (h.type == RT_PRIM_SPHERE) {
            d.sph_local = (int32_t)sph.size();
            sph.push_back(make_float4(h.p[0], h.p[1], h.p[2], h.p[3]));
            mid.push_back(h.mat);
            gid.push_back(i);
        } else {
            others.push_back(i);
        }
        // the halo pass changes a result only when something is selected; a plane answers it with
        // its plain hit whether selected or not (inc/RTObjects.h:192-194)
        if (any_selected && (h.selected || h.type == RT_PRIM_PLANE)) halo.push_back(i);
    }
    const int n_spheres = (int)sph.size();
    // pair-interleaved scan words: padded to a multiple of eight spheres with spheres that cannot
    // pass the filter (q = -1e30), plus four spare quads so that the scan's read-ahead never leaves
    // the arrays.  q = r^2 - |C|^2 is formed in double and enlarged by the filter's margin.
    const int n_quads = ((n_spheres + 7) / 8) * 2;
    const size_t n_slots = (size_t)(n_quads + 4) * 4;
    std::vector<float4> pairs(n_slots, make_float4(0, 0, 0, 0)); // n_slots spheres -> n_slots/2 pairs -> n_slots words
    std::vector<float4> r2q(n_slots / 4, make_float4(0, 0, 0, 0));
    for (size_t i = 0; i < n_slots; ++i) {
        float x = 0, y = 0, z = 0, q = -1e30f, r2 = 0;
        if ((int)i < n_spheres) {
            const float4 &sp = sph[i];
            x = sp.x; y = sp.y; z = sp.z;
            r2 = sp.w * sp.w;
            const double c2 = (double)x * x + (double)y * y + (double)z * z, rr = (double)sp.w * sp.w;
            q = std::nextafter((float)(rr - c2 + (double)FILTER_MARGIN * (c2 + rr)), INFINITY);
        }
        float4 &A = pairs[(i / 2) * 2], &B = pairs[(i / 2) * 2 + 1];
        if (i % 2 == 0) { A.x = x; A.z = y; B.x = z; B.z = q; }
        else { A.y = x; A.w = y; B.y = z; B.w = q; }
        reinterpret_cast<float *>(&r2q[i / 4])[i % 4] = r2;
    }
    std::vector<DMat> mats((size_t)n_mats);
    for (int i = 0; i < n_mats; ++i) {
        DMat m;
        const rt_material &h = h_mats[i];
        float tbits;
        uint32_t ty = h.type;
        memcpy(&tbits, &ty, 4);
        m.albedo_type = make_float4(h.albedo[0], h.albedo[1], h.albedo[2], tbits);
        m.emission_fuzz = make_float4(h.emission[0], h.emission[1], h.emission[2], h.fuzz);
        m.ior_diffuse = make_float4(h.ior, h.diffuse[0], h.diffuse[1], h.diffuse[2]);
        mats[i] = m;
    }
    if (mats.empty()) {
        DMat m;
        memset(&m, 0, sizeof m);
        mats.push_back(m);
    }
    std::vector<float> verts(h_vertices, h_vertices + 3 * (size_t)n_vertices);
    std::vector<DMat> sph_mats(mid.size());
    for (size_t i = 0; i < mid.size(); ++i) sph_mats[i] = mats[mid[i]];
    int rc;
    if ((rc = to_device(ctx, &ctx->d_sph, sph)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_pairs, pairs)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_r2q, r2q)) != RT_OK) return rc;
    MfTable mft = build_mf_table(sph);
    ctx->mf_usable = false;
    if (mft.ok) {
        HIP_TRY(ctx, reserve(ctx, (void **)&ctx->d_mf_table, mft.words.size() * 4));
        HIP_TRY(ctx, hipMemcpy(ctx->d_mf_table, mft.words.data(), mft.words.size() * 4, hipMemcpyHostToDevice));
        if ((rc = to_device(ctx, &ctx->d_mf_rows, mft.rows)) != RT_OK) return rc;
        if ((rc = to_device(ctx, &ctx->d_mf_row_idx, mft.row_idx)) != RT_OK) return rc;
        const int padded = ((int)sph.size() + MF_BLOCK - 1) / MF_BLOCK * MF_BLOCK;
        ctx->mf_cols = std::max(mft.fr.n_blocks * MF_BLOCK, padded);
        ctx->mf = mft.fr;
        // how many workgroups of the variant this table takes stay resident on a CU: asked once per scene, so a
        // render issues nothing but its counter reset and its launch
        const bool fits = mf_lds_resident(mft.fr.n_blocks) <= MF_LDS_BYTES;
        int resident = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &resident,
                fits ? reinterpret_cast<const void *>(&render_kernel<SRC_MFMA, false>)
                     : reinterpret_cast<const void *>(&render_kernel<SRC_MFMA_STREAMED, false>),
                wg_threads_of(fits ? SRC_MFMA : SRC_MFMA_STREAMED),
                fits ? mf_lds_resident(mft.fr.n_blocks) : mf_lds_streamed()) != hipSuccess)
            resident = 0;
        (void)hipGetLastError();
        ctx->mf_resident = resident;
        // the general kernel keeps the table resident in LDS beside its larger path state when it fits ...
        int resident_full = 0;
        if (ctx->mf_full_ok && mf_lds_resident(mft.fr.n_blocks, true) <= MF_LDS_BYTES &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident_full,
                                                         reinterpret_cast<const void *>(&render_kernel<SRC_MFMA, true>),
                                                         wg_threads_of(SRC_MFMA, true),
                                                         mf_lds_resident(mft.fr.n_blocks, true)) != hipSuccess)
            resident_full = 0;
        (void)hipGetLastError();
        ctx->mf_resident_full = resident_full;
        // ... and streams a table that does not (the sphere-only kernel's does not fit either then: whole tiles)
        int streamed_full = 0;
        if (resident_full < 1 && !fits && ctx->mf_full_streamed_ok &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&streamed_full,
                                                         reinterpret_cast<const void *>(&render_kernel<SRC_MFMA_STREAMED, true>),
                                                         wg_threads_of(SRC_MFMA_STREAMED, true), mf_lds_streamed(true)) != hipSuccess)
            streamed_full = 0;
        (void)hipGetLastError();
        ctx->mf_streamed_full = streamed_full;
        ctx->mf_usable = resident >= 1; // a device that cannot hold the workgroup scans with the packed filter
    }
    if ((rc = to_device(ctx, &ctx->d_mat_id, mid)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_mats, mats)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_sph_mats, sph_mats)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_prims, prims)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_verts, verts)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_others, others)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_halo, halo)) != RT_OK) return rc;
    if ((rc = to_device(ctx, &ctx->d_sph_gid, gid)) != RT_OK) return rc;
    ctx->n_spheres = n_spheres;
    ctx->n_quads = n_quads;
    ctx->n_mats = n_mats;
    ctx->n_prims = n_prims;
    ctx->n_others = (int)others.size();
    ctx->n_halo = (int)halo.size();
    ctx->have_scene = true;
    return RT_OK;
}

} // namespace

RT_API int rt_upload_scene(rt_ctx *ctx, int n_spheres, const rt_sphere *h_spheres, const uint32_t *h_mat_id,
                           int n_mats, const rt_material *h_mats) {
    if (!ctx) return RT_ERR_INVALID;
    if (n_spheres < 0 || n_mats < 0) return fail(ctx, RT_ERR_INVALID, "negative count");
    if (n_spheres > 0 && (!h_spheres || !h_mat_id || !h_mats || n_mats == 0))
        return fail(ctx, RT_ERR_INVALID, "null scene array");
    std::vector<rt_prim> prims((size_t)n_spheres);
    for (int i = 0; i < n_spheres; ++i) {
        rt_prim &pr = prims[i];
        memset(&pr, 0, sizeof pr);
        pr.type = RT_PRIM_SPHERE;
        pr.mat = h_mat_id[i];
        pr.p[0] = h_spheres[i].cx;
        pr.p[1] = h_spheres[i].cy;
        pr.p[2] = h_spheres[i].cz;
        pr.p[3] = h_spheres[i].r;
    }
    return upload_prims(ctx, n_spheres, prims.data(), 0, nullptr, n_mats, h_mats);
}

RT_API int rt_upload_scene_ex(rt_ctx *ctx, int n_prims, const rt_prim *h_prims, int n_vertices,
                              const float *h_vertices, int n_mats, const rt_material *h_mats) {
    if (!ctx) return RT_ERR_INVALID;
    return upload_prims(ctx, n_prims, h_prims, n_vertices, h_vertices, n_mats, h_mats);
}

RT_API int rt_upload_lights(rt_ctx *ctx, int n_lights, const rt_light *h_lights) {
    if (!ctx) return RT_ERR_INVALID;
    if (n_lights < 0 || (n_lights > 0 && !h_lights)) return fail(ctx, RT_ERR_INVALID, "bad light array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_lights = 0;
    if (n_lights == 0) return RT_OK;
    std::vector<DLight> l((size_t)n_lights);
    for (int i = 0; i < n_lights; ++i) {
        const rt_light &h = h_lights[i];
        l[i].pos_pow = make_float4(h.position[0], h.position[1], h.position[2], h.view_pow);
        l[i].ambient = make_float4(h.ambient[0], h.ambient[1], h.ambient[2], 0);
        l[i].diffuse = make_float4(h.diffuse[0], h.diffuse[1], h.diffuse[2], 0);
    }
    int rc = to_device(ctx, &ctx->d_lights, l);
    if (rc != RT_OK) return rc;
    ctx->n_lights = n_lights;
    return RT_OK;
}

RT_API int rt_set_camera(rt_ctx *ctx, const rt_camera *cam) {
    if (!ctx) return RT_ERR_INVALID;
    if (!cam) return fail(ctx, RT_ERR_INVALID, "camera is null");
    ctx->cam = *cam;
    ctx->have_camera = true;
    return RT_OK;
}

namespace {
// Launches the render kernel of `p` with `queue` (device memory, 256 bytes) as its work counter.
int render_impl(rt_ctx *ctx, const rt_params *p, void *d_rgba, void *d_accum, uint64_t *d_rays, void *stream,
                unsigned *queue) {
    int rc = check_params(ctx, p);
    if (rc != RT_OK) return rc;
    if (!d_rgba) return fail(ctx, RT_ERR_INVALID, "d_rgba is null");
    if ((p->flags & RT_FLAG_ACCUMULATE) && !d_accum)
        return fail(ctx, RT_ERR_INVALID, "RT_FLAG_ACCUMULATE needs d_accum");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    KParams k;
    k.width = p->width;
    k.height = p->height;
    k.spp = p->samples_per_pixel;
    k.max_depth = p->max_depth;
    k.sample_offset = p->sample_offset;
    k.seed = p->seed;
    k.flags = p->flags;
    k.strip_first = p->strip_first;
    k.strip_stride = p->strip_stride;
    // Work-item tiles: the smallest of seven shapes (1x1 ... 8x8 pixels) that holds RT_ITEM_SAMPLES samples.  Small
    // items keep the longest one short (a tile whose every path runs to the depth limit costs 25 times the
    // average, and the kernel cannot end before its last item does); much smaller ones run dry before the long
    // paths of the tile before them have ended (a wave holds only two tiles), and every item costs a trip to the
    // global counter, a descriptor and a partial-sector store.  Measured: 100 samples per pixel 800 / 1600 / 3200
    // samples per item 4.65 / 4.32 / 5.12 ms; 500 per pixel 500 / 1000 / 2000 / 4000 per item 26.97 / 26.42 /
    // 26.51 / 28.49 ms; 1000 per pixel 1000 / 2000 / 4000 per item 551.9 / 547.0 / 546.0 ms; 256 per pixel on
    // 10 001 spheres 512 / 1024 / 2048 / 4096 per item 4689 / 4684 / 4722 / 4750 ms.  The radiance sums are exact
    // integers, so the shape changes no byte.
    int nstrips = local_strips(p);
    if (nstrips == 0) return RT_OK;
    {
        // small tiles are single rows (a tile's row is one store: fewer partial sectors in HBM, same time); from 16
        // pixels on they are squares (measured on the four-sphere scene: 8x2 is 5 % slower than 4x4)
        static const int shapes[7][2] = {{0, 0}, {1, 0}, {2, 0}, {2, 1}, {2, 2}, {3, 2}, {3, 3}}; // 1x1, 2x1, 4x1, 4x2, 4x4, 8x4, 8x8 (log2 w, h)
        const int spp = p->samples_per_pixel;
        int pick = 0;
        while (pick < 6 && ((long long)spp << (shapes[pick][0] + shapes[pick][1])) < RT_ITEM_SAMPLES) ++pick;
        // a frame of very many items takes larger ones while every wave still gets 256 of them: an item costs a
        // pass over the sphere list for its primary rays (4K, 1000 samples per pixel, 2 / 4 / 8 / 16 / 32 pixels per
        // item: 367.9 / 358.7 / 355.4 / 356.0 / 357.8 ms; 10 001 spheres, 256 per pixel, 4 / 8 / 16: 3.80 / 3.75 / 3.76 s)
        const long long waves = (long long)ctx->n_cus * 16;
        auto items_at = [&](int s) {
            const int tw = 1 << shapes[s][0], th = 1 << shapes[s][1];
            return (long long)nstrips * ((p->width + tw - 1) / tw) * (RT_STRIP_ROWS / th);
        };
        while (pick < 6 && items_at(pick) >= 256 * waves) ++pick;
        if (ctx->force_tile >= 0) pick = ctx->force_tile;
        k.tile_w_log2 = shapes[pick][0];
        k.tile_h_log2 = shapes[pick][1];
    }
    const int tw = 1 << k.tile_w_log2, th = 1 << k.tile_h_log2;
    k.tiles_x = (p->width + tw - 1) / tw;
    k.tiles_per_strip = k.tiles_x * (RT_STRIP_ROWS / th);
    k.n_quads = ctx->n_quads;
    k.n_spheres = ctx->n_spheres;
    k.n_lights = (p->flags & RT_FLAG_DIRECT_LIGHT) ? ctx->n_lights : 0;
    k.inv_w = 1.0f / (float)p->width;
    k.inv_h = 1.0f / (float)p->height;
    memcpy(k.cam + 0, ctx->cam.center, 12);
    memcpy(k.cam + 3, ctx->cam.upper_left, 12);
    memcpy(k.cam + 6, ctx->cam.right, 12);
    memcpy(k.cam + 9, ctx->cam.down, 12);
    DeviceScene sc = device_scene(ctx);
    const long long items = (long long)nstrips * k.tiles_per_strip;
    if (items > 0x7fffffffLL) return fail(ctx, RT_ERR_INVALID, "too many tiles for one call");
    k.n_items = (int)items;
    // persistent grid: enough workgroups to fill every CU at the kernel's occupancy; the
    // item queue does the balancing, so more workgroups than that would only idle
    long long want = (items + WAVES_PER_WG - 1) / WAVES_PER_WG;
    long long cap = (long long)ctx->n_cus * ctx->wgs_per_cu;
    dim3 grid((unsigned)(want < cap ? want : cap)), block(WG_THREADS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    uchar4 *o_rgba = (uchar4 *)d_rgba;
    float4 *o_acc = (float4 *)d_accum;
    unsigned long long *o_rays = (unsigned long long *)d_rays;
    HIP_TRY(ctx, hipMemsetAsync(queue, 0, QUEUE_SLOT_BYTES, st));
    // the general kernel serves lights, the non-sphere primitives and the selection halo
    const bool full = k.n_lights > 0 || ctx->n_others > 0 || ctx->n_halo > 0;
    size_t lds = WAVE_STATE_BYTES;
    int src = SRC_SCALAR;
    // Sphere words come through scalar loads, except that the general kernel with a short sphere list
    // runs faster on its LDS-resident copy (its scalar variant needs 128 VGPRs and spills SGPRs:
    // measured 13 % slower on scenes of two spheres, 5 % faster on 485 spheres with a light).
    if (!ctx->scalar_source || (full && ctx->n_spheres < FULL_SCALAR_MIN_SPHERES)) {
        const bool resident = ctx->n_quads * 4 <= RESIDENT_MAX;
        src = resident ? SRC_LDS_RESIDENT : SRC_LDS_STREAMED;
        lds += (size_t)(resident ? (ctx->n_quads > 0 ? ctx->n_quads : 1) : STREAM_QUADS) * 5 * sizeof(float4);
    }
    // the general kernel's long sphere lists go through the matrix-core filter too
    const bool mf_full = full && (ctx->mf_resident_full >= 1 || ctx->mf_streamed_full >= 1) && ctx->n_spheres >= ctx->mf_full_min;
    if (ctx->mf_usable && ctx->mf_wanted && (!full || mf_full)) {
        // matrix-core filter: one table per workgroup in LDS (or two tiles of it), as many workgroups as stay resident
        const bool fits = full ? ctx->mf_resident_full >= 1 : mf_lds_resident(ctx->mf.n_blocks) <= MF_LDS_BYTES;
        src = fits ? SRC_MFMA : SRC_MFMA_STREAMED;
        k.mf = ctx->mf;
        lds = fits ? mf_lds_resident(ctx->mf.n_blocks, full) : mf_lds_streamed(full);
        const int resident = full ? (fits ? ctx->mf_resident_full : ctx->mf_streamed_full) : ctx->mf_resident;
        const int wpw = wg_threads_of(src, full) / 64;
        want = (items + wpw - 1) / wpw;
        cap = (long long)ctx->n_cus * resident;
        grid = dim3((unsigned)(want < cap ? want : cap));
        block = dim3(wg_threads_of(src, full));
    } else {
        memset(&k.mf, 0, sizeof k.mf);
    }
    k.prim_min = src == SRC_MFMA ? (full ? ctx->prim_min_full : ctx->prim_min) : src == SRC_MFMA_STREAMED ? ctx->prim_min_streamed : 0;
    const KArgs ka = {k, sc, o_rgba, o_acc, o_rays, ctx->count_scans ? (unsigned long long *)(ctx->frame_rays + 1) : nullptr, queue};
    ctx->count_scans = false;
#define RT_LAUNCH(S, L) hipLaunchKernelGGL((render_kernel<S, L>), grid, block, lds, st, ka)
    if (src == SRC_MFMA) { if (full) RT_LAUNCH(SRC_MFMA, true); else RT_LAUNCH(SRC_MFMA, false); }
    else if (src == SRC_MFMA_STREAMED) { if (full) RT_LAUNCH(SRC_MFMA_STREAMED, true); else RT_LAUNCH(SRC_MFMA_STREAMED, false); }
    else if (src == SRC_SCALAR) { if (full) RT_LAUNCH(SRC_SCALAR, true); else RT_LAUNCH(SRC_SCALAR, false); }
    else if (src == SRC_LDS_RESIDENT) { if (full) RT_LAUNCH(SRC_LDS_RESIDENT, true); else RT_LAUNCH(SRC_LDS_RESIDENT, false); }
    else { if (full) RT_LAUNCH(SRC_LDS_STREAMED, true); else RT_LAUNCH(SRC_LDS_STREAMED, false); }
#undef RT_LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_variant = src + (full ? 8 : 0);
    return RT_OK;
}
} // namespace

RT_API int rt_render(rt_ctx *ctx, const rt_params *p, void *d_rgba, void *d_accum, uint64_t *d_rays,
                     void *stream) {
    if (!ctx) return RT_ERR_INVALID;
    unsigned char *slot = ctx->d_queue + (size_t)(ctx->queue_next++ % QUEUE_SLOTS) * QUEUE_SLOT_BYTES;
    return render_impl(ctx, p, d_rgba, d_accum, d_rays, stream, reinterpret_cast<unsigned *>(slot));
}

RT_API int rt_render_noalloc(rt_ctx *ctx, const rt_params *p, void *d_rgba, void *d_accum, uint64_t *d_rays,
                             void *stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->d_scratch) return fail(ctx, RT_ERR_STATE, "rt_render_noalloc: the device's scratch array was not found");
    // the counter is 32 bits wide: a call may cover at most 2^31 - 1 work items (tiles of 1 to 64 pixels)
    if (p && p->width > 0 && p->height > 0 && (long long)p->width * (long long)p->height > 0x7fffffffLL)
        return fail(ctx, RT_ERR_INVALID, "rt_render_noalloc: more work items than the 32-bit counter in scratch holds");
    const unsigned n = g_scratch_next[ctx->device].fetch_add(1u);
    unsigned char *slot = ctx->d_scratch + (size_t)(n % QUEUE_SLOTS) * QUEUE_SLOT_BYTES;
    return render_impl(ctx, p, d_rgba, d_accum, d_rays, stream, reinterpret_cast<unsigned *>(slot));
}

namespace {
// Renders `p` into the context's own frame buffers on its own stream; asynchronous.  The kernel is
// bracketed by ev_start / ev_stop.  With `h_accum_in` the radiance sums continue from it.
int render_into_frame(rt_ctx *ctx, const rt_params *p, bool want_accum, const float *h_accum_in) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // whole strips, so that a plain frame (strip_stride 1) and a compact one have the same layout
    const size_t rows = ((size_t)rt_output_rows(p) + RT_STRIP_ROWS - 1) / RT_STRIP_ROWS * RT_STRIP_ROWS;
    const size_t npx = rows * (size_t)p->width;
    HIP_TRY(ctx, reserve(ctx, &ctx->frame_rgba, npx * 4));
    if (want_accum) HIP_TRY(ctx, reserve(ctx, &ctx->frame_accum, npx * 16));
    HIP_TRY(ctx, hipMemsetAsync(ctx->frame_rays, 0, 16, ctx->stream));
    ctx->count_scans = true;
    HIP_TRY(ctx, hipMemsetAsync(ctx->frame_rgba, 0, npx * 4, ctx->stream));
    if (want_accum) {
        if (h_accum_in)
            HIP_TRY(ctx, hipMemcpyAsync(ctx->frame_accum, h_accum_in, (size_t)rt_output_rows(p) * (size_t)p->width * 16,
                                        hipMemcpyHostToDevice, ctx->stream));
        else
            HIP_TRY(ctx, hipMemsetAsync(ctx->frame_accum, 0, npx * 16, ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    int rc = rt_render(ctx, p, ctx->frame_rgba, want_accum ? ctx->frame_accum : nullptr, ctx->frame_rays, ctx->stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    return RT_OK;
}
} // namespace

RT_API int rt_render_host(rt_ctx *ctx, const rt_params *p, uint8_t *h_rgba, float *h_accum, uint64_t *h_rays) {
    int rc = check_params(ctx, p);
    if (rc != RT_OK) return rc;
    if (!h_rgba) return fail(ctx, RT_ERR_INVALID, "h_rgba is null");
    if ((p->flags & RT_FLAG_ACCUMULATE) && !h_accum)
        return fail(ctx, RT_ERR_INVALID, "RT_FLAG_ACCUMULATE needs h_accum");
    const size_t npx = (size_t)rt_output_rows(p) * (size_t)p->width;
    if (npx == 0) {
        if (h_rays) *h_rays = 0;
        return RT_OK;
    }
    const bool accumulate = (p->flags & RT_FLAG_ACCUMULATE) != 0;
    rc = render_into_frame(ctx, p, h_accum != nullptr, accumulate ? h_accum : nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h_rgba, ctx->frame_rgba, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (h_accum) HIP_TRY(ctx, hipMemcpyAsync(h_accum, ctx->frame_accum, npx * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (h_rays) HIP_TRY(ctx, hipMemcpyAsync(h_rays, ctx->frame_rays, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&ctx->last_scans, ctx->frame_rays + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev_start, ctx->ev_stop));
    return RT_OK;
}

RT_API uint64_t rt_device_allocations(const rt_ctx *ctx) { return ctx ? ctx->n_allocs : 0; }

RT_API float rt_last_kernel_ms(const rt_ctx *ctx) { return ctx ? ctx->last_kernel_ms : 0.f; }
RT_API int rt_last_kernel_variant(const rt_ctx *ctx) { return ctx ? ctx->last_variant : -1; }
RT_API uint64_t rt_last_scans(const rt_ctx *ctx) { return ctx ? ctx->last_scans : 0; }

RT_API int rt_first_hit_ex(rt_ctx *ctx, int n_rays, const float *d_org, const float *d_dir, float tmin,
                           uint32_t flags, int32_t *d_id, float *d_t, float *d_normal, uint8_t *d_front,
                           uint8_t *d_expanded, void *stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (n_rays < 0) return fail(ctx, RT_ERR_INVALID, "negative ray count");
    if (n_rays == 0) return RT_OK;
    if (!d_org || !d_dir || !d_id) return fail(ctx, RT_ERR_INVALID, "null ray/id array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceScene sc = device_scene(ctx);
    const int expanded_state = (flags & RT_FH_EXPANDED) ? 1 : 0;
    const int general = (ctx->n_others > 0 || (expanded_state && ctx->n_halo > 0)) ? 1 : 0;
    dim3 grid((unsigned)((n_rays + WG_THREADS - 1) / WG_THREADS)), block(WG_THREADS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & RT_FH_REFINE)
        hipLaunchKernelGGL(first_hit_kernel<true>, grid, block, 0, st, n_rays, d_org, d_dir, tmin, sc, ctx->n_spheres,
                           ctx->n_prims, general, expanded_state, d_id, d_t, d_normal, d_front, d_expanded);
    else if (!general && ctx->mf_usable && ctx->mf_wanted)
        // the plain query runs the scan the render kernel runs on this scene
        hipLaunchKernelGGL(first_hit_mfma_kernel, grid, block, 0, st, n_rays, d_org, d_dir, tmin, sc, ctx->mf, d_id, d_t,
                           d_normal, d_front, d_expanded);
    else
        hipLaunchKernelGGL(first_hit_kernel<false>, grid, block, 0, st, n_rays, d_org, d_dir, tmin, sc,
                           ctx->n_spheres, ctx->n_prims, general, expanded_state, d_id, d_t, d_normal, d_front,
                           d_expanded);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

RT_API int rt_first_hit(rt_ctx *ctx, int n_rays, const float *d_org, const float *d_dir, float tmin, uint32_t flags,
                        int32_t *d_id, float *d_t, float *d_normal, uint8_t *d_front, void *stream) {
    return rt_first_hit_ex(ctx, n_rays, d_org, d_dir, tmin, flags, d_id, d_t, d_normal, d_front, nullptr, stream);
}

namespace {
// small RAII helper for the host-pointer wrappers
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};
} // namespace

RT_API int rt_first_hit_ex_host(rt_ctx *ctx, int n_rays, const float *h_org, const float *h_dir, float tmin,
                                uint32_t flags, int32_t *h_id, float *h_t, float *h_normal, uint8_t *h_front,
                                uint8_t *h_expanded) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (n_rays < 0) return fail(ctx, RT_ERR_INVALID, "negative ray count");
    if (n_rays == 0) return RT_OK;
    if (!h_org || !h_dir || !h_id) return fail(ctx, RT_ERR_INVALID, "null ray/id array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)n_rays;
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_org, n * 12));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_dir, n * 12));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_id, n * 4));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_t, n * 4));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_nrm, n * 12));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_front, n));
    HIP_TRY(ctx, reserve(ctx, &ctx->fh_exp, n));
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->fh_org, h_org, n * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->fh_dir, h_dir, n * 12, hipMemcpyHostToDevice, st));
    int rc = rt_first_hit_ex(ctx, n_rays, (const float *)ctx->fh_org, (const float *)ctx->fh_dir, tmin, flags,
                             (int32_t *)ctx->fh_id, (float *)ctx->fh_t, (float *)ctx->fh_nrm, (uint8_t *)ctx->fh_front,
                             (uint8_t *)ctx->fh_exp, st);
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h_id, ctx->fh_id, n * 4, hipMemcpyDeviceToHost, st));
    if (h_t) HIP_TRY(ctx, hipMemcpyAsync(h_t, ctx->fh_t, n * 4, hipMemcpyDeviceToHost, st));
    if (h_normal) HIP_TRY(ctx, hipMemcpyAsync(h_normal, ctx->fh_nrm, n * 12, hipMemcpyDeviceToHost, st));
    if (h_front) HIP_TRY(ctx, hipMemcpyAsync(h_front, ctx->fh_front, n, hipMemcpyDeviceToHost, st));
    if (h_expanded) HIP_TRY(ctx, hipMemcpyAsync(h_expanded, ctx->fh_exp, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

RT_API int rt_first_hit_host(rt_ctx *ctx, int n_rays, const float *h_org, const float *h_dir, float tmin,
                             uint32_t flags, int32_t *h_id, float *h_t, float *h_normal, uint8_t *h_front) {
    return rt_first_hit_ex_host(ctx, n_rays, h_org, h_dir, tmin, flags, h_id, h_t, h_normal, h_front, nullptr);
}

RT_API int rt_scatter_host(rt_ctx *ctx, int n, const uint32_t *h_mat, const float *h_in_dir, const float *h_point,
                           const float *h_normal, const uint8_t *h_front, const float *h_u, float *h_att,
                           float *h_out_org, float *h_out_dir, uint8_t *h_scattered_flag) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (n < 0) return fail(ctx, RT_ERR_INVALID, "negative count");
    if (n == 0) return RT_OK;
    if (!h_mat || !h_in_dir || !h_point || !h_normal || !h_front || !h_u || !h_att || !h_out_org || !h_out_dir ||
        !h_scattered_flag)
        return fail(ctx, RT_ERR_INVALID, "null array");
    for (int i = 0; i < n; ++i)
        if (h_mat[i] >= (uint32_t)ctx->n_mats) return fail(ctx, RT_ERR_INVALID, "material index out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t N = (size_t)n;
    DevBuf mat, ind, pnt, nrm, fr, u, att, oo, od, fl;
    HIP_TRY(ctx, mat.alloc(N * 4));
    HIP_TRY(ctx, ind.alloc(N * 12));
    HIP_TRY(ctx, pnt.alloc(N * 12));
    HIP_TRY(ctx, nrm.alloc(N * 12));
    HIP_TRY(ctx, fr.alloc(N));
    HIP_TRY(ctx, u.alloc(N * 12));
    HIP_TRY(ctx, att.alloc(N * 12));
    HIP_TRY(ctx, oo.alloc(N * 12));
    HIP_TRY(ctx, od.alloc(N * 12));
    HIP_TRY(ctx, fl.alloc(N));
    HIP_TRY(ctx, hipMemcpy(mat.p, h_mat, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ind.p, h_in_dir, N * 12, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(pnt.p, h_point, N * 12, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(nrm.p, h_normal, N * 12, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(fr.p, h_front, N, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(u.p, h_u, N * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, n, ctx->d_mats,
                       (const uint32_t *)mat.p, (const float *)ind.p, (const float *)pnt.p, (const float *)nrm.p,
                       (const uint8_t *)fr.p, (const float *)u.p, (float *)att.p, (float *)oo.p, (float *)od.p,
                       (uint8_t *)fl.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(h_att, att.p, N * 12, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(h_out_org, oo.p, N * 12, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(h_out_dir, od.p, N * 12, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(h_scattered_flag, fl.p, N, hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---------------------------------------------------------------------------
// Device groups: one frame rendered by several GPUs of one process.
//
// Pixels are independent (the reference's own OpenMP loop over pixels, src/Camera.cpp:80-87, relies
// on it) and random streams are keyed on the global pixel id, so member i of n renders strips
// i, i+n, ... of the frame compactly on its own stream, the strips travel to the first member's
// GPU (RCCL send/recv over xGMI between distinct GPUs; plain device copies when a GPU appears
// more than once in the list), and one kernel there deals them back into row order.
// ---------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <chrono>

namespace {

// the few RCCL entry points the gather needs, bound at run time: a process that renders on one GPU
// never loads the library
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(lib, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(lib, "ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};

// dst holds whole strips in row order; src holds the cnt strips r, r+n, ... back to back
__global__ void assemble_strips(uint4 *__restrict__ dst, const uint4 *__restrict__ src, int words_per_strip, int cnt,
                                int r, int n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)cnt * words_per_strip) return;
    const int k = (int)(i / words_per_strip), w = (int)(i - (long long)k * words_per_strip);
    dst[(size_t)(r + k * n) * (size_t)words_per_strip + (size_t)w] = src[i];
}

} // namespace

struct rt_group {
    std::vector<rt_ctx *> m; // member i renders strips i, i + size, ...
    std::vector<int> dev;
    std::string err;
    enum { COPY, RCCL, RCCL_LOOPBACK } transport = COPY;
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    std::vector<hipEvent_t> done;
    void *gather = nullptr; // on member 0's GPU: the compact strips of every member, back to back
    void *frame = nullptr;  // on member 0's GPU: the assembled frame, whole strips
};

namespace {
int gfail(rt_group *g, int code, const std::string &msg) {
    if (g) g->err = msg;
    return code;
}
#define GROUP_TRY(g, expr)                                                                             \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return gfail(g, RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NCCL_TRY(g, expr)                                                                              \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return gfail(g, RT_ERR_HIP, std::string(#expr) + ": " + g->rccl.GetErrorString(r_)); \
    } while (0)
} // namespace

RT_API int rt_group_create(int n_members, const int *devices, rt_group **out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    if (n_members <= 0 || n_members > 64 || !devices) return RT_ERR_INVALID;
    rt_group *g = new (std::nothrow) rt_group();
    if (!g) return RT_ERR_INVALID;
    bool distinct = true;
    for (int i = 0; i < n_members; ++i) {
        rt_ctx *c = nullptr;
        int rc = rt_create(devices[i], &c);
        if (rc != RT_OK) {
            rt_group_destroy(g);
            return rc;
        }
        g->m.push_back(c);
        g->dev.push_back(devices[i]);
        for (int j = 0; j < i; ++j) distinct = distinct && devices[j] != devices[i];
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            rt_group_destroy(g);
            return RT_ERR_HIP;
        }
        g->done.push_back(e);
    }
    // RCCL between distinct GPUs; RT_AMD_GROUP_TRANSPORT=copy forces peer copies, =rccl-loopback
    // (one member only) sends the frame to itself through RCCL, which exercises the same calls
    const char *want = getenv("RT_AMD_GROUP_TRANSPORT");
    const bool loopback = want && strcmp(want, "rccl-loopback") == 0 && n_members == 1;
    if ((distinct && n_members > 1 && !(want && strcmp(want, "copy") == 0)) || loopback) {
        if (g->rccl.load()) {
            g->comms.resize((size_t)n_members);
            if (g->rccl.CommInitAll(g->comms.data(), n_members, devices) == ncclSuccess)
                g->transport = loopback ? rt_group::RCCL_LOOPBACK : rt_group::RCCL;
            else
                g->comms.clear();
        }
        if (g->transport == rt_group::COPY && want && strncmp(want, "rccl", 4) == 0) {
            rt_group_destroy(g);
            return RT_ERR_HIP; // RCCL was asked for by name and is not usable
        }
    }
    if (g->transport == rt_group::COPY)
        for (int i = 1; i < n_members; ++i)
            if (devices[i] != devices[0]) {
                // direct peer writes when the link allows them; the copy works (staged) without
                (void)hipSetDevice(devices[i]);
                (void)hipDeviceEnablePeerAccess(devices[0], 0);
                (void)hipGetLastError();
            }
    *out = g;
    return RT_OK;
}

RT_API void rt_group_destroy(rt_group *g) {
    if (!g) return;
    for (ncclComm_t c : g->comms)
        if (c) (void)g->rccl.CommDestroy(c);
    for (size_t i = 0; i < g->done.size(); ++i)
        if (g->done[i]) (void)hipEventDestroy(g->done[i]);
    for (rt_ctx *c : g->m) rt_destroy(c); // frees g->gather / g->frame with member 0's buffers
    delete g;
}

RT_API const char *rt_group_last_error(const rt_group *g) { return g ? g->err.c_str() : "null group"; }
RT_API int rt_group_size(const rt_group *g) { return g ? (int)g->m.size() : 0; }
RT_API const char *rt_group_transport(const rt_group *g) {
    if (!g) return "";
    return g->transport == rt_group::COPY ? "copy" : (g->transport == rt_group::RCCL ? "rccl" : "rccl-loopback");
}
RT_API uint64_t rt_group_device_allocations(const rt_group *g) {
    uint64_t n = 0;
    if (g)
        for (const rt_ctx *c : g->m) n += c->n_allocs;
    return n;
}

RT_API int rt_group_upload_scene_ex(rt_group *g, int n_prims, const rt_prim *h_prims, int n_vertices,
                                    const float *h_vertices, int n_mats, const rt_material *h_mats) {
    if (!g) return RT_ERR_INVALID;
    for (rt_ctx *c : g->m) {
        int rc = rt_upload_scene_ex(c, n_prims, h_prims, n_vertices, h_vertices, n_mats, h_mats);
        if (rc != RT_OK) return gfail(g, rc, c->err);
    }
    return RT_OK;
}

RT_API int rt_group_upload_lights(rt_group *g, int n_lights, const rt_light *h_lights) {
    if (!g) return RT_ERR_INVALID;
    for (rt_ctx *c : g->m) {
        int rc = rt_upload_lights(c, n_lights, h_lights);
        if (rc != RT_OK) return gfail(g, rc, c->err);
    }
    return RT_OK;
}

RT_API int rt_group_set_camera(rt_group *g, const rt_camera *cam) {
    if (!g) return RT_ERR_INVALID;
    for (rt_ctx *c : g->m) {
        int rc = rt_set_camera(c, cam);
        if (rc != RT_OK) return gfail(g, rc, c->err);
    }
    return RT_OK;
}

RT_API int rt_group_render(rt_group *g, const rt_params *p, void **d_frame, uint64_t *h_rays, rt_group_timing *timing) {
    if (!g) return RT_ERR_INVALID;
    if (!p) return gfail(g, RT_ERR_INVALID, "params is null");
    if (p->strip_first != 0 || p->strip_stride != 1)
        return gfail(g, RT_ERR_INVALID, "a group deals the strips itself: strip_first must be 0 and strip_stride 1");
    if (p->flags & RT_FLAG_ACCUMULATE) return gfail(g, RT_ERR_UNSUPPORTED, "a group renders RGBA8 frames only");
    const int n = (int)g->m.size();
    rt_ctx *root = g->m[0];
    const auto t_begin = std::chrono::steady_clock::now();

    // every member renders its strips on its own stream
    std::vector<rt_params> mp((size_t)n, *p);
    std::vector<size_t> bytes((size_t)n), offset((size_t)n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        mp[i].strip_first = i;
        mp[i].strip_stride = n;
        int rc = check_params(g->m[i], &mp[i]);
        if (rc != RT_OK) return gfail(g, rc, g->m[i]->err);
    }
    const size_t strip_bytes = (size_t)RT_STRIP_ROWS * (size_t)p->width * 4;
    const int strips = (p->height + RT_STRIP_ROWS - 1) / RT_STRIP_ROWS;
    for (int i = 0; i < n; ++i) {
        bytes[i] = (size_t)local_strips(&mp[i]) * strip_bytes;
        offset[i] = total;
        total += bytes[i];
        if (bytes[i] == 0) continue;
        // a lone member renders the plain row-major frame (its compact layout equals it but for the
        // rows of a last partial strip)
        int rc = render_into_frame(g->m[i], &mp[i], false, nullptr);
        if (rc != RT_OK) return gfail(g, rc, g->m[i]->err);
    }

    GROUP_TRY(g, hipSetDevice(g->dev[0]));
    GROUP_TRY(g, reserve(root, &g->frame, (size_t)strips * strip_bytes));
    GROUP_TRY(g, reserve(root, &g->gather, total));
    const int first_moved = g->transport == rt_group::RCCL_LOOPBACK ? 0 : 1;
    if (g->transport == rt_group::COPY) {
        for (int r = 1; r < n; ++r) {
            if (bytes[r] == 0) continue;
            GROUP_TRY(g, hipSetDevice(g->dev[r]));
            char *dst = static_cast<char *>(g->gather) + offset[r];
            if (g->dev[r] == g->dev[0])
                GROUP_TRY(g, hipMemcpyAsync(dst, g->m[r]->frame_rgba, bytes[r], hipMemcpyDeviceToDevice, g->m[r]->stream));
            else
                GROUP_TRY(g, hipMemcpyPeerAsync(dst, g->dev[0], g->m[r]->frame_rgba, g->dev[r], bytes[r], g->m[r]->stream));
            GROUP_TRY(g, hipEventRecord(g->done[r], g->m[r]->stream));
            GROUP_TRY(g, hipSetDevice(g->dev[0]));
            GROUP_TRY(g, hipStreamWaitEvent(root->stream, g->done[r], 0));
        }
    } else {
        // one grouped exchange: the root posts a receive per member, every member one send
        NCCL_TRY(g, g->rccl.GroupStart());
        for (int r = first_moved; r < n; ++r) {
            if (bytes[r] == 0) continue;
            NCCL_TRY(g, g->rccl.Recv(static_cast<char *>(g->gather) + offset[r], bytes[r], ncclUint8, r, g->comms[0],
                                     root->stream));
            NCCL_TRY(g, g->rccl.Send(g->m[r]->frame_rgba, bytes[r], ncclUint8, 0, g->comms[r], g->m[r]->stream));
        }
        NCCL_TRY(g, g->rccl.GroupEnd());
    }
    GROUP_TRY(g, hipSetDevice(g->dev[0]));
    const int wps = (int)(strip_bytes / 16);
    for (int r = 0; r < n; ++r) {
        const int cnt = (int)(bytes[r] / strip_bytes);
        if (cnt == 0) continue;
        const void *src = r < first_moved ? root->frame_rgba : static_cast<char *>(g->gather) + offset[r];
        const long long words = (long long)cnt * wps;
        hipLaunchKernelGGL(assemble_strips, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, root->stream,
                           static_cast<uint4 *>(g->frame), static_cast<const uint4 *>(src), wps, cnt, r, n);
    }
    GROUP_TRY(g, hipGetLastError());
    // all members' work has been ordered into the root stream (events / RCCL); wait for it, then for
    // the members' own streams, which by then have nothing left but to report
    GROUP_TRY(g, hipStreamSynchronize(root->stream));
    uint64_t rays = 0;
    float kmin = 1e30f, kmax = 0.f;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] == 0) continue;
        GROUP_TRY(g, hipSetDevice(g->dev[i]));
        GROUP_TRY(g, hipStreamSynchronize(g->m[i]->stream));
        uint64_t r = 0;
        GROUP_TRY(g, hipMemcpy(&r, g->m[i]->frame_rays, 8, hipMemcpyDeviceToHost));
        rays += r;
        GROUP_TRY(g, hipEventElapsedTime(&g->m[i]->last_kernel_ms, g->m[i]->ev_start, g->m[i]->ev_stop));
        kmin = fminf(kmin, g->m[i]->last_kernel_ms);
        kmax = fmaxf(kmax, g->m[i]->last_kernel_ms);
    }
    GROUP_TRY(g, hipSetDevice(g->dev[0]));
    if (h_rays) *h_rays = rays;
    if (d_frame) *d_frame = g->frame;
    if (timing) {
        timing->kernel_ms_min = kmin;
        timing->kernel_ms_max = kmax;
        timing->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        timing->gather_ms = timing->total_ms - kmax; // launch, exchange, assembly and the waits around them
    }
    return RT_OK;
}

RT_API int rt_group_render_host(rt_group *g, const rt_params *p, uint8_t *h_rgba, uint64_t *h_rays,
                                rt_group_timing *timing) {
    if (!g) return RT_ERR_INVALID;
    if (!h_rgba) return gfail(g, RT_ERR_INVALID, "h_rgba is null");
    const auto t_begin = std::chrono::steady_clock::now();
    void *d_frame = nullptr;
    int rc = rt_group_render(g, p, &d_frame, h_rays, timing);
    if (rc != RT_OK) return rc;
    GROUP_TRY(g, hipMemcpy(h_rgba, d_frame, (size_t)p->width * (size_t)p->height * 4, hipMemcpyDeviceToHost));
    if (timing) {
        timing->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        timing->gather_ms = timing->total_ms - timing->kernel_ms_max;
    }
    return RT_OK;
}
