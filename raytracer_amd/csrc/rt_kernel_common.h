// rt_kernel_common.h -- launch parameters, device scene view, radiance sums in fixed point and the two
// plain sphere scans (LDS broadcast reads, scalar loads) shared by the kernels of rt_amd.hip.
#pragma once
#include "../../include/rt_amd.h"
#include "rt_device.h"
#include "rt_prims.h"
#include "rt_mfma.h"

using namespace rt;

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
enum { SRC_LDS_RESIDENT = 0, SRC_LDS_STREAMED = 1, SRC_SCALAR = 2, SRC_MFMA = 3, SRC_MFMA_STREAMED = 4, SRC_MFMA_CULLED = 5 };
constexpr bool is_mfma(int src) { return src == SRC_MFMA || src == SRC_MFMA_STREAMED || src == SRC_MFMA_CULLED; }

// The matrix-core variant shares one filter table per workgroup, so its workgroups are as large as
// they come (16 waves: one workgroup per CU at four waves per SIMD), and its queue only ever holds
// the remainder of one batch plus one turn of the recording loop.
constexpr size_t MF_LDS_BYTES = 159 * 1024; // dynamic LDS a workgroup of the matrix-core variants may use: a CU has 160 KiB, and the workgroup-wide vote of the streamed variant keeps a little static LDS
#ifndef RT_MF_THREADS
#define RT_MF_THREADS 1024
#endif
#ifndef RT_MF_STREAM_THREADS
#define RT_MF_STREAM_THREADS 1024
#endif
// The general kernel (lights, other primitive kinds, the halo) carries more state per path: its matrix-core
// variant gets three waves per SIMD (170 registers each) in a workgroup of twelve.
#ifndef RT_MF_FULL_THREADS
#define RT_MF_FULL_THREADS 768
#endif
constexpr int wg_threads_of(int src, bool full = false) {
    return src == SRC_MFMA || src == SRC_MFMA_CULLED ? (full ? RT_MF_FULL_THREADS : RT_MF_THREADS)
           : src == SRC_MFMA_STREAMED ? (full ? RT_MF_FULL_THREADS : RT_MF_STREAM_THREADS) : WG_THREADS;
}
// (the streamed variant's waves scan in step, a barrier per table tile, so a wave that stops to decide a batch holds
// up its workgroup; a queue long enough to leave nearly all decisions to the end of the scan (448 entries) was
// measured on the 10 001-sphere frame and changed nothing: 3.78 s either way)
#ifndef RT_MF_STREAM_QUEUE
#define RT_MF_STREAM_QUEUE 128
#endif
constexpr int cand_cap_of(int src) { return src == SRC_MFMA_STREAMED ? RT_MF_STREAM_QUEUE : is_mfma(src) ? 128 : CAND_CAP; }
constexpr size_t acc_bytes_of(int src, bool full = false) {
    return (size_t)(wg_threads_of(src, full) / 64) * WAVE_LDS_WORDS * sizeof(unsigned long long);
}
// the variant that culls table blocks by their bounds notes per wave which blocks it wants: one word per 32 blocks
constexpr int MF_COARSE_MAX = 256; // (262 144 spheres; longer tables are streamed whole)
constexpr size_t wave_state_bytes_of(int src, bool full = false) {
    return acc_bytes_of(src, full) +
           (size_t)(wg_threads_of(src, full) / 64) * (cand_cap_of(src) * sizeof(unsigned) + 64 * sizeof(unsigned long long) +
                                                      (src == SRC_MFMA_CULLED ? MF_COARSE_MAX * sizeof(uint32_t) : 0));
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
    int prim_min;    // primary-ray lists: 0 = none, every ray takes the scan; general kernels: a scan runs when at least this many lanes need one
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
    const int *mf_row_idx;  // ... and its index in `sph` (rows are in the leaf order of a k-d partition)
    const uint4v *mf_coarse; // the table of block bounds in front of a table too long for LDS (MfmaFrame::n_coarse + 1 blocks)
    FullScene full; // general primitive list (used by the FULL kernel variants only)
};

// Everything a render launch is given, as the kernel's one argument.
struct KArgs {
    KParams p;
    DeviceScene sc;
    uchar4 *rgba;
    float4 *accum;                // may be null
    unsigned long long *rays_out; // may be null
    unsigned long long *scans_out; // may be null: receives the number of scans the waves ran (64 lanes each) and, in the next
                                   // word, the table blocks (32 rows x 64 lanes) they walked when blocks are culled by their bounds
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

} // namespace
