// rt_mfma.h — the sphere filter of the scan on the matrix cores (gfx950 v_mfma_f32_32x32x16_f16).
//
// The filter of rt_device.h is a bilinear form in per-sphere and per-ray quantities.  In a frame
// centred on the sphere cloud and scaled by its radius Rs (C' = (C - c0)/Rs, O' = (O - c0)/Rs, the
// unit direction dh unchanged, p = O' - dh (O'.dh) the point of the ray's line nearest the centre):
//
//   disc/Rs^2 = sum_ij C'_i C'_j dh_i dh_j  +  2 C'.p  +  (r^2 - |C - c0|^2)/Rs^2  -  |p|^2
//
// which is the dot product of eleven per-sphere numbers (x^2, y^2, z^2, 2xy, 2xz, 2yz, 2x, 2y, 2z, q, 1)
// with eleven per-ray numbers (dx^2, dy^2, dz^2, dx dy, dx dz, dy dz, px, py, pz, 1, m - |p|^2; m is the
// margin below).  One 32x32x16 MFMA pair (K = 32) evaluates it for 32 spheres x 32 rays in 64 cycles of
// the matrix pipe; the packed-FP32 chain needs 7 instructions of 4.4 cycles per sphere PAIR per 64 rays.
//
// f16 carries 11 significant bits, so every number is split into a high and a low half-float
// (a = a_hi + a_lo, 22 bits) and each product a*b enters the sum as a_hi b_hi + a_hi b_lo + a_lo b_hi:
// 9 x 3 slots for the geometric terms, 2 for q (its partner is the constant 1), 2 for the per-ray term
// (partner 1) = 31 of the 32 k slots.  Both sides are scaled by 2^10 so that the low halves of all
// but negligible numbers are normal f16 values.  What is dropped and rounded is bounded by about
// 2^-19 of sum |a_i b_i| <= 5 (frame units); the accumulation in the matrix pipe adds at most 32 * 2^-24 of that
// sum (K = 32 float additions).  tests/test_filter_bound.py rebuilds both operand sides bit for bit on the CPU and
// checks |representation error| + accumulation bound < m for every pair of three scenes, origins up to sixty cloud
// radii out: the worst pair uses 0.28 of m, the sum of magnitudes reaches 4.6.
// The per-ray term carries the margin m = MF_MARGIN + 2^-20 max|O'_i| (the second part covers the
// rounding of O' itself for origins far from the cloud), so the sign of the result errs only towards
// "candidate": like the packed filter, this one may pass a sphere the ray misses and never drops one it
// hits.  Everything that passes is decided by sphere_root() from the candidate queue, so a frame has
// the same bytes whichever filter ran.
//
// The per-sphere side is NEGATED on the host: the result is negative for a candidate, and the sign
// bits of the 32 accumulator registers of a lane are gathered with one v_alignbit_b32 each.
//
// The rows of the table are the spheres in the leaf order of a balanced k-d partition of the cloud, not in scene
// order: a block of 32 rows is then a compact box of neighbours, fewer blocks hold a candidate for any ray of a
// wave, and fewer have their sign bits gathered.
//
// A table too long for LDS is not walked block by block: a second, short table holds one sphere around each
// block, and a wave runs the filter over those first and then only over the blocks whose bound some ray of it
// meets (scan_mfma_culled).  Both steps are conservative filters in front of the same exact decision.
//
// A sphere much larger than the cloud of the others (a ground sphere) would set Rs and with it the
// resolution of the frame; up to MF_MAX_BIG such spheres are kept out of the table and decided directly for every ray.
#pragma once
#include "rt_device.h"

namespace rt {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

constexpr int MF_BLOCK = 32;             // spheres per table block (rows of one MFMA tile)
constexpr int MF_BLOCK_WORDS = 2 * 64;   // 16-byte words per block: two k-steps x 64 lanes
constexpr int MF_COARSE_WORDS = 3 * 64;  // ... per block of the table of block bounds: a third k-step holds the rows of the test for what lies behind a ray
constexpr int MF_MAX_BIG = 8;            // spheres decided outside the table
constexpr float MF_SCALE = 1024.0f;      // both operand sides are scaled by this
constexpr float MF_MARGIN = 3.0517578125e-5f;     // 2^-15, in frame units (radius of the cloud = 1)
constexpr float MF_MARGIN_FAR = 9.5367431640625e-7f; // 2^-20 per frame unit of max|O'_i|
constexpr float MF_CULL = 1.002f;        // a line farther than this from the centre meets no table sphere
constexpr float MF_NEVER = 60000.0f;     // magnitude that outweighs every other term
constexpr float MF_AHEAD_MARGIN = 0.001953125f; // 2^-9 of the bounds' frame: a bound this little behind a ray's origin still counts as ahead
constexpr float MF_AHEAD_CLAMP = 30.0f;  // origins farther than this before or past the cloud's centre (frame units) are treated as this far
#ifndef RT_MF_STRETCH_MAX
#define RT_MF_STRETCH_MAX 4.0f
#endif
constexpr float MF_STRETCH_MAX = RT_MF_STRETCH_MAX;   // the most a thin axis of the cloud is stretched in the frame of the block bounds

// What the kernel needs to know about the table (part of the launch parameters).
struct MfmaFrame {
    float cx, cy, cz; // c0: centre of the cloud
    float scale;      // MF_SCALE / Rs
    int n_blocks;     // table blocks of 32 rows, padded to whole tiles when the table may be streamed (rows past the list never pass)
    int n_used;       // blocks that hold a sphere: what a scan of the resident table walks
    int n_coarse;     // blocks of the table of block bounds in front of a table too long for LDS (0: none)
    float csx, csy, csz; // the bounds' own frame: MF_SCALE / its radius per axis, a flat cloud's thin axes stretched (scan_mfma_culled)
    int coarse_in_lds; // the table of bounds is resident in LDS (else, too long for that, it is read from memory like the table)
    int n_big;
    int big[MF_MAX_BIG]; // indices of the spheres decided directly
};

// Matrix row (0..31) of a table block that holds the block's i-th sphere.
constexpr int mf_row_of_sphere(int i) { return (i & 3) + 8 * ((i & 15) >> 2) + 4 * (i >> 4); }

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

__device__ __forceinline__ uint32_t pack_rtz(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}
__device__ __forceinline__ float half_lo(uint32_t w) { return (float)__builtin_bit_cast(half2v, w).x; }
__device__ __forceinline__ float half_hi(uint32_t w) { return (float)__builtin_bit_cast(half2v, w).y; }

// The ray side of the product, as the B operand of both k-steps for the two 32-ray halves of the wave.
struct RayOperand {
    half8 b[2][2]; // [ray block: lanes 0-31 / 32-63][k-step]
};

// k slots, two per 32-bit word w (sphere side in brackets, both sides carry MF_SCALE):
//   w = f, f < 9 : (a_hi[f], a_lo[f])          [S_hi[f], S_hi[f]]
//   w = 9..12    : (a_hi[2g], a_hi[2g+1])      [S_lo[2g], S_lo[2g+1]]     g = w - 9
//   w = 13       : (a_hi[8], 1)                [S_lo[8], Q_hi]
//   w = 14       : (1, 0)                      [Q_lo, 0]
//   w = 15       : (t_hi, t_lo)                [1, 1]
// Lane l of a 32x32x16 operand holds k = 8 (l >> 5) + j of its row / column (j = 0..7), so a lane's own
// sixteen words are exchanged between the wave's halves (v_permlane32_swap) into the four operands.
// (`scale`: MF_SCALE over the frame's radius, per axis; `dh`: the ray's unit direction in that frame)
__device__ __forceinline__ RayOperand mfma_ray_operand(V3 o, V3 dh, bool live, float cx, float cy, float cz, V3 scale) {
    // O' scaled by MF_SCALE; p in two steps: the first lands on the line near the centre (any multiple
    // of dh does), the second removes what is left along dh, so p's error does not grow with |O'|^2
    const V3 oc = mk((o.x - cx) * scale.x, (o.y - cy) * scale.y, (o.z - cz) * scale.z);
    const V3 o2 = madd(oc, dh, -dot(oc, dh));
    const V3 p = madd(o2, dh, -dot(o2, dh));
    const float pp = dot(p, p) * (1.0f / MF_SCALE); // MF_SCALE * |p|^2
    const float far = fmaxf(fmaxf(fabsf(oc.x), fabsf(oc.y)), fabsf(oc.z)); // MF_SCALE * max|O'_i|
    float t = fmaf(MF_MARGIN_FAR, far, MF_MARGIN * MF_SCALE) - pp;
    live = live && pp <= MF_CULL * MF_CULL * MF_SCALE;
    const V3 ds = dh * 32.0f; // sqrt(MF_SCALE)
    float a[9] = {ds.x * ds.x, ds.y * ds.y, ds.z * ds.z, ds.x * ds.y, ds.x * ds.z, ds.y * ds.z, p.x, p.y, p.z};
    if (!live) {
#pragma unroll
        for (int f = 0; f < 9; ++f) a[f] = 0.0f;
        t = -MF_NEVER;
    }
    uint32_t w[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) w[9 + g] = pack_rtz(a[2 * g], a[2 * g + 1]);
    w[13] = pack_rtz(a[8], MF_SCALE);
    w[14] = 0x00006400u; // (1024, 0)
#pragma unroll
    for (int f = 0; f < 9; ++f) {
        const uint32_t hw = w[9 + (f >> 1)];
        const float hi = (f & 1) ? half_hi(hw) : half_lo(hw);
        w[f] = pack_rtz(hi, a[f] - hi); // hi converts back exactly; the remainder is exact in f32
    }
    {
        const float hi = half_lo(pack_rtz(t, 0.0f));
        w[15] = pack_rtz(hi, t - hi);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        auto r0 = __builtin_amdgcn_permlane32_swap(w[i], w[4 + i], false, false);
        w[i] = r0[0];
        w[4 + i] = r0[1];
        auto r1 = __builtin_amdgcn_permlane32_swap(w[8 + i], w[12 + i], false, false);
        w[8 + i] = r1[0];
        w[12 + i] = r1[1];
    }
    RayOperand op;
    op.b[0][0] = __builtin_bit_cast(half8, (uint4v){w[0], w[1], w[2], w[3]});
    op.b[1][0] = __builtin_bit_cast(half8, (uint4v){w[4], w[5], w[6], w[7]});
    op.b[0][1] = __builtin_bit_cast(half8, (uint4v){w[8], w[9], w[10], w[11]});
    op.b[1][1] = __builtin_bit_cast(half8, (uint4v){w[12], w[13], w[14], w[15]});
    return op;
}

// One 32x32 result tile: register v of lane l is matrix row (v & 3) + 8 (v >> 2) + 4 (l >> 5) of the table
// block behind (a0, a1) against the ray of lane l & 31 of the ray half behind (b0, b1).  The host lays the
// block's spheres out so that this row holds sphere v + 16 (l >> 5) of the block (mf_row_of_sphere): a
// result's sphere index is then linear in its register number.
__device__ __forceinline__ float16v mfma_tile(half8 a0, half8 a1, half8 b0, half8 b1) {
    const float16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, zero, 0, 0, 0), 0, 0, 0);
}
// ... added to `c`
__device__ __forceinline__ float16v mfma_tile_onto(half8 a0, half8 a1, half8 b0, half8 b1, const float16v &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0), 0, 0, 0);
}

// Sign bits of result tiles.  On this chip every vector instruction holds the
// issue port for about four cycles, so what counts is their number: gathering the signs takes one
// v_alignbit_b32 per register, OR-ing the registers half as many instructions (v_or3_b32), and a ray
// passes the filter for about one sphere in three hundred: more than half of the blocks pass for no
// lane of the wave and are never gathered.
__device__ __forceinline__ uint32_t or16(const float16v &acc) {
    uint32_t any = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) any |= __float_as_uint(acc[v]);
    return any;
}
// Sign bits of the two result tiles of a block: bit 2 v + h = register v of ray half h.
__device__ __forceinline__ uint32_t gather32(const float16v &acc0, const float16v &acc1) {
    uint32_t lo = 0, hi = 0; // two chains: dependent instructions are not back to back
#pragma unroll
    for (int v = 7; v >= 0; --v) {
        lo = __builtin_amdgcn_alignbit(lo, __float_as_uint(acc1[v]), 31);
        hi = __builtin_amdgcn_alignbit(hi, __float_as_uint(acc1[8 + v]), 31);
        lo = __builtin_amdgcn_alignbit(lo, __float_as_uint(acc0[v]), 31);
        hi = __builtin_amdgcn_alignbit(hi, __float_as_uint(acc0[8 + v]), 31);
    }
    return lo | (hi << 16);
}

// Filter of `nblk` table blocks at `tab_lane` (LDS, this lane's words; block b of the run is table block
// blk0 + b) against the wave's rays; what passes goes through the queue, numbered by table row (`sph` here is
// the sphere of every row, cq.row_idx its index in the scene).  The run is followed by one more
// readable block (its content does not matter).
// Software pipeline: the matrix pipe works on one ray half's tile while the other's is examined, and the
// table words of the next block are on their way from LDS meanwhile.
// FLUSH_AT: queue fill from which batches are decided inside the loop (a turn of the recording loop adds up to 64
// entries, so the queue holds FLUSH_AT + 64).
template <int FLUSH_AT = 64>
__device__ __forceinline__ void mfma_scan_blocks(const uint4v *tab_lane, int blk0, int nblk, const RayOperand &op,
                                                 const CandQ &cq, int &count, int lane, V3 o, V3 dh, float tmin,
                                                 int self_id, const float4 *__restrict__ sph, SecClock &sec) {
    // queue entry of the result in register v of ray half h: owner lane | sphere << 6, which is this plus
    // (2 v + h) << 5 plus the block number << 11
    const unsigned lane_part = (unsigned)(lane & 31) | ((unsigned)(lane >> 5) << 10);
    half8 a0 = __builtin_bit_cast(half8, tab_lane[0]), a1 = __builtin_bit_cast(half8, tab_lane[64]);
    float16v acc0 = mfma_tile(a0, a1, op.b[0][0], op.b[0][1]);
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
        const float16v acc1 = mfma_tile(a0, a1, op.b[1][0], op.b[1][1]);
        a0 = __builtin_bit_cast(half8, tab_lane[(2 * blk + 2) * 64]);
        a1 = __builtin_bit_cast(half8, tab_lane[(2 * blk + 3) * 64]);
        // one test per block: does any lane have a negative among its 32 results?  (Measured: tests
        // per group of eight registers save instructions and lose as much to their branches.)
#ifdef RT_MF_PROBE_NO_CANDIDATES // timing probe: the products and their examination alone (frames are wrong)
        const bool some = __ballot((int)(or16(acc0) | or16(acc1)) == 0x7fffffff) != 0;
#else
        const bool some = __ballot((int)(or16(acc0) | or16(acc1)) < 0) != 0;
#endif
        uint32_t m = 0;
        if (some) m = gather32(acc0, acc1);
        // acc0's registers go to the next block's products only now that its signs are taken
        acc0 = mfma_tile(a0, a1, op.b[0][0], op.b[0][1]);
        if (!some) continue;
        RT_STAT(2); // tuning build: blocks whose sign bits were gathered (per wave)
        sec.mark(9);
        const unsigned block_part = lane_part | ((unsigned)(blk0 + blk) << 11);
        // one (ray, sphere) pair per lane and turn, lowest register first
        for (unsigned long long any = __ballot(m != 0); any != 0; any = __ballot(m != 0)) {
            RT_STAT(3); // turns of the recording loop (per wave)
            if (m != 0) {
                RT_STAT_LANES(5); // pairs recorded
                const unsigned bit = (unsigned)__builtin_ctz(m);
                m &= m - 1;
                cq.ent[count + lane_rank(any)] = block_part + (bit << 5);
            }
            count += (int)__popcll(any);
            if (count >= FLUSH_AT) {
                sec.mark(10);
                do resolve_batch(cq, count, 64, lane, o, dh, tmin, self_id, sph);
                while (count >= 64);
                sec.mark(9);
            }
        }
        sec.mark(8);
    }
}

// What is left in the queue, then the few spheres outside the table, then the winner.
template <class F>
__device__ __forceinline__ void mfma_scan_finish(const F &fr, const CandQ &cq, int &count, int lane, V3 o, V3 dh, bool live,
                                                 float tmin, int self_id, const float4 *__restrict__ sph,
                                                 const float4 *__restrict__ rows, float &best_t, int &best_id,
                                                 SecClock &sec) {
    sec.mark(10);
    while (count > 0) resolve_batch(cq, count, min(count, 64), lane, o, dh, tmin, self_id, rows);
    sec.mark(11);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned long long key = cq.slot[lane];
    const cptr<float4> csph = as_const(sph);
    for (int i = 0; i < fr.n_big; ++i) {
        const int s = fr.big[i];
        const float4 sp = load_const4(csph, s); // same address in every lane: a scalar load
        const float cand = sphere_root(o, dh, sp.x, sp.y, sp.z, sp.w * sp.w, tmin);
        if (cand > tmin && cand < __builtin_inff() && s != self_id) {
            const unsigned long long k = ((unsigned long long)__float_as_uint(cand) << 32) | (unsigned long long)(unsigned)s;
            key = k < key ? k : key;
        }
    }
    if (live && key != NO_HIT_KEY) {
        best_t = __uint_as_float((unsigned)(key >> 32));
        best_id = (int)(unsigned)key;
    }
}

// Closest sphere hit of every lane's ray: the MFMA filter over the table resident in LDS, candidates
// through the wave's queue (rt_device.h), the few spheres outside the table directly.  Whole wave; a
// lane without a ray passes live = false.
template <class F> // MfmaFrame, in whatever address space the launch parameters live
__device__ __forceinline__ void scan_mfma(const uint4v *tab, const F &fr, V3 o, V3 dh, bool live, float tmin,
                                          int self_id, int lane, const CandQ &cq, const float4 *__restrict__ sph,
                                          const float4 *__restrict__ rows, float &best_t, int &best_id, SecClock &sec) {
    const RayOperand op = mfma_ray_operand(o, dh, live, fr.cx, fr.cy, fr.cz, mk(fr.scale, fr.scale, fr.scale));
    cq.slot[lane] = NO_HIT_KEY;
    int count = 0; // queue fill, wave-uniform
    sec.mark(8); // tuning build: 8 = matrix products and sign masks, 9 = recording, 10 = decisions, 11 = the rest
    mfma_scan_blocks(tab + lane, 0, fr.n_used, op, cq, count, lane, o, dh, tmin, self_id, rows, sec);
    mfma_scan_finish(fr, cq, count, lane, o, dh, live, tmin, self_id, sph, rows, best_t, best_id, sec);
}

// The ray side of the test for what lies behind a ray (scan_mfma_culled): one 32x32x8 product.  With U, P a bound's
// centre and radius and o, d the ray's origin and unit direction, all in the bounds' frame, some point of the bound
// lies ahead of the origin iff U.d + P - o.d > 0.  Eight k slots (sphere side in brackets; every product carries
// 2^20 = MF_SCALE^2 like the filter's values: 1024 x 1024, 32 x 32768, 32 x 32768):
//   dx [Ux]   dy [Uy]   dz [Uz]   1 [P]   t_hi [1]   t_lo [1]   0   0
// with t = margin - o.d, clamped to +-MF_AHEAD_CLAMP (an origin farther before the cloud has all of it ahead, one
// farther past it all of it behind, clamped or not).  Single half-floats: U, P rounded to nearest (2^-12 each) and d
// towards zero (2^-11) put the sum off by less than 2^-11 * 1.5 |U| |d| + 2^-12 P <= 0.001 frame units, half of
// MF_AHEAD_MARGIN.  The sphere side is negated: the result is negative for a bound that lies, at least in part, ahead,
// and 2^20 times how far it lies behind otherwise.  That number, or zero, is what the filter's products for the bound
// are added to: a bound the line meets (a negative filter value, at most about 2^20 P^2 in size) stays wanted unless it
// lies behind the origin by more than the filter value says the line is inside it -- a few thousandths of the frame.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
struct AheadOperand {
    half4 b[2]; // [ray block: lanes 0-31 / 32-63]
};
__device__ __forceinline__ AheadOperand mfma_ahead_operand(V3 o, V3 dh, bool live, float cx, float cy, float cz, V3 scale) {
    const float inv = 1.0f / MF_SCALE;
    const V3 ou = mk((o.x - cx) * scale.x * inv, (o.y - cy) * scale.y * inv, (o.z - cz) * scale.z * inv); // frame units
    float t = fminf(fmaxf(MF_AHEAD_MARGIN - dot(ou, dh), -MF_AHEAD_CLAMP), MF_AHEAD_CLAMP) * 32.0f;
    V3 ds = dh * MF_SCALE;
    float one = 32.0f;
    if (!(live && t == t)) { // no ray (or one that is not a number): nothing lies ahead of it
        ds = mk(0, 0, 0);
        one = 0.0f;
        t = -1.0f;
    }
    const float th = half_lo(pack_rtz(t, 0.0f));
    uint32_t w[4] = {pack_rtz(ds.x, ds.y), pack_rtz(ds.z, one), pack_rtz(th, t - th), 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        auto r0 = __builtin_amdgcn_permlane32_swap(w[i], w[2 + i], false, false);
        w[i] = r0[0];
        w[2 + i] = r0[1];
    }
    AheadOperand op;
    op.b[0] = __builtin_bit_cast(half4, (uint2v){w[0], w[1]});
    op.b[1] = __builtin_bit_cast(half4, (uint2v){w[2], w[3]});
    return op;
}

// Which rows of a block of bounds are wanted by ANY ray of one half of the wave's rays: a negative value.  Bit b = row b
// of the block (register v of lanes 0-31 is row v, of lanes 32-63 row 16 + v, see mfma_tile).  Wave-uniform.
__device__ __forceinline__ uint32_t rows_wanted(const float16v &m) {
    // the lane's sixteen sign bits (bit v = register v), one v_alignbit_b32 each ...
    uint32_t mask = 0;
#pragma unroll
    for (int v = 15; v >= 0; --v) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(m[v]), 31);
    // ... OR-ed over the 32 lanes of each ray half: within rows of sixteen lanes by shifts of 1, 2, 4, 8 (a lane beyond the
    // row's start contributes 0), then row 0's last lane into row 1 and row 2's into row 3; lanes 31 and 63 hold the halves
    mask |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, 0x111, 0xf, 0xf, true); // row_shr:1
    mask |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, 0x112, 0xf, 0xf, true); // row_shr:2
    mask |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, 0x114, 0xf, 0xf, true); // row_shr:4
    mask |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, 0x118, 0xf, 0xf, true); // row_shr:8
    mask |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, 0x142, 0xa, 0xf, true); // row_bcast:15 into rows 1 and 3
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)mask, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)mask, 63);
    return lo | (hi << 16);
}
// How far a bound lies behind a ray's origin, or zero: what the filter's products for the bound are added to.
__device__ __forceinline__ float16v behind_or_zero(float16v a) {
#pragma unroll
#ifdef RT_MF_PROBE_NO_AHEAD // tuning probe: nothing counts as lying behind
    for (int v = 0; v < 16; ++v) a[v] = 0.0f;
#else
    // (as integers: a negative float is a negative integer, a positive one keeps its bits; one instruction, where fmaxf
    // costs two for the sake of signalling NaNs -- a NaN only comes from a lane without a ray, which wants nothing)
    for (int v = 0; v < 16; ++v) a[v] = __int_as_float(max(__float_as_int(a[v]), 0));
#endif
    return a;
}

// The same for a table too long for LDS, with the table of block bounds (`ctab`, resident in LDS, fr.n_coarse blocks
// of MF_COARSE_WORDS and one more readable one) in front of it.  First the filter runs over the bounds, 32 blocks per
// coarse block, and notes per coarse block which of its 32 blocks are wanted: those whose bound passes for some ray of
// the wave and does not lie behind that ray's origin (`wanted`: fr.n_coarse words of this wave in LDS).  Then it runs over the wanted blocks of the table in one
// loop, their words read from memory (`gtab`) one block ahead of the products, candidates through the queue as in
// mfma_scan_blocks.  A block whose bound no ray's line meets holds no sphere any of the rays can hit, so the winner
// is the one scan_mfma finds.  Whole wave, no workgroup step: the waves of a workgroup scan independently.
// `visited` counts the blocks walked, those of the table of bounds included.
// A table of bounds too long for LDS itself (scenes of more than about 28 000 spheres) is read from memory (`gcoarse`), one
// block ahead of its products, like the table.
template <int FLUSH_AT = 64, class F>
__device__ __forceinline__ void scan_mfma_culled(const uint4v *ctab, const uint4v *__restrict__ gcoarse, const uint4v *__restrict__ gtab, uint32_t *wanted, const F &fr,
                                                 V3 o, V3 dh, bool live, float tmin, int self_id, int lane, const CandQ &cq,
                                                 const float4 *__restrict__ sph, const float4 *__restrict__ rows,
                                                 float &best_t, int &best_id, unsigned &visited, SecClock &sec) {
    cq.slot[lane] = NO_HIT_KEY;
    int count = 0;
    sec.mark(8);
    const int n_coarse = fr.n_coarse;
    {
        // The bounds live in a frame of their own: the axes along which the cloud is thin are stretched (a flat field of
        // spheres would otherwise get bounds that reach as far above it as along it), and a line stays a line.  The ray's
        // direction in that frame is its direction times the axis scales, made a unit vector again.
        const V3 cs = mk(fr.csx, fr.csy, fr.csz);
        const V3 ds = mk(dh.x * cs.x, dh.y * cs.y, dh.z * cs.z);
        const V3 dc = ds * __frsqrt_rn(dot(ds, ds));
        const RayOperand op = mfma_ray_operand(o, dc, live, fr.cx, fr.cy, fr.cz, cs);
        const AheadOperand ah = mfma_ahead_operand(o, dc, live, fr.cx, fr.cy, fr.cz, cs);
        const float16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // one block of 32 bounds: which of its blocks are wanted (one ray half at a time: this loop has no registers for a
        // second result tile beside the two operands and the path's state)
        auto bounds_block = [&](int cb, half8 c0, half8 c1, half4 c2) {
            // (a half of the rays that wants none of the 32 bounds -- no negative value in any lane -- skips the gather of the signs)
            uint32_t w = 0;
            {
                const float16v t = mfma_tile_onto(c0, c1, op.b[0][0], op.b[0][1], behind_or_zero(__builtin_amdgcn_mfma_f32_32x32x8f16(c2, ah.b[0], zero, 0, 0, 0)));
                if (__ballot((int)or16(t) < 0) != 0) w = rows_wanted(t);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float16v t = mfma_tile_onto(c0, c1, op.b[1][0], op.b[1][1], behind_or_zero(__builtin_amdgcn_mfma_f32_32x32x8f16(c2, ah.b[1], zero, 0, 0, 0)));
                if (__ballot((int)or16(t) < 0) != 0) w |= rows_wanted(t);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef RT_MF_PROBE_WANT_ALL // tuning probe: every block is walked (the words past the table's end are rows that never pass)
            w = cb * MF_BLOCK + MF_BLOCK <= fr.n_used ? ~0u : (1u << (fr.n_used - cb * MF_BLOCK)) - 1u;
#endif
            if (lane == 0) wanted[cb] = w;
        };
        if (fr.coarse_in_lds) {
            const uint4v *clane = ctab + lane;
#pragma unroll 1
            for (int cb = 0; cb < n_coarse; ++cb) // (no read-ahead from LDS: no registers for a second set of words either)
                bounds_block(cb, __builtin_bit_cast(half8, clane[(3 * cb) * 64]), __builtin_bit_cast(half8, clane[(3 * cb + 1) * 64]),
                             *reinterpret_cast<const half4 *>(clane + (3 * cb + 2) * 64)); // (the first 8 bytes of the lane's third word)
        } else {
            const __amdgpu_buffer_rsrc_t ctb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4v *>(gcoarse), 0, (n_coarse + 1) * MF_COARSE_WORDS * 16, 0x00020000);
            const int lane_bytes = lane * 16;
            auto words = [&](int cb, half8 &c0, half8 &c1, half4 &c2) {
                c0 = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(ctb, lane_bytes, cb * (MF_COARSE_WORDS * 16), 0));
                c1 = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(ctb, lane_bytes, cb * (MF_COARSE_WORDS * 16) + 1024, 0));
                c2 = __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(ctb, lane_bytes, cb * (MF_COARSE_WORDS * 16) + 2048, 0));
            };
            half8 n0, n1;
            half4 n2;
            words(0, n0, n1, n2);
#pragma unroll 1
            for (int cb = 0; cb < n_coarse; ++cb) {
                const half8 c0 = n0, c1 = n1;
                const half4 c2 = n2;
                words(cb + 1, n0, n1, n2); // (one more block is readable behind the last)
                bounds_block(cb, c0, c1, c2);
            }
        }
        visited += (unsigned)n_coarse;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // (the table's own ray operand is formed only now: sixteen registers the loop above has no room for)
    asm volatile("" : "+v"(o.x), "+v"(dh.x));
    const RayOperand op = mfma_ray_operand(o, dh, live, fr.cx, fr.cy, fr.cz, mk(fr.scale, fr.scale, fr.scale));
    // the wanted blocks in ascending order, -1 after the last (wave-uniform)
    int next_word = 0, word_base = 0;
    uint32_t word = 0;
    auto next_block = [&]() -> int {
        while (word == 0) {
            if (next_word >= n_coarse) return -1;
            word = (uint32_t)__builtin_amdgcn_readfirstlane((int)wanted[next_word]);
            word_base = next_word * MF_BLOCK;
            ++next_word;
        }
        const int b = word_base + (int)__builtin_ctz(word);
        word &= word - 1;
        return b;
    };
    // the table as a buffer: a block's words are at the block's (scalar) byte offset plus this lane's 16 bytes, so a
    // load costs no address arithmetic in vector registers
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4v *>(gtab), 0, (fr.n_blocks + 1) * MF_BLOCK_WORDS * 16, 0x00020000);
    const int lane_bytes = lane * 16;
    auto block_words = [&](int b, half8 &w0, half8 &w1) {
        w0 = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(tab, lane_bytes, b * (MF_BLOCK_WORDS * 16), 0));
        w1 = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(tab, lane_bytes, b * (MF_BLOCK_WORDS * 16) + 1024, 0));
    };
    int blk = next_block();
    if (blk >= 0) {
        half8 a0, a1;
        block_words(blk, a0, a1);
        int nxt = next_block();
        float16v acc0 = mfma_tile(a0, a1, op.b[0][0], op.b[0][1]);
#pragma unroll 1
        for (;;) {
            ++visited;
            const float16v acc1 = mfma_tile(a0, a1, op.b[1][0], op.b[1][1]);
            int nxt_s = nxt;
            asm("" : "+s"(nxt_s)); // wave-uniform, and tested on the scalar unit
            const bool more = nxt_s >= 0;
            if (more) block_words(nxt, a0, a1);
            const bool some = __ballot((int)(or16(acc0) | or16(acc1)) < 0) != 0;
            uint32_t m = 0;
            if (some) m = gather32(acc0, acc1);
            if (more) acc0 = mfma_tile(a0, a1, op.b[0][0], op.b[0][1]);
            if (some) {
                RT_STAT(2);
                sec.mark(9);
                // (the lane's part of a queue entry is worked out here, where it is needed, not kept in a register
                // across the scan: this kernel has none to spare)
                unsigned ln = (unsigned)lane;
                asm volatile("" : "+v"(ln));
                const unsigned block_part = (ln & 31u) | ((ln >> 5) << 10) | ((unsigned)blk << 11);
                for (unsigned long long any = __ballot(m != 0); any != 0; any = __ballot(m != 0)) {
                    RT_STAT(3);
                    if (m != 0) {
                        RT_STAT_LANES(5);
                        const unsigned bit = (unsigned)__builtin_ctz(m);
                        m &= m - 1;
                        cq.ent[count + lane_rank(any)] = block_part + (bit << 5);
                    }
                    count += (int)__popcll(any);
                    if (count >= FLUSH_AT) {
                        sec.mark(10);
                        do resolve_batch(cq, count, 64, lane, o, dh, tmin, self_id, rows);
                        while (count >= 64);
                        sec.mark(9);
                    }
                }
                sec.mark(8);
            }
            if (!more) break;
            blk = nxt;
            nxt = next_block();
        }
    }
    mfma_scan_finish(fr, cq, count, lane, o, dh, live, tmin, self_id, sph, rows, best_t, best_id, sec);
}

// The same with a table too large for LDS: it passes through two LDS buffers of MF_TILE_BLOCKS blocks,
// the whole workgroup scanning in step (one barrier per tile; every wave of the workgroup must call this,
// with or without rays).  Before a wave works on the current tile its threads send their share of the next
// one from memory straight into the other buffer (global_load_lds_dwordx4: no registers, no LDS stores), so
// the fetch costs no waiting.
// (20 blocks: the longest tiles two buffers of which fit beside sixteen waves' state; a scan of 10 001 spheres then
// crosses 16 barriers instead of 20, 1.6 % of its frame)
#ifndef RT_MF_TILE_BLOCKS
#define RT_MF_TILE_BLOCKS 20
#endif
constexpr int MF_TILE_BLOCKS = RT_MF_TILE_BLOCKS;
constexpr int MF_TILE_WORDS = (MF_TILE_BLOCKS + 1) * MF_BLOCK_WORDS; // with the block the read-ahead touches
#ifndef RT_MF_GLDS
#define RT_MF_GLDS 1 // the next tile goes from memory straight into LDS (global_load_lds_dwordx4); 0: through registers
#endif
template <int WG_THREADS_T, int QUEUE_CAP, class F>
__device__ __forceinline__ void scan_mfma_streamed(uint4v *bufs, const uint4v *__restrict__ gtab, const F &fr, V3 o, V3 dh,
                                                   bool live, float tmin, int self_id, int lane, const CandQ &cq,
                                                   const float4 *__restrict__ sph, const float4 *__restrict__ rows,
                                                   float &best_t, int &best_id, SecClock &sec) {
    constexpr int TILE = MF_TILE_BLOCKS * MF_BLOCK_WORDS;                  // 16-byte words of a tile
    constexpr int PER_THREAD = (TILE + WG_THREADS_T - 1) / WG_THREADS_T;   // ... per thread
    constexpr bool EVEN = PER_THREAD * WG_THREADS_T == TILE;               // else the last turn is for some threads only
    const RayOperand op = mfma_ray_operand(o, dh, live, fr.cx, fr.cy, fr.cz, mk(fr.scale, fr.scale, fr.scale));
    cq.slot[lane] = NO_HIT_KEY;
    int count = 0;
    const int n_tiles = fr.n_blocks / MF_TILE_BLOCKS; // the table is padded to whole tiles
    const int tid = threadIdx.x;
#if RT_MF_GLDS
    // Tile t of the table into buffer `buf`, without passing through registers: every lane names its own 16 bytes of
    // the table, and a wave's 1 KiB lands at the wave's (uniform) LDS address + 16 x lane, which is where the
    // lane-linear copy of the tile wants them.  (tid + k * WG_THREADS_T < TILE is the same for all lanes of a wave.)
    auto fetch = [&](int t, int buf) {
        const uint4v *g = gtab + (size_t)t * TILE;
        uint4v *dst = bufs + buf * MF_TILE_WORDS;
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k)
            if (EVEN || tid + k * WG_THREADS_T < TILE)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(g + tid + k * WG_THREADS_T),
                    (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)(dst + tid + k * WG_THREADS_T), 16, 0, 0);
    };
    // the previous scan's readers of buffer 0 are done: every wave has passed the last barrier of that scan
    fetch(0, 0);
    sec.mark(12); // tuning build: 12 = first tile and its barrier, 13 = the barriers between tiles
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this thread's part of the tile has landed
    __syncthreads();
    sec.mark(8);
    for (int t = 0; t < n_tiles; ++t) {
        // buffer (t + 1) & 1 held tile t - 1, which every wave has finished with (the barrier that ended it)
        if (t + 1 < n_tiles) fetch(t + 1, (t + 1) & 1);
        mfma_scan_blocks<QUEUE_CAP - 64>(bufs + (t & 1) * MF_TILE_WORDS + lane, t * MF_TILE_BLOCKS, MF_TILE_BLOCKS, op, cq, count,
                                         lane, o, dh, tmin, self_id, rows, sec);
        sec.mark(13);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
        __syncthreads(); // the next tile is complete, and nobody still reads the buffer after it
        sec.mark(8);
    }
#else
    uint4v next[PER_THREAD];
    // the previous scan's readers of buffer 0 are done: every wave has passed the last barrier of that scan
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k)
        if (EVEN || tid + k * WG_THREADS_T < TILE) bufs[tid + k * WG_THREADS_T] = gtab[tid + k * WG_THREADS_T];
    sec.mark(12); // tuning build: 12 = first tile and its barrier, 13 = the barriers between tiles
    __syncthreads();
    sec.mark(8);
    for (int t = 0; t < n_tiles; ++t) {
        if (t + 1 < n_tiles) {
            const uint4v *g = gtab + (size_t)(t + 1) * MF_TILE_BLOCKS * MF_BLOCK_WORDS;
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k)
                if (EVEN || tid + k * WG_THREADS_T < TILE) next[k] = g[tid + k * WG_THREADS_T];
        }
        mfma_scan_blocks<QUEUE_CAP - 64>(bufs + (t & 1) * MF_TILE_WORDS + lane, t * MF_TILE_BLOCKS, MF_TILE_BLOCKS, op, cq, count,
                                         lane, o, dh, tmin, self_id, rows, sec);
        if (t + 1 < n_tiles) {
            uint4v *dst = bufs + ((t + 1) & 1) * MF_TILE_WORDS;
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k)
                if (EVEN || tid + k * WG_THREADS_T < TILE) dst[tid + k * WG_THREADS_T] = next[k];
        }
        sec.mark(13);
        __syncthreads(); // the next tile is complete, and nobody still reads the buffer after it
        sec.mark(8);
    }
#endif
    mfma_scan_finish(fr, cq, count, lane, o, dh, live, tmin, self_id, sph, rows, best_t, best_id, sec);
}

#endif // device code

} // namespace rt
