// Probe: a packed FP32 instruction returning wrong results in lanes 48-63 of a wave on gfx950 (MI355X).
// Found in the render kernel (the side planes of a tile's pyramid came out wrong about twice in 10^5 tile openings, only in
// lanes 48-63; tools/dbg_diff_tiles.py); the instruction was located by patching lane-by-lane checks into that kernel's
// assembly:  v_pk_fma_f32 v[d:d+1], s[a:a+1], v[b:b+1], v[c:c+1] op_sel:[1,0,1] op_sel_hi:[0,0,0].
// Here the instruction runs alone in a loop in some waves while other waves of the workgroup run matrix products and/or
// LDS atomics; every lane checks both results every time.  What the runs show (profiles/r03/hazard_pk_probe.txt):
//  * the LOW result comes out without its addend -- the addend that op_sel takes from the HIGH half of the third source
//    reads as 0 -- in lanes 48-63 only, thousands to a million times per run;
//  * only while other waves issue MFMAs (32x32x16 or 16x16x32); never beside LDS atomics or plain FMAs, never alone;
//  * only when the instruction also has an SGPR pair among its sources (first or second) AND op_sel re-routes the third
//    source; with VGPR sources throughout, with the SGPR pair in the plain pairing, with only the SGPR pair's halves swapped,
//    as two v_fma_f32 or as v_pk_mul_f32 it never happened.
// build: hipcc --offload-arch=gfx950 -O2 -o hazard_pk tools/hazard_pk.hip ; run: ./hazard_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// FORM 0: SGPR pair, op_sel swapping the halves of sources 0 and 2 (the compiler's form); 1: the same, all sources VGPRs;
// 2: SGPR pair, no op_sel; 3: two plain v_fma_f32 computing the same; 4: v_pk_mul_f32 with the SGPR pair and op_sel;
// 5: only the SGPR pair's halves swapped; 6: only the third source re-routed (a form the compiler also made); 7: the SGPR
// pair as the second source, form 0 otherwise; 8: only the third source's halves swapped
// BUSY 1: the other waves run v_mfma_f32_32x32x16_f16; 2: LDS atomics; 3: both; 4: plain FP32 multiply-adds; 8: v_mfma_f32_16x16x32_f16
template <int FORM, int BUSY>
__global__ void probe(unsigned long long *out, const float *in, int iters, int busy_mask) {
    const int lane = threadIdx.x & 63;
    __shared__ unsigned long long lds[64];
    const int wave = (int)(threadIdx.x >> 6);
    if ((busy_mask >> (wave & 3)) & 1) {
        typedef float f16v __attribute__((ext_vector_type(16)));
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        f16v acc = {0};
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v acc4 = {0};
        h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 2, 2, 2, 2};
        float f = in[lane];
        for (int i = 0; i < iters / 4; ++i) {
            if (BUSY & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            if (BUSY & 8) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4, 0, 0, 0);
            if (BUSY & 2) atomicMin(&lds[(lane * 7 + i) & 63], (unsigned long long)i);
            if (BUSY & 4) f = fmaf(f, 1.0001f, 0.5f);
        }
        if (acc[0] == 12345.678f || acc4[0] == 12345.678f || f == 12345.678f) out[0] = 1; // (never)
        return;
    }
    unsigned bad = 0;
    const float ry = in[1], bz = in[2], by = in[3], rz = in[4]; // 2, 3, 4, 5
    unsigned long long first = 0;
    for (int i = 0; i < iters; ++i) {
        float lo, hi;
        const float fx = 0.001f * (float)(i & 1023);
#define HEAD "s_mov_b32 s20, %3\n\ts_mov_b32 s21, %6\n\tv_mov_b32 v100, %2\n\tv_mov_b32 v101, 288.0\n\tv_mov_b32 v104, %4\n\tv_mov_b32 v105, %5\n\tv_mov_b32 v106, %3\n\tv_mov_b32 v107, %6\n\ts_nop 7\n\t"
#define TAIL "\n\ts_nop 7\n\tv_mov_b32 %0, v102\n\tv_mov_b32 %1, v103"
#define OPS : "=v"(lo), "=v"(hi) : "v"(fx), "s"(ry), "v"(by), "v"(bz), "s"(rz) : "s20", "s21", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"
        if (FORM == 0) asm volatile(HEAD "v_pk_fma_f32 v[102:103], s[20:21], v[100:101], v[104:105] op_sel:[1,0,1] op_sel_hi:[0,0,0]" TAIL OPS);
        else if (FORM == 1) asm volatile(HEAD "v_pk_fma_f32 v[102:103], v[106:107], v[100:101], v[104:105] op_sel:[1,0,1] op_sel_hi:[0,0,0]" TAIL OPS);
        else if (FORM == 2) asm volatile(HEAD "v_pk_fma_f32 v[102:103], s[20:21], v[100:101], v[104:105]" TAIL OPS);
        else if (FORM == 3) asm volatile(HEAD "v_fma_f32 v102, s21, v100, v105\n\tv_fma_f32 v103, s20, v100, v104" TAIL OPS);
        else if (FORM == 4) asm volatile(HEAD "v_pk_mul_f32 v[102:103], s[20:21], v[100:101] op_sel:[1,0] op_sel_hi:[0,0]" TAIL OPS);
        else if (FORM == 5) asm volatile(HEAD "v_pk_fma_f32 v[102:103], s[20:21], v[100:101], v[104:105] op_sel:[1,0,0] op_sel_hi:[0,1,1]" TAIL OPS);
        else if (FORM == 6) asm volatile(HEAD "v_pk_fma_f32 v[102:103], s[20:21], v[100:101], v[104:105] op_sel:[0,0,1] op_sel_hi:[1,0,1]" TAIL OPS);
        else if (FORM == 7) asm volatile(HEAD "v_pk_fma_f32 v[102:103], v[100:101], s[20:21], v[104:105] op_sel:[0,1,1] op_sel_hi:[0,0,0]" TAIL OPS);
        else asm volatile(HEAD "v_pk_fma_f32 v[102:103], s[20:21], v[100:101], v[104:105] op_sel:[0,0,1] op_sel_hi:[1,1,0]" TAIL OPS);
        float want_lo, want_hi;
        if (FORM == 0 || FORM == 1 || FORM == 3 || FORM == 7) { want_lo = fmaf(rz, fx, bz); want_hi = fmaf(ry, fx, by); } // s21 * v100 + v105 ; s20 * v100 + v104
        else if (FORM == 2) { want_lo = fmaf(ry, fx, by); want_hi = fmaf(rz, 288.0f, bz); }             // s20 * v100 + v104 ; s21 * v101 + v105
        else if (FORM == 4) { want_lo = rz * fx; want_hi = ry * fx; }
        else if (FORM == 5) { want_lo = fmaf(rz, fx, by); want_hi = fmaf(ry, 288.0f, bz); }             // s21 * v100 + v104 ; s20 * v101 + v105
        else if (FORM == 6) { want_lo = fmaf(ry, fx, bz); want_hi = fmaf(rz, fx, bz); }                 // s20 * v100 + v105 ; s21 * v100 + v105
        else { want_lo = fmaf(ry, fx, bz); want_hi = fmaf(rz, 288.0f, by); }                           // s20 * v100 + v105 ; s21 * v101 + v104
        if ((lo != want_lo) || (hi != want_hi)) {
            if (!bad) first = ((unsigned long long)__float_as_uint(lo) << 32) | __float_as_uint(hi);
            ++bad;
        }
    }
    if (bad) {
        atomicAdd(&out[1 + (lane >> 4)], (unsigned long long)bad);
        atomicMax(&out[5], first);
    }
}

template <int FORM, int BUSY>
static void run(const char *name, unsigned long long *d_out, const float *d_in, int blocks, int threads, int busy_mask) {
    (void)hipMemset(d_out, 0, 6 * sizeof(unsigned long long));
    hipLaunchKernelGGL((probe<FORM, BUSY>), dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 200000, busy_mask);
    (void)hipDeviceSynchronize();
    unsigned long long h[6];
    (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    unsigned lo = (unsigned)(h[5] >> 32), hi = (unsigned)h[5];
    printf("%-44s %4d x %4d, busy waves mask %x: wrong by quarter of the wave %llu %llu %llu %llu", name, blocks, threads, busy_mask, h[1], h[2], h[3], h[4]);
    if (h[5]) printf("   (one wrong pair: low %g high %g)", *(float *)&lo, *(float *)&hi);
    printf("\n");
}

int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;
    float *d_in;
    unsigned long long *d_out;
    (void)hipMalloc(&d_in, 4096);
    (void)hipMalloc(&d_out, 64);
    (void)hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice);
    const int shapes[][3] = {{512, 512, 0x8}, {512, 256, 0x8}, {512, 512, 0x0}};
    for (auto &s : shapes) {
        run<0, 3>("SGPR pair + op_sel | others: MFMA + LDS", d_out, d_in, s[0], s[1], s[2]);
        run<0, 1>("SGPR pair + op_sel | others: MFMA", d_out, d_in, s[0], s[1], s[2]);
        run<0, 2>("SGPR pair + op_sel | others: LDS atomics", d_out, d_in, s[0], s[1], s[2]);
        run<0, 4>("SGPR pair + op_sel | others: plain FMA", d_out, d_in, s[0], s[1], s[2]);
        run<1, 3>("VGPR pair + op_sel | others: MFMA + LDS", d_out, d_in, s[0], s[1], s[2]);
        run<2, 3>("SGPR pair, no op_sel | others: MFMA + LDS", d_out, d_in, s[0], s[1], s[2]);
        run<3, 3>("two v_fma_f32 | others: MFMA + LDS", d_out, d_in, s[0], s[1], s[2]);
        run<4, 3>("v_pk_mul_f32 SGPR + op_sel | others: MFMA + LDS", d_out, d_in, s[0], s[1], s[2]);
        run<5, 1>("only the SGPR pair swapped | others: MFMA", d_out, d_in, s[0], s[1], s[2]);
        run<6, 1>("only source 2 re-routed | others: MFMA", d_out, d_in, s[0], s[1], s[2]);
        run<7, 1>("SGPR pair as source 1 | others: MFMA", d_out, d_in, s[0], s[1], s[2]);
        run<8, 1>("only source 2 swapped | others: MFMA", d_out, d_in, s[0], s[1], s[2]);
        run<0, 8>("SGPR pair + op_sel | others: MFMA 16x16x32", d_out, d_in, s[0], s[1], s[2]);
    }
    return 0;
}
