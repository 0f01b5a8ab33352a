// Micro-benchmark (gfx950), second set: issue cost of the integer / conversion instructions around the
// matrix-core filter (sign gathering, packing), alone and beside v_mfma_f32_32x32x16_f16, at 1, 2 and 4
// waves per SIMD.  Same method as valu_mix.hip: independent chains in inline asm, cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_mix2.bin tools/valu_mix2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

enum { M_ALIGNBIT, M_OR3, M_AND, M_CMP, M_PKRTZ, M_PERM, M_LSHLOR, M_BFI, M_MIN3, M_CVTF32F16, M_MFMA, M_MFMA_ALIGN8,
       M_MFMA_ALIGN16, M_MFMA_OR3_8, M_MODES };
static const char *names[M_MODES] = {"v_alignbit_b32 x8", "v_or3_b32 x8", "v_and_b32 x8", "v_cmp_lt_f32 x8", "v_cvt_pkrtz_f16_f32 x8",
                                     "v_perm_b32 x8", "v_lshl_or_b32 x8", "v_bfi_b32 x8", "v_min3_f32 x8", "v_cvt_f32_f16 x8",
                                     "mfma 32x32x16 f16 x1", "1 mfma + 8 alignbit", "1 mfma + 16 alignbit", "1 mfma + 8 or3"};
// instructions counted per trip (the mixes count every instruction, the MFMA included)
static const int per_trip[M_MODES] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 1, 9, 17, 9};

#define EIGHT(OP, A) OP(A##0) OP(A##1) OP(A##2) OP(A##3) OP(A##4) OP(A##5) OP(A##6) OP(A##7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float s) {
    float x = threadIdx.x * 1e-3f + 0.5f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned m = 1664525u, c = __float_as_uint(s);
    half8 a = {(_Float16)x, 1, 2, 3, 4, 5, 6, 7}, b = {(_Float16)s, 2, 1, 1, 1, 1, 1, 1};
    float16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#define OP_ALIGN(U) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(U) : "v"(m));
#define OP_OR3(U) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(U) : "v"(m), "v"(c));
#define OP_AND(U) asm volatile("v_and_b32 %0, %0, %1" : "+v"(U) : "v"(m));
#define OP_CMP(U) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(U), "v"(c) : "vcc");
#define OP_PKRTZ(U) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(U) : "v"(c));
#define OP_PERM(U) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(U) : "v"(m), "v"(c));
#define OP_LSHLOR(U) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(U) : "v"(m));
#define OP_BFI(U) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(U) : "v"(m), "v"(c));
#define OP_MIN3(U) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(U) : "v"(m), "v"(c));
#define OP_CVT(U) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(U));
            if (MODE == M_ALIGNBIT) { EIGHT(OP_ALIGN, u) }
            else if (MODE == M_OR3) { EIGHT(OP_OR3, u) }
            else if (MODE == M_AND) { EIGHT(OP_AND, u) }
            else if (MODE == M_CMP) { EIGHT(OP_CMP, u) }
            else if (MODE == M_PKRTZ) { EIGHT(OP_PKRTZ, u) }
            else if (MODE == M_PERM) { EIGHT(OP_PERM, u) }
            else if (MODE == M_LSHLOR) { EIGHT(OP_LSHLOR, u) }
            else if (MODE == M_BFI) { EIGHT(OP_BFI, u) }
            else if (MODE == M_MIN3) { EIGHT(OP_MIN3, u) }
            else if (MODE == M_CVTF32F16) { EIGHT(OP_CVT, u) }
            else if (MODE == M_MFMA) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            } else if (MODE == M_MFMA_ALIGN8) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
                EIGHT(OP_ALIGN, u)
            } else if (MODE == M_MFMA_ALIGN16) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
                EIGHT(OP_ALIGN, u)
                EIGHT(OP_ALIGN, u)
            } else if (MODE == M_MFMA_OR3_8) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
                EIGHT(OP_OR3, u)
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int j = 0; j < 16; ++j) r += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(int waves_per_simd, int iters, float *d, unsigned long long *dc, std::vector<unsigned long long> &h) {
    int grid = 256 * waves_per_simd; // one workgroup of 256 threads = one wave per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, 64, 1.0001f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    size_t nw = (size_t)grid * 4;
    hipMemcpy(h.data(), dc, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nw);
    double med = (double)h[nw / 2];
    double trips = (double)iters * 8;
    printf("  %-24s w/SIMD %d: %7.2f cyc per trip per wave, %6.2f cyc per instr per SIMD  (%.3f ms)\n", names[MODE],
           waves_per_simd, med / trips, med / (trips * per_trip[MODE] * waves_per_simd), ms);
}

int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *dc; hipMalloc(&dc, 256 * 8 * 4 * 8);
    std::vector<unsigned long long> h(256 * 8 * 4);
    const int iters = 2000;
    for (int w : {1, 2, 4}) {
        printf("waves per SIMD = %d\n", w);
        run<M_ALIGNBIT>(w, iters, d, dc, h);
        run<M_OR3>(w, iters, d, dc, h);
        run<M_AND>(w, iters, d, dc, h);
        run<M_CMP>(w, iters, d, dc, h);
        run<M_PKRTZ>(w, iters, d, dc, h);
        run<M_PERM>(w, iters, d, dc, h);
        run<M_LSHLOR>(w, iters, d, dc, h);
        run<M_BFI>(w, iters, d, dc, h);
        run<M_MIN3>(w, iters, d, dc, h);
        run<M_CVTF32F16>(w, iters, d, dc, h);
        run<M_MFMA>(w, iters, d, dc, h);
        run<M_MFMA_ALIGN8>(w, iters, d, dc, h);
        run<M_MFMA_ALIGN16>(w, iters, d, dc, h);
        run<M_MFMA_OR3_8>(w, iters, d, dc, h);
    }
    return 0;
}
