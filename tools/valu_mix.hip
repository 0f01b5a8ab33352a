// Micro-benchmark (gfx950): issue cost, in shader cycles per wave-instruction per SIMD, of the
// instruction classes the render kernel is made of, alone and mixed with packed FMAs, at 1 to 8
// waves per SIMD.  Every stream is a set of independent chains written in inline asm so that the
// compiler cannot fuse, drop or re-order them.  Cycles come from s_memtime inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_mix.bin tools/valu_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));

enum { M_FMA, M_PKFMA, M_MAX3, M_CMPSEL, M_MULLO, M_MULHI, M_SQRT, M_SIN, M_RCP, M_PK_MAX3, M_PK_CMP, M_PK_SFMA,
       M_PK_MULLO, M_PK_SQRT, M_ADD_U32, M_XOR, M_PK_ADDU, M_PK_SGPR, M_MODES };
static const char *names[M_MODES] = {
    "v_fma_f32 x8", "v_pk_fma_f32 x8", "v_max3_f32 x8", "v_cmp+v_cndmask x4", "v_mul_lo_u32 x8", "v_mul_hi_u32 x8",
    "v_sqrt_f32 x8", "v_sin_f32 x8", "v_rcp_f32 x8", "7 pk_fma + 1 max3", "6 pk_fma + cmp + cndmask", "4 pk_fma + 4 fma",
    "4 pk_fma + 4 mul_lo", "6 pk_fma + 2 sqrt", "v_add_u32 x8", "v_xor_b32 x8", "4 pk_fma + 4 add_u32",
    "v_pk_fma_f32 sgpr-operand x8"};
// wave-instructions per loop trip
static const int per_trip[M_MODES] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float s, float sx) {
    float x = threadIdx.x * 1e-3f + 0.5f;
    float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    f2 p0 = {x, x}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
    f2 ss = {s, s}, xx = {x, x};
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned m = 1664525u;
    f2 sg = {sx, sx};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
        if (MODE == M_FMA) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(s), "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(s), "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(s), "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(s), "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(s), "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(s), "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(s), "v"(x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(s), "v"(x));
        } else if (MODE == M_PKFMA) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(ss), "v"(xx));
        } else if (MODE == M_PK_SGPR) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "s"(sg), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "s"(sg), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "s"(sg), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p3) : "s"(sg), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p4) : "s"(sg), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p5) : "s"(sg), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p6) : "s"(sg), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p7) : "s"(sg), "v"(xx));
        } else if (MODE == M_MAX3) {
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(s), "v"(x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(s), "v"(x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(s), "v"(x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(s), "v"(x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(s), "v"(x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(s), "v"(x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(s), "v"(x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(s), "v"(x));
        } else if (MODE == M_CMPSEL) {
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(s), "v"(x) : "vcc");
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(s), "v"(x) : "vcc");
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(s), "v"(x) : "vcc");
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(s), "v"(x) : "vcc");
        } else if (MODE == M_MULLO) {
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(m));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u1) : "v"(m));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u2) : "v"(m));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u3) : "v"(m));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u4) : "v"(m));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u5) : "v"(m));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u6) : "v"(m));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u7) : "v"(m));
        } else if (MODE == M_MULHI) {
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u0) : "v"(m));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u1) : "v"(m));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u2) : "v"(m));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u3) : "v"(m));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u4) : "v"(m));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u5) : "v"(m));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u6) : "v"(m));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u7) : "v"(m));
        } else if (MODE == M_ADD_U32) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(m));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u1) : "v"(m));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u2) : "v"(m));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u3) : "v"(m));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u4) : "v"(m));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u5) : "v"(m));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u6) : "v"(m));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u7) : "v"(m));
        } else if (MODE == M_XOR) {
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u0) : "v"(m));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u1) : "v"(m));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u2) : "v"(m));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u3) : "v"(m));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u4) : "v"(m));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u5) : "v"(m));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u6) : "v"(m));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u7) : "v"(m));
        } else if (MODE == M_SQRT) {
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a0));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a1));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a2));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a3));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a4));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a5));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a6));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a7));
        } else if (MODE == M_SIN) {
            asm volatile("v_sin_f32 %0, %0" : "+v"(a0));
            asm volatile("v_sin_f32 %0, %0" : "+v"(a1));
            asm volatile("v_sin_f32 %0, %0" : "+v"(a2));
            asm volatile("v_sin_f32 %0, %0" : "+v"(a3));
            asm volatile("v_sin_f32 %0, %0" : "+v"(a4));
            asm volatile("v_sin_f32 %0, %0" : "+v"(a5));
            asm volatile("v_sin_f32 %0, %0" : "+v"(a6));
            asm volatile("v_sin_f32 %0, %0" : "+v"(a7));
        } else if (MODE == M_RCP) {
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a1));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a2));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a3));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a4));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a5));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a6));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a7));
        } else if (MODE == M_PK_MAX3) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(ss), "v"(xx));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(s), "v"(x));
        } else if (MODE == M_PK_CMP) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(ss), "v"(xx));
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(s), "v"(x) : "vcc");
        } else if (MODE == M_PK_SFMA) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(ss), "v"(xx));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(s), "v"(x));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(ss), "v"(xx));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(s), "v"(x));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(ss), "v"(xx));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(s), "v"(x));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(ss), "v"(xx));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(s), "v"(x));
        } else if (MODE == M_PK_MULLO) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(ss), "v"(xx));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(ss), "v"(xx));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u1) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(ss), "v"(xx));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u2) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(ss), "v"(xx));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u3) : "v"(m));
        } else if (MODE == M_PK_ADDU) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(ss), "v"(xx));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(ss), "v"(xx));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u1) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(ss), "v"(xx));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u2) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(ss), "v"(xx));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(u3) : "v"(m));
        } else if (MODE == M_PK_SQRT) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(ss), "v"(xx));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a0));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(ss), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(ss), "v"(xx));
            asm volatile("v_sqrt_f32 %0, %0" : "+v"(a1));
        }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f2 r = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    out[blockIdx.x * blockDim.x + threadIdx.x] =
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r.x + r.y + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(int waves_per_simd, int iters, float *d, unsigned long long *dc, std::vector<unsigned long long> &h) {
    // one workgroup of 256 threads = one wave per SIMD; w workgroups per CU = w waves per SIMD
    int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, 64, 1.0001f, 1.0001f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, iters, 1.0001f, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    size_t nw = (size_t)grid * 4;
    hipMemcpy(h.data(), dc, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nw);
    double med = (double)h[nw / 2];
    double instr = (double)iters * 8 * per_trip[MODE];                 // per wave
    double cyc_per_instr_simd = med / (instr * waves_per_simd);         // all waves of the SIMD run side by side
    double ghz = instr * waves_per_simd / (ms * 1e-3) * cyc_per_instr_simd / 1e9;
    printf("  %-28s w/SIMD %d: %6.2f cyc per wave-instr per SIMD  (%.3f ms, ~%.2f GHz)\n", names[MODE], waves_per_simd,
           cyc_per_instr_simd, ms, ghz);
}

int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *dc; hipMalloc(&dc, 256 * 8 * 4 * 8);
    std::vector<unsigned long long> h(256 * 8 * 4);
    const int iters = 4000;
    for (int w : {1, 2, 4, 5, 8}) {
        printf("waves per SIMD = %d\n", w);
        run<M_FMA>(w, iters, d, dc, h);
        run<M_PKFMA>(w, iters, d, dc, h);
        run<M_PK_SGPR>(w, iters, d, dc, h);
        run<M_MAX3>(w, iters, d, dc, h);
        run<M_CMPSEL>(w, iters, d, dc, h);
        run<M_ADD_U32>(w, iters, d, dc, h);
        run<M_XOR>(w, iters, d, dc, h);
        run<M_MULLO>(w, iters, d, dc, h);
        run<M_MULHI>(w, iters, d, dc, h);
        run<M_SQRT>(w, iters, d, dc, h);
        run<M_SIN>(w, iters, d, dc, h);
        run<M_RCP>(w, iters, d, dc, h);
        run<M_PK_MAX3>(w, iters, d, dc, h);
        run<M_PK_CMP>(w, iters, d, dc, h);
        run<M_PK_SFMA>(w, iters, d, dc, h);
        run<M_PK_MULLO>(w, iters, d, dc, h);
        run<M_PK_ADDU>(w, iters, d, dc, h);
        run<M_PK_SQRT>(w, iters, d, dc, h);
    }
    return 0;
}
