// Micro-benchmark: sustained FP32 VALU rate on gfx950 for plain and packed FMA,
// and for the ray/sphere inner-loop instruction mix, versus waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float x = threadIdx.x * 1e-3f;
    if (MODE == 0) { // 8 independent v_fma_f32 chains
        float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = fmaf(a0, s, x); a1 = fmaf(a1, s, x); a2 = fmaf(a2, s, x); a3 = fmaf(a3, s, x);
                a4 = fmaf(a4, s, x); a5 = fmaf(a5, s, x); a6 = fmaf(a6, s, x); a7 = fmaf(a7, s, x);
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (MODE == 1) { // 8 independent v_pk_fma_f32 chains
        f2 a0 = {x, x}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        f2 ss = {s, s}, xx = {x, x};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(ss), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(ss), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(ss), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(ss), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(ss), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(ss), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(ss), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(ss), "v"(xx));
            }
        }
        f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
    } else if (MODE == 2) { // plain fma via asm to pin the instruction
        float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(s), "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(s), "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(s), "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(s), "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(s), "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(s), "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(s), "v"(x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(s), "v"(x));
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (MODE == 3) { // packed mul/add mix (v_pk_mul_f32 + v_pk_add_f32)
        f2 a0 = {x, x}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
        f2 ss = {s, s}, xx = {x, x};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(ss));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a1) : "v"(xx));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(ss));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a3) : "v"(xx));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(ss));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a1) : "v"(xx));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(ss));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a3) : "v"(xx));
            }
        }
        f2 r = a0 + a1 + a2 + a3;
        out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
    } else if (MODE == 4) { // v_cmp + v_cndmask mix with fma (select-heavy)
        float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = fmaf(a0, s, x);
                a1 = a0 > a1 ? a0 : a1 * s;
                a2 = fmaf(a2, s, x);
                a3 = a2 > a3 ? a2 : a3 * s;
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    }
}

template <int MODE>
double run(int blocks_per_cu, int iters, float *d, double ops_per_iter_per_lane) {
    int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, 16, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lanes = (double)grid * 256;
    return lanes * iters * ops_per_iter_per_lane / (ms * 1e-3) / 1e12; // Tera lane-ops/s
}

int main() {
    float *d; hipMalloc(&d, 256 * 64 * 256 * 4);
    int iters = 20000;
    printf("blocks/CU(256thr) waves/SIMD |  fma TFLOP/s | pk_fma TFLOP/s | asm fma | pk mul/add TFLOP/s | fma+cmp/sel Tops/s\n");
    for (int b : {1, 2, 3, 4, 6, 8}) {
        double f0 = run<0>(b, iters, d, 64 * 2.0);
        double f1 = run<1>(b, iters, d, 64 * 4.0);
        double f2_ = run<2>(b, iters, d, 64 * 2.0);
        double f3 = run<3>(b, iters, d, 64 * 2.0);
        double f4 = run<4>(b, iters, d, 16 * 5.0); // fma, cmp, mul, cndmask ... count issued VALU: fma,cmp,mul,cndmask x2 = 8/unroll? approx
        printf("%d %d | %.1f | %.1f | %.1f | %.1f | %.1f\n", b, b, f0, f1, f2_, f3, f4);
    }
    return 0;
}
