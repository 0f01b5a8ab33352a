// Probe for gfx950: (1) the register layout of v_mfma_f32_32x32x2_f32 results, (2) whether fp32 MFMA
// and packed-FP32 VALU work overlap on a SIMD (same wave, and across waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

// D = A (32x2) * B (2x32): lane l supplies A[l%32][l/32] and B[l/32][l%32].
__global__ void layout_kernel(float *out) {
    int l = threadIdx.x;
    float a = 1.0f + (l % 32) + 100.0f * (l / 32); // A[row][k] = 1 + row + 100 k
    float b = (l / 32 == 0) ? 1.0f : 0.0f;         // B[0][col] = 1, B[1][col] = 0  ->  D[row][col] = A[row][0] = 1 + row
    f16v c = {0};
    f16v d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = d[r];
    // second product to get the column mapping: B[0][col] = col, A[row][0] = 1, A[row][1] = 0
    a = (l / 32 == 0) ? 1.0f : 0.0f;
    b = (l / 32 == 0) ? (float)(l % 32) : 0.0f;
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[1024 + l * 16 + r] = d[r];
}

typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

// Same with the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16): MODE 0 packed-FMA only, 1 MFMA only, 2 both
template <int MODE, int NV>
__global__ __launch_bounds__(256) void mix_bf16_kernel(float *out, int iters, float s) {
    float x = threadIdx.x * 1e-3f;
    f2 a0 = {x, x}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    f2 ss = {s, s}, xx = {x, x};
    f16v c0 = {0}, c1 = {0};
    bf8v fa, fb;
    for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(x + k); fb[k] = (__bf16)(s * k); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE != 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, c1, 0, 0, 0);
            }
            if (MODE != 1) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(ss), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(ss), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(ss), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(ss), "v"(xx));
                }
            }
        }
    }
    f2 r = a0 + a1 + a2 + a3;
    float acc = r.x + r.y;
    for (int k = 0; k < 16; ++k) acc += c0[k] + c1[k];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int NV>
static void run_bf16(const char *name, float *d_out, int wgs_per_cu) {
    int iters = 2000, cus = 256;
    dim3 grid(cus * wgs_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mix_bf16_kernel<MODE, NV><<<grid, block>>>(d_out, 10, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix_bf16_kernel<MODE, NV><<<grid, block>>>(d_out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)grid.x * 4;
    double pk = MODE != 1 ? waves * iters * 4.0 * NV * 4 : 0;
    double mf = MODE != 0 ? waves * iters * 4.0 * 2 : 0;
    printf("%-28s wgs/cu %d: %8.3f ms  valu %6.1f TF  mfma(bf16) %7.1f TF\n", name, wgs_per_cu, ms,
           pk * 64 * 4 / (ms * 1e-3) / 1e12, mf * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12);
}

// MODE 0: packed-FMA only; 1: MFMA only; 2: both in one instruction stream (NV valu per MFMA)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void mix_kernel(float *out, int iters, float s) {
    float x = threadIdx.x * 1e-3f;
    f2 a0 = {x, x}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    f2 ss = {s, s}, xx = {x, x};
    f16v c0 = {0}, c1 = {0};
    float fa = x, fb = s;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE != 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, c1, 0, 0, 0);
            }
            if (MODE != 1) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(ss), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(ss), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(ss), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(ss), "v"(xx));
                }
            }
        }
    }
    f2 r = a0 + a1 + a2 + a3;
    float acc = r.x + r.y;
    for (int k = 0; k < 16; ++k) acc += c0[k] + c1[k];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int NV>
static void run(const char *name, float *d_out, int wgs_per_cu) {
    int iters = 2000, cus = 256;
    dim3 grid(cus * wgs_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mix_kernel<MODE, NV><<<grid, block>>>(d_out, 10, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix_kernel<MODE, NV><<<grid, block>>>(d_out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)grid.x * 4;
    double pk = MODE != 1 ? waves * iters * 4.0 * NV * 4 : 0;   // packed instructions issued (per wave)
    double mf = MODE != 0 ? waves * iters * 4.0 * 2 : 0;        // MFMA instructions issued
    double tf_valu = pk * 64 * 4 / (ms * 1e-3) / 1e12;          // 2 lanes-wide x fma = 4 flop per lane
    double tf_mfma = mf * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
    printf("%-28s wgs/cu %d: %8.3f ms  valu %6.1f TF  mfma %6.1f TF\n", name, wgs_per_cu, ms, tf_valu, tf_mfma);
}

int main() {
    float *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * 4 * sizeof(float));
    std::vector<float> h(2048);
    layout_kernel<<<1, 64>>>(d_out);
    hipMemcpy(h.data(), d_out, 2048 * 4, hipMemcpyDeviceToHost);
    printf("row of D held by (lane, reg): lane 0: ");
    for (int r = 0; r < 16; ++r) printf("%g ", h[0 * 16 + r] - 1);
    printf("\n                              lane 32: ");
    for (int r = 0; r < 16; ++r) printf("%g ", h[32 * 16 + r] - 1);
    printf("\n                              lane 5: ");
    for (int r = 0; r < 16; ++r) printf("%g ", h[5 * 16 + r] - 1);
    printf("\ncolumn of D held by lane 0, 5, 37 (reg 0): %g %g %g\n", h[1024 + 0], h[1024 + 5 * 16], h[1024 + 37 * 16]);
    for (int w : {1, 2, 4}) {
        run<0, 4>("pk_fma only (16/iter-step)", d_out, w);
        run<1, 4>("mfma only (2/iter-step)", d_out, w);
        run<2, 4>("both: 16 pk per 2 mfma", d_out, w);
        run<2, 8>("both: 32 pk per 2 mfma", d_out, w);
        run<0, 8>("pk_fma only (32/iter-step)", d_out, w);
    }
    for (int w : {1, 2, 4}) {
        run_bf16<0, 4>("pk_fma only (16/iter-step)", d_out, w);
        run_bf16<1, 4>("bf16 mfma only (2/iter-step)", d_out, w);
        run_bf16<2, 4>("both: 16 pk per 2 mfma", d_out, w);
        run_bf16<2, 2>("both: 8 pk per 2 mfma", d_out, w);
        run_bf16<0, 2>("pk_fma only (8/iter-step)", d_out, w);
    }
    return 0;
}
