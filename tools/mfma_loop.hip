// Micro-benchmark (gfx950): the block loop of the matrix-core sphere filter on its own -- table words from
// LDS one block ahead, two dependent v_mfma_f32_32x32x16_f16 per ray half, 16 v_or3_b32 over the 32 results,
// one ballot -- in variants that leave parts out, to see which part sets the pace.  16 waves per workgroup,
// one workgroup per CU, as the render kernel runs it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_loop.bin tools/mfma_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

enum { V_FULL, V_NO_LDS, V_NO_OR, V_MFMA_ONLY, V_OR_ONLY, V_TILES, V_TILES_BARRIER, V_TILES_STREAM, V_MODES };
static const char *names[V_MODES] = {"full loop", "no LDS reads", "no or / ballot", "MFMA only", "or + LDS, no MFMA",
                                     "full, restarted every 16 blocks", "  + a barrier every 16 blocks", "  + the next 16 blocks staged through registers"};

__device__ __forceinline__ float16v tile(half8 a0, half8 a1, half8 b0, half8 b1) {
    const float16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, zero, 0, 0, 0), 0, 0, 0);
}
__device__ __forceinline__ uint32_t or16(const float16v &acc) {
    uint32_t any = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) any |= __float_as_uint(acc[v]);
    return any;
}

template <int MODE>
__global__ __launch_bounds__(1024, 4) void k(const uint4v *__restrict__ gtab, int n_blocks, int reps, float *out,
                                             unsigned long long *cyc, float s) {
    extern __shared__ __attribute__((aligned(16))) uint4v tab[];
    for (int i = threadIdx.x; i < (n_blocks + 1) * 128; i += 1024) tab[i] = gtab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint4v *tl = tab + lane;
    half8 b00, b01, b10, b11;
    for (int j = 0; j < 8; ++j) {
        b00[j] = (_Float16)(s * (1 + j + lane));
        b01[j] = (_Float16)(s * (2 + j));
        b10[j] = (_Float16)(s * (3 + j + lane));
        b11[j] = (_Float16)(s * (4 + j));
    }
    unsigned found = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        half8 a0 = __builtin_bit_cast(half8, tl[0]), a1 = __builtin_bit_cast(half8, tl[64]);
        float16v acc0 = tile(a0, a1, b00, b01);
#pragma unroll 1
        for (int blk = 0; blk < n_blocks; ++blk) {
            float16v acc1;
            if (MODE == V_OR_ONLY) {
                acc1 = acc0;
                acc1[0] += 1.0f;
            } else {
                acc1 = tile(a0, a1, b10, b11);
            }
            if (MODE != V_NO_LDS && MODE != V_MFMA_ONLY) {
                a0 = __builtin_bit_cast(half8, tl[(2 * blk + 2) * 64]);
                a1 = __builtin_bit_cast(half8, tl[(2 * blk + 3) * 64]);
            }
            if (MODE != V_NO_OR && MODE != V_MFMA_ONLY) {
                const bool some = __ballot((int)(or16(acc0) | or16(acc1)) < 0) != 0;
                if (some) found++;
            } else {
                asm volatile("" ::"v"(acc0), "v"(acc1));
            }
            if (MODE == V_OR_ONLY) {
                acc0[1] += 1.0f;
            } else {
                acc0 = tile(a0, a1, b00, b01);
            }
        }
        asm volatile("" ::"v"(acc0));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)found;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const uint4v *dtab, int n_blocks, int reps, float *d, unsigned long long *dc, std::vector<unsigned long long> &h) {
    const size_t lds = (size_t)(n_blocks + 1) * 2048;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, dtab, n_blocks, 2, d, dc, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, dtab, n_blocks, reps, d, dc, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const size_t nw = 256 * 16;
    hipMemcpy(h.data(), dc, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nw);
    const double blocks = (double)reps * n_blocks;
    // wall time per block per SIMD (4 waves share a SIMD) at the shader clock the event time implies
    printf("  %-44s %8.3f ms  = %6.1f ns per block per wave; s_memtime median %7.1f ticks per block per wave\n", names[MODE], ms,
           ms * 1e6 / blocks, (double)h[nw / 2] / blocks);
}

// The same work per block (32 table rows x 64 rays, K = 32) as eight v_mfma_f32_16x16x32_f16: one instruction covers the
// whole K of 16 rows x 16 rays (4 result registers); the table block is two A operands, the rays four B operands.
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024, 4) void k16(const uint4v *__restrict__ gtab, int n_blocks, int reps, float *out,
                                               unsigned long long *cyc, float s) {
    extern __shared__ __attribute__((aligned(16))) uint4v tab[];
    for (int i = threadIdx.x; i < (n_blocks + 1) * 128; i += 1024) tab[i] = gtab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint4v *tl = tab + lane;
    half8 b[4];
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 8; ++j) b[g][j] = (_Float16)(s * (1 + g + j + (g & 1 ? 0 : lane)));
    unsigned found = 0;
    const float4v zero = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        half8 a0 = __builtin_bit_cast(half8, tl[0]), a1 = __builtin_bit_cast(half8, tl[64]);
        float4v c[8];
        c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[0], zero, 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[0], zero, 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[1], zero, 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[1], zero, 0, 0, 0);
#pragma unroll 1
        for (int blk = 0; blk < n_blocks; ++blk) {
            c[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[2], zero, 0, 0, 0);
            c[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[2], zero, 0, 0, 0);
            c[6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[3], zero, 0, 0, 0);
            c[7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[3], zero, 0, 0, 0);
            a0 = __builtin_bit_cast(half8, tl[(2 * blk + 2) * 64]);
            a1 = __builtin_bit_cast(half8, tl[(2 * blk + 3) * 64]);
            uint32_t any = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) any |= __float_as_uint(c[t][v]);
            const bool some = __ballot((int)any < 0) != 0;
            if (some) found++;
            c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[0], zero, 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[0], zero, 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[1], zero, 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[1], zero, 0, 0, 0);
        }
        asm volatile("" ::"v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)found;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

void run16(const uint4v *dtab, int n_blocks, int reps, float *d, unsigned long long *dc, std::vector<unsigned long long> &h) {
    const size_t lds = (size_t)(n_blocks + 1) * 2048;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k16, dim3(256), dim3(1024), lds, 0, dtab, n_blocks, 2, d, dc, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k16, dim3(256), dim3(1024), lds, 0, dtab, n_blocks, reps, d, dc, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const size_t nw = 256 * 16;
    hipMemcpy(h.data(), dc, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nw);
    const double blocks = (double)reps * n_blocks;
    printf("  %-44s %8.3f ms  = %6.1f ns per block per wave; s_memtime median %7.1f ticks per block per wave\n",
           "full loop as eight 16x16x32 products", ms, ms * 1e6 / blocks, (double)h[nw / 2] / blocks);
}

// The streamed form: the loop restarts every 16 blocks (pipeline prologue and one wasted read-ahead tile), with a
// workgroup barrier per tile, and with the next tile fetched from global memory into registers before the blocks
// and stored to the other LDS buffer after them.
template <int MODE>
__global__ __launch_bounds__(1024, 4) void kt(const uint4v *__restrict__ gtab, int n_tiles, int reps, float *out,
                                              unsigned long long *cyc, float s) {
    extern __shared__ __attribute__((aligned(16))) uint4v tab[];
    constexpr int TB = 16, TW = (TB + 1) * 128;
    for (int i = threadIdx.x; i < 2 * TW; i += 1024) tab[i] = gtab[i % (TB * 128)];
    __syncthreads();
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    half8 b00, b01, b10, b11;
    for (int j = 0; j < 8; ++j) {
        b00[j] = (_Float16)(s * (1 + j + lane));
        b01[j] = (_Float16)(s * (2 + j));
        b10[j] = (_Float16)(s * (3 + j + lane));
        b11[j] = (_Float16)(s * (4 + j));
    }
    unsigned found = 0;
    for (int r = 0; r < reps; ++r) {
        for (int t = 0; t < n_tiles; ++t) {
            uint4v next[2];
            if (MODE == V_TILES_STREAM) {
                const uint4v *g = gtab + (size_t)((t + 1) % n_tiles) * TB * 128;
                next[0] = g[tid];
                next[1] = g[tid + 1024];
            }
            const uint4v *tl = tab + (t & 1) * TW + lane;
            half8 a0 = __builtin_bit_cast(half8, tl[0]), a1 = __builtin_bit_cast(half8, tl[64]);
            float16v acc0 = tile(a0, a1, b00, b01);
#pragma unroll 1
            for (int blk = 0; blk < TB; ++blk) {
                float16v acc1 = tile(a0, a1, b10, b11);
                a0 = __builtin_bit_cast(half8, tl[(2 * blk + 2) * 64]);
                a1 = __builtin_bit_cast(half8, tl[(2 * blk + 3) * 64]);
                const bool some = __ballot((int)(or16(acc0) | or16(acc1)) < 0) != 0;
                if (some) found++;
                acc0 = tile(a0, a1, b00, b01);
            }
            asm volatile("" ::"v"(acc0));
            if (MODE == V_TILES_STREAM) {
                uint4v *dst = tab + ((t + 1) & 1) * TW;
                dst[tid] = next[0];
                dst[tid + 1024] = next[1];
            }
            if (MODE != V_TILES) __syncthreads();
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)found;
    (void)cyc;
}

template <int MODE>
void run_tiled(const uint4v *dtab, int n_tiles, int reps, float *d, unsigned long long *dc) {
    const size_t lds = 2 * 17 * 2048;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&kt<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kt<MODE>, dim3(256), dim3(1024), lds, 0, dtab, n_tiles, 2, d, dc, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kt<MODE>, dim3(256), dim3(1024), lds, 0, dtab, n_tiles, reps, d, dc, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double blocks = (double)reps * n_tiles * 16;
    printf("  %-44s %8.3f ms  = %6.1f ns per block per wave\n", names[MODE], ms, ms * 1e6 / blocks);
}

int main(int argc, char **argv) {
    const bool random_data = argc > 1;
    const int n_blocks = 16, reps = 20000;
    std::vector<uint32_t> tabh((size_t)(n_blocks + 1) * 128 * 4);
    // half-float words near 1.0 so that every product is positive: no sign bit, no candidate
    // (argument "random": random magnitudes below 1, still all positive, to see what the data costs in power)
    unsigned seed = 12345u;
    auto rnd_pos = [&]() { seed = seed * 1664525u + 1013904223u; return (uint32_t)(0x3000u + ((seed >> 12) & 0x0bffu)); };
    for (auto &w : tabh) w = random_data ? (rnd_pos() | (rnd_pos() << 16)) : 0x3c003c00u;
    uint4v *dtab; hipMalloc(&dtab, tabh.size() * 4);
    hipMemcpy(dtab, tabh.data(), tabh.size() * 4, hipMemcpyHostToDevice);
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    unsigned long long *dc; hipMalloc(&dc, 256 * 16 * 8);
    std::vector<unsigned long long> h(256 * 16);
    printf("16 waves per CU (4 per SIMD), %d blocks of 32 rows x 64 rays, %d scans per wave\n", n_blocks, reps);
    run<V_FULL>(dtab, n_blocks, reps, d, dc, h);
    run16(dtab, n_blocks, reps, d, dc, h);
    run<V_FULL>(dtab, n_blocks, reps, d, dc, h);
    run16(dtab, n_blocks, reps, d, dc, h);
    run<V_NO_LDS>(dtab, n_blocks, reps, d, dc, h);
    run<V_NO_OR>(dtab, n_blocks, reps, d, dc, h);
    run<V_MFMA_ONLY>(dtab, n_blocks, reps, d, dc, h);
    run<V_OR_ONLY>(dtab, n_blocks, reps, d, dc, h);
    {
        // a table of 20 tiles (the 10 001-sphere scene), streamed
        const int n_tiles = 20;
        std::vector<uint32_t> big((size_t)n_tiles * 16 * 128 * 4, 0x3c003c00u);
        if (random_data) for (auto &w : big) w = rnd_pos() | (rnd_pos() << 16);
        uint4v *dbig; hipMalloc(&dbig, big.size() * 4);
        hipMemcpy(dbig, big.data(), big.size() * 4, hipMemcpyHostToDevice);
        run_tiled<V_TILES>(dbig, n_tiles, random_data ? 20000 : 1000, d, dc);
        run_tiled<V_TILES_BARRIER>(dbig, n_tiles, 1000, d, dc);
        run_tiled<V_TILES_STREAM>(dbig, n_tiles, 1000, d, dc);
    }
    return 0;
}
