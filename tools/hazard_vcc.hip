// Probe: how many wait states does gfx950 need between a VALU compare that writes VCC and a VOP3 v_cndmask_b32 that
// reads VCC as an explicit scalar operand (the form the compiler must use when a source carries a neg modifier)?
// The compiler's hazard recognizer asks for two wait states and counts ANY instruction as one, scalar ALU moves included.
// Each wave runs the sequence  v_cmp (VCC := 0) ... v_cmp (VCC := all ones) ; <filler> ; v_cndmask_e64  many times and
// counts, per quarter of the wave, the lanes whose select still saw the OLD mask.
// build: hipcc --offload-arch=gfx950 -O2 -o hazard_vcc tools/hazard_vcc.hip ; run: ./hazard_vcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int FILLER> // 0: nothing, 1: one s_mov, 2: two s_mov, 3: s_nop 0, 4: s_nop 1, 5: two v_nop, 6: three s_mov, 7: s_nop 3
__global__ void probe(unsigned long long *out, const float *in, int iters, int busy_mask) {
    const int lane = threadIdx.x & 63;
    const float a = in[threadIdx.x];        // > 0 in every lane
    const float b = -a;
    unsigned bad = 0;
    float c = 1.0f;
    unsigned t0 = 1, t1 = 2;
    // some waves only load from memory in a loop, so that the probing waves often have the SIMD's issue slots to themselves
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((busy_mask >> (wave & 3)) & 1) {
        float s = 0;
        for (int i = 0; i < iters / 8; ++i) s += __builtin_nontemporal_load(in + ((i * 977 + threadIdx.x) & 1023));
        if (s == 12345.678f) out[0] = 1; // (never)
        return;
    }
    for (int i = 0; i < iters; ++i) {
        float r;
        c += 1.0f;
        if (FILLER == 0)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        else if (FILLER == 1)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\ts_mov_b32 %1, %2\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        else if (FILLER == 2)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\ts_mov_b32 %1, %2\n\ts_mov_b32 %2, %1\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        else if (FILLER == 3)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        else if (FILLER == 4)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        else if (FILLER == 5)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\tv_nop\n\tv_nop\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        else if (FILLER == 6)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\ts_mov_b32 %1, %2\n\ts_mov_b32 %2, %1\n\ts_mov_b32 %1, %2\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        else if (FILLER == 7)
            asm volatile("v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\ts_nop 3\n\tv_cndmask_b32_e64 %0, %5, -%5, vcc"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc");
        bad += r != -c; // the select saw the old mask (all zeros) if it returned +c
    }
    if (bad) atomicAdd(&out[1 + (lane >> 4)], (unsigned long long)bad);
    if (t0 == 0xdeadbeefu && t1 == 0xdeadbeefu) out[0] = 2; // (keeps the scalar moves alive)
}


// Second probe: the select's SOURCE register is overwritten by the very next instruction, a packed FP32 multiply
// (v_cndmask_b32_e64 r, c, -c, vcc ; v_pk_mul_f32 v[c:c+1], ...), as in the code the compiler made for the side planes of
// a tile's pyramid.  KIND 0: v_pk_mul_f32 follows; 1: v_pk_fma_f32 follows; 2: a plain v_mul_f32 follows.
template <int KIND>
__global__ void probe_war(unsigned long long *out, const float *in, int iters, int busy_mask) {
    const int lane = threadIdx.x & 63;
    const float a = in[threadIdx.x];
    const float b = -a;
    unsigned bad = 0;
    float c = 1.0f;
    unsigned t0 = 1, t1 = 2;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((busy_mask >> (wave & 3)) & 1) {
        float s = 0;
        for (int i = 0; i < iters / 8; ++i) s += __builtin_nontemporal_load(in + ((i * 977 + threadIdx.x) & 1023));
        if (s == 12345.678f) out[0] = 1;
        return;
    }
    for (int i = 0; i < iters; ++i) {
        float r;
        c += 1.0f;
#define RT_WAR_HEAD "v_mov_b32 v100, %5\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v102, 2.0\n\tv_mov_b32 v103, 2.0\n\tv_mov_b32 v104, 4.0\n\tv_mov_b32 v105, 4.0\n\ts_nop 7\n\t" \
                    "v_cmp_gt_f32_e32 vcc, 0, %3\n\ts_nop 7\n\tv_cmp_gt_f32_e32 vcc, 0, %4\n\ts_mov_b32 %1, %2\n\ts_mov_b32 %2, %1\n\tv_cndmask_b32_e64 %0, v100, -v100, vcc\n\t"
        if (KIND == 0)
            asm volatile(RT_WAR_HEAD "v_pk_mul_f32 v[100:101], v[102:103], v[104:105]\n\ts_nop 7"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc", "v100", "v101", "v102", "v103", "v104", "v105");
        else if (KIND == 1)
            asm volatile(RT_WAR_HEAD "v_pk_fma_f32 v[100:101], v[102:103], v[104:105], v[104:105]\n\ts_nop 7"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc", "v100", "v101", "v102", "v103", "v104", "v105");
        else
            asm volatile(RT_WAR_HEAD "v_mul_f32_e32 v100, v102, v104\n\ts_nop 7"
                         : "=v"(r), "+s"(t0), "+s"(t1) : "v"(a), "v"(b), "v"(c) : "vcc", "v100", "v101", "v102", "v103", "v104", "v105");
        bad += r != -c;
    }
    if (bad) atomicAdd(&out[1 + (lane >> 4)], (unsigned long long)bad);
    if (t0 == 0xdeadbeefu && t1 == 0xdeadbeefu) out[0] = 2;
}

template <int KIND>
static void run_war(const char *name, unsigned long long *d_out, const float *d_in, int blocks, int threads, int busy_mask) {
    hipMemset(d_out, 0, 5 * sizeof(unsigned long long));
    hipLaunchKernelGGL(probe_war<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 200000, busy_mask);
    hipDeviceSynchronize();
    unsigned long long h[5];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("source overwritten by %-12s blocks %4d x %4d threads, loading waves mask %x: wrong selects by quarter of the wave: %llu %llu %llu %llu\n", name, blocks, threads, busy_mask,
           h[1], h[2], h[3], h[4]);
}

template <int FILLER>
static void run(const char *name, unsigned long long *d_out, const float *d_in, int blocks, int threads, int busy_mask) {
    hipMemset(d_out, 0, 5 * sizeof(unsigned long long));
    const int iters = 200000;
    hipLaunchKernelGGL(probe<FILLER>, dim3(blocks), dim3(threads), 0, 0, d_out, d_in, iters, busy_mask);
    hipDeviceSynchronize();
    unsigned long long h[5];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-12s blocks %4d x %4d threads, loading waves mask %x: lanes that saw the old VCC, by quarter of the wave: %llu %llu %llu %llu\n", name, blocks, threads, busy_mask,
           h[1], h[2], h[3], h[4]);
}

int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;
    float *d_in;
    unsigned long long *d_out;
    hipMalloc(&d_in, 4096);
    hipMalloc(&d_out, 64);
    hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice);
    const int shapes[][3] = {{256, 64, 0}, {256, 256, 0}, {256, 768, 0}, {256, 768, 0xe}, {256, 1024, 0xe}, {1024, 256, 0x6}};
    for (auto &s : shapes) {
        run<0>("no filler", d_out, d_in, s[0], s[1], s[2]);
        run<1>("1 s_mov", d_out, d_in, s[0], s[1], s[2]);
        run<2>("2 s_mov", d_out, d_in, s[0], s[1], s[2]);
        run<6>("3 s_mov", d_out, d_in, s[0], s[1], s[2]);
        run<3>("s_nop 0", d_out, d_in, s[0], s[1], s[2]);
        run<4>("s_nop 1", d_out, d_in, s[0], s[1], s[2]);
        run<5>("2 v_nop", d_out, d_in, s[0], s[1], s[2]);
        run<7>("s_nop 3", d_out, d_in, s[0], s[1], s[2]);
    }
    for (auto &s : shapes) {
        run_war<0>("v_pk_mul_f32", d_out, d_in, s[0], s[1], s[2]);
        run_war<1>("v_pk_fma_f32", d_out, d_in, s[0], s[1], s[2]);
        run_war<2>("v_mul_f32", d_out, d_in, s[0], s[1], s[2]);
    }
    return 0;
}
