// rt_amd.hip -- the one translation unit of librt_amd.so: HIP kernels (gfx950) and the C ABI declared in
// include/rt_amd.h.  Its parts, one concern each, are included below in this order:
//   rt_kernel_common.h   launch parameters, device scene view, fixed-point radiance sums, the plain sphere scans
//   render_kernel.inc    the persistent path-trace kernel and its per-tile primary-ray lists
//   query_kernels.inc    stage-wise kernels: filter values, first hits, scatter, random words
//   abi_context.inc      rt_ctx, error reporting, rt_create / rt_destroy
//   abi_scene.inc        scene / light / camera upload, the matrix-core filter's table
//   abi_render.inc       rt_render, rt_render_host: work-item shape, variant and grid of a launch
//   abi_query.inc        rt_first_hit*, rt_scatter_host, rt_rng4*, rt_filter_*
//   abi_group.inc        rt_group_*: one frame on several GPUs of one process (rccl_api.h binds RCCL)
// The work-counter array of rt_render_noalloc, the look-up of its device address and that entry point itself
// are in this file.
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
#include <chrono>

#include "rt_kernel_common.h"
#include "rccl_api.h"

#define RT_API extern "C" __attribute__((visibility("default")))

#include "render_kernel.inc"
#include "query_kernels.inc"

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

#include "abi_context.inc"
#include "abi_scene.inc"
#include "abi_render.inc"
#include "abi_query.inc"
#include "abi_group.inc"

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

