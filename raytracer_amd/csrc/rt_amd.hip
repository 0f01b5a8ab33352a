// rt_amd.hip -- the one translation unit of librt_amd.so: HIP kernels (gfx950) and the C ABI declared in
// include/rt_amd.h.  Its parts, one concern each, are included below in this order:
//   rt_kernel_common.h   launch parameters, device scene view, fixed-point radiance sums, the plain sphere scans
//   render_kernel.inc    the persistent path-trace kernel and its per-tile primary-ray lists
//   query_kernels.inc    stage-wise kernels: filter values, first hits, scatter, random words
//   abi_context.inc      rt_ctx, error reporting, rt_create / rt_destroy
//   abi_scene.inc        scene / light / camera upload, the matrix-core filter's table and, for long tables, its blocks' bounds
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
// takes one of QUEUE_SLOTS counters (256 bytes apart), so renders issued back to back on different
// streams do not share one.  Every counter has an event, recorded behind the launch that uses it: a
// counter is handed out again only once that launch has finished, and a call that finds all of them
// with launches in flight fails with RT_ERR_STATE instead of resetting a counter under a running kernel.
constexpr int QUEUE_SLOTS = 16;
constexpr size_t QUEUE_SLOT_BYTES = 256;

// Counters of rt_render_noalloc: a device array of the code object instead of a hipMalloc'ed buffer.
// Every GPU holds its own copy at its own address; `scratch` itself is only a host-side handle, the
// device address comes from hipGetSymbolAddress (scratch_ring below) and is what kernels and
// hipMemsetAsync receive.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static_assert(RT_NOALLOC_SCRATCH_BYTES == QUEUE_SLOTS * QUEUE_SLOT_BYTES, "scratch holds one ring of work counters");

namespace {

struct CounterRing {
    unsigned char *base = nullptr;      // QUEUE_SLOTS counters, QUEUE_SLOT_BYTES apart, in device memory
    hipEvent_t done[QUEUE_SLOTS] = {};  // recorded behind the last launch that used the counter ...
    hipStream_t last[QUEUE_SLOTS] = {}; // ... on this stream
    bool used[QUEUE_SLOTS] = {};
    unsigned next = 0;
};

// Events of a ring, created once (with the ring's device current).
hipError_t ring_create_events(CounterRing &r) {
    for (int i = 0; i < QUEUE_SLOTS; ++i)
        if (!r.done[i]) {
            hipError_t e = hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

void ring_destroy_events(CounterRing &r) {
    for (int i = 0; i < QUEUE_SLOTS; ++i)
        if (r.done[i]) {
            (void)hipEventDestroy(r.done[i]);
            r.done[i] = nullptr;
        }
}

// A counter that the launch about to be issued on `st` may reset, searched from the one after the last handed out:
// one whose last launch ran on `st` itself (stream order puts the reset behind that launch, however many frames a
// caller queues on one stream), else one whose last launch has finished; -1 when every counter belongs to a launch
// still running on another stream.  A stream under graph capture can neither be asked about its events nor
// record one outside the graph, so `ask` is false there and the ring is used round robin, unchecked.
int ring_acquire(CounterRing &r, bool ask, hipStream_t st) {
    for (int k = 0; k < QUEUE_SLOTS; ++k) {
        const int i = (int)((r.next + (unsigned)k) % QUEUE_SLOTS);
        if (!ask || !r.used[i] || r.last[i] == st || hipEventQuery(r.done[i]) == hipSuccess) {
            r.next = (unsigned)i + 1u;
            return i;
        }
        (void)hipGetLastError(); // hipErrorNotReady is an answer, not a failure
    }
    return -1;
}

// The launch that uses counter `slot` has been issued on `st`.
hipError_t ring_mark(CounterRing &r, int slot, hipStream_t st) {
    r.used[slot] = true;
    r.last[slot] = st;
    return hipEventRecord(r.done[slot], st);
}

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;               // guards the rings below from hand-out to the event behind the launch
CounterRing g_scratch_ring[MAX_DEVICES];  // per device: its copy of `scratch`, null until looked up

// The ring in `scratch` on `device`: address looked up and events created once per device, with that
// device current (at rt_create, so that rt_render_noalloc itself creates nothing).
hipError_t scratch_ring(int device, CounterRing **out) {
    *out = nullptr;
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    CounterRing &r = g_scratch_ring[device];
    if (!r.base) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        e = ring_create_events(r);
        if (e != hipSuccess) return e;
        r.base = static_cast<unsigned char *>(ptr);
    }
    *out = &r;
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
    if (!ctx->scratch_ring || !ctx->scratch_ring->base)
        return fail(ctx, RT_ERR_STATE, "rt_render_noalloc: the device's scratch array was not found");
    // the counter is 32 bits wide: a call may cover at most 2^31 - 1 work items (tiles of 1 to 64 pixels)
    if (p && p->width > 0 && p->height > 0 && (long long)p->width * (long long)p->height > 0x7fffffffLL)
        return fail(ctx, RT_ERR_INVALID, "rt_render_noalloc: more work items than the 32-bit counter in scratch holds");
    // every context on this GPU shares the ring: one caller at a time from hand-out to the event behind its launch
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return render_impl(ctx, p, d_rgba, d_accum, d_rays, stream, *ctx->scratch_ring);
}
