"""rt_render_noalloc (work counter in the library's static device array) against rt_render, through
the C ABI: byte-identical frames, radiance sums and ray counts on two calls in a row, the second with
a smaller input.  The entry point has no counterpart in the reference; bar: bit-exact (integer/byte
outputs of the same kernel)."""
import numpy as np
import pytest

from raytracer_amd import Renderer, RenderParams, scenes

pytestmark = pytest.mark.gpu


class _DevBuf:
    """A device buffer from the HIP runtime the library itself loaded (no torch in this test: importing
    it after the library would bring a second HIP runtime into the process)."""

    def __init__(self, hip, nbytes):
        import ctypes as C

        self.hip, self.nbytes, self.ptr = hip, nbytes, C.c_void_p()
        assert hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0
        assert hip.hipMemset(self.ptr, 0, C.c_size_t(nbytes)) == 0

    def data_ptr(self):
        return self.ptr.value

    def to_numpy(self, dtype, shape):
        import ctypes as C

        out = np.zeros(shape, dtype=dtype)
        assert out.nbytes == self.nbytes
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(self.nbytes), 2) == 0  # D2H
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def _render(r, p, noalloc):
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")  # already in the process: the library links against it
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    n = p.width * p.height
    rgba, accum, rays = _DevBuf(hip, n * 4), _DevBuf(hip, n * 16), _DevBuf(hip, 8)
    r.render_device(p, rgba, accum, rays, None, noalloc=noalloc)
    assert hip.hipDeviceSynchronize() == 0
    out = (rgba.to_numpy(np.uint8, (p.height, p.width, 4)), accum.to_numpy(np.float32, (p.height, p.width, 4)),
           int(rays.to_numpy(np.int64, (1,))[0]))
    for b in (rgba, accum, rays):
        b.free()
    return out


CASES = [("random_spheres", 200, 120, 24, 50), ("small_bubble", 96, 54, 16, 10)]  # the second is the smaller input


def test_noalloc_matches_render_on_two_calls_in_a_row():
    with Renderer(0) as r:
        def run(noalloc):
            out = []
            for name, W, H, spp, depth in CASES:
                r.set_scene(scenes.by_name(name))
                out.append(_render(r, RenderParams(W, H, samples_per_pixel=spp, max_depth=depth, seed=7), noalloc))
            return out

        got = run(noalloc=True)  # two calls of the new entry point in a row
        want = run(noalloc=False)
        for (name, *_), g, w in zip(CASES, got, want):
            assert np.array_equal(g[0], w[0]), f"{name}: RGBA8 frames differ"
            assert np.array_equal(g[1], w[1]), f"{name}: radiance sums differ"
            assert g[2] == w[2] and g[2] > 0
            assert g[0][..., :3].any(), "frame is black"


def test_renders_in_flight_on_two_streams_do_not_share_a_work_counter():
    """rt_render is asynchronous on the caller's stream; every launch takes the next of a ring of work
    counters, so two renders issued back to back on two streams of one context (double-buffered frames)
    both come out whole.  Same for rt_render_noalloc and its static ring."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    W, H = 256, 144
    pa = RenderParams(W, H, samples_per_pixel=48, max_depth=50, seed=1)
    pb = RenderParams(W, H, samples_per_pixel=48, max_depth=50, seed=2)
    with Renderer(0) as r:
        r.set_scene(scenes.by_name("random_spheres"))
        want_a, want_b = _render(r, pa, False)[0], _render(r, pb, False)[0]
        streams = [C.c_void_p(), C.c_void_p()]
        for st in streams:
            assert hip.hipStreamCreate(C.byref(st)) == 0
        for noalloc in (False, True):
            bufs = [_DevBuf(hip, W * H * 4), _DevBuf(hip, W * H * 4)]
            for rep in range(3):  # several pairs: the two kernels overlap on the GPU
                r.render_device(pa, bufs[0], None, None, streams[0].value, noalloc=noalloc)
                r.render_device(pb, bufs[1], None, None, streams[1].value, noalloc=noalloc)
            assert hip.hipDeviceSynchronize() == 0
            got_a = bufs[0].to_numpy(np.uint8, (H, W, 4))
            got_b = bufs[1].to_numpy(np.uint8, (H, W, 4))
            assert np.array_equal(got_a, want_a) and np.array_equal(got_b, want_b), f"noalloc={noalloc}"
            for b in bufs:
                b.free()
        for st in streams:
            hip.hipStreamDestroy(st)


def test_a_seventeenth_render_in_flight_is_refused_not_corrupted():
    """Sixteen work counters per ring, each handed out again only when the launch that used it has finished: of
    seventeen long renders issued at once on seventeen streams, sixteen come out whole and the seventeenth is
    either refused with RT_ERR_STATE (nothing launched, nothing written) or, if a counter had come free by then,
    whole as well.  Never a frame with tiles missing or rendered twice.  Same for rt_render_noalloc's ring."""
    import ctypes as C

    from raytracer_amd import RtError

    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    W, H, N = 640, 360, 17  # about 5 ms of GPU each: the first is still running when the last is asked for
    with Renderer(0) as r:
        r.set_scene(scenes.by_name("random_spheres"))
        params = [RenderParams(W, H, samples_per_pixel=512, max_depth=50, seed=100 + i) for i in range(N)]
        want = [_render(r, p, False)[0] for p in params]  # one at a time
        streams = [C.c_void_p() for _ in range(N)]
        for st in streams:
            assert hip.hipStreamCreate(C.byref(st)) == 0
        for noalloc in (False, True):
            bufs = [_DevBuf(hip, W * H * 4) for _ in range(N)]
            refused = []
            for i in range(N):
                try:
                    r.render_device(params[i], bufs[i], None, None, streams[i].value, noalloc=noalloc)
                except RtError as e:
                    assert "(-4)" in str(e) and "in flight" in str(e), str(e)
                    refused.append(i)
            assert hip.hipDeviceSynchronize() == 0
            assert len(refused) <= 1 and all(i == N - 1 for i in refused), refused
            for i in range(N):
                got = bufs[i].to_numpy(np.uint8, (H, W, 4))
                if i in refused:
                    assert not got.any(), "a refused render wrote pixels"
                else:
                    assert np.array_equal(got, want[i]), f"noalloc={noalloc}: frame {i} of {N} in flight differs"
            # with everything finished the ring is free again
            r.render_device(params[0], bufs[0], None, None, streams[0].value, noalloc=noalloc)
            assert hip.hipDeviceSynchronize() == 0
            assert np.array_equal(bufs[0].to_numpy(np.uint8, (H, W, 4)), want[0])
            print(f"noalloc={noalloc}: refused {refused}")
            for b in bufs:
                b.free()
        for st in streams:
            hip.hipStreamDestroy(st)


def test_any_number_of_renders_may_be_queued_on_one_stream():
    """Stream order already keeps a counter's reset behind the launch that used it last on the same stream: forty
    frames queued on one stream without a wait in between all render (an interactive loop or a benchmark does
    this), through both entry points."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    W, H = 480, 270
    with Renderer(0) as r:
        r.set_scene(scenes.by_name("random_spheres"))
        st = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(st)) == 0
        for noalloc in (False, True):
            buf = _DevBuf(hip, W * H * 4)
            for i in range(40):
                r.render_device(RenderParams(W, H, samples_per_pixel=64, max_depth=50, seed=i), buf, None, None, st.value,
                                noalloc=noalloc)
            assert hip.hipDeviceSynchronize() == 0
            got = buf.to_numpy(np.uint8, (H, W, 4))
            buf.free()
            want = _render(r, RenderParams(W, H, samples_per_pixel=64, max_depth=50, seed=39), False)[0]
            assert np.array_equal(got, want), f"noalloc={noalloc}"
        hip.hipStreamDestroy(st)


def test_noalloc_rejects_what_it_cannot_count():
    from raytracer_amd import RtError

    with Renderer(0) as r:
        r.set_scene(scenes.small_scene())
        class Fake:
            def data_ptr(self):
                return 256  # never dereferenced: the call is refused before any launch

        with pytest.raises(RtError):
            # 65536 x 65536 pixels: more work items than the 32-bit counter holds
            r.render_device(RenderParams(65536, 65536, 1, 1), Fake(), None, None, None, noalloc=True)
