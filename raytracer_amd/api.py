"""ctypes binding of the C ABI in include/rt_amd.h.

There is no fallback: if the HIP library is missing or a call fails, an
exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")  # RT_AMD_LIB: tuning builds

STRIP_ROWS = 8
FLAG_ACCUMULATE = 1
FLAG_DIRECT_LIGHT = 2
FH_REFINE = 1
FH_EXPANDED = 2

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_EMISSIVE = 0, 1, 2, 3

# numpy views of the ABI structs
SPHERE_DTYPE = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("r", "<f4")])
MATERIAL_DTYPE = np.dtype(
    [("type", "<u4"), ("albedo", "<f4", 3), ("fuzz", "<f4"), ("ior", "<f4"), ("emission", "<f4", 3), ("diffuse", "<f4", 3)]
)
LIGHT_DTYPE = np.dtype([("position", "<f4", 3), ("ambient", "<f4", 3), ("diffuse", "<f4", 3), ("specular", "<f4", 3),
                        ("view_pow", "<f4"), ("_pad", "<f4", 3)])
PRIM_DTYPE = np.dtype([("type", "<u4"), ("mat", "<u4"), ("first_vertex", "<u4"), ("n_vertices", "<u4"), ("p", "<f4", 8),
                       ("selected", "<u4"), ("_pad", "<u4", 3)])
PRIM_SPHERE, PRIM_PLANE, PRIM_CUBE, PRIM_POLYGON = 0, 1, 2, 3
CAMERA_DTYPE = np.dtype([("center", "<f4", 3), ("upper_left", "<f4", 3), ("right", "<f4", 3), ("down", "<f4", 3)])
assert SPHERE_DTYPE.itemsize == 16 and MATERIAL_DTYPE.itemsize == 48 and CAMERA_DTYPE.itemsize == 48
assert LIGHT_DTYPE.itemsize == 64 and PRIM_DTYPE.itemsize == 64


class RtError(RuntimeError):
    pass


class _Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("samples_per_pixel", C.c_int32),
        ("max_depth", C.c_int32),
        ("sample_offset", C.c_int32),
        ("seed", C.c_uint32),
        ("flags", C.c_uint32),
        ("strip_first", C.c_int32),
        ("strip_stride", C.c_int32),
    ]


class _GroupTiming(C.Structure):
    _fields_ = [("kernel_ms_min", C.c_float), ("kernel_ms_max", C.c_float), ("gather_ms", C.c_float),
                ("total_ms", C.c_float)]


@dataclass
class RenderParams:
    """CameraRenderProperties (reference inc/Camera.h:14-22) + frame geometry + strip partition."""

    width: int
    height: int
    samples_per_pixel: int = 3
    max_depth: int = 10
    sample_offset: int = 0
    seed: int = 0
    flags: int = 0
    strip_first: int = 0
    strip_stride: int = 1

    def to_c(self) -> _Params:
        return _Params(
            self.width, self.height, self.samples_per_pixel, self.max_depth, self.sample_offset,
            self.seed & 0xFFFFFFFF, self.flags, self.strip_first, self.strip_stride,
        )


EXPORTS = [
    "rt_version", "rt_rng4", "rt_rng4_device_host", "rt_output_rows", "rt_create", "rt_destroy", "rt_last_error",
    "rt_upload_scene", "rt_upload_scene_ex", "rt_upload_lights", "rt_set_camera", "rt_render", "rt_render_noalloc",
    "rt_render_host", "rt_last_kernel_ms", "rt_last_kernel_variant", "rt_last_scans", "rt_last_blocks", "rt_device_allocations", "rt_filter_describe", "rt_filter_values_host", "rt_filter_tables_host",
    "rt_group_create", "rt_group_destroy", "rt_group_last_error", "rt_group_size", "rt_group_transport",
    "rt_group_device_allocations", "rt_group_upload_scene_ex", "rt_group_upload_lights", "rt_group_set_camera",
    "rt_group_render", "rt_group_render_host",
    "rt_first_hit", "rt_first_hit_host", "rt_first_hit_ex", "rt_first_hit_ex_host", "rt_scatter_host",
]

class _FilterInfo(C.Structure):
    _fields_ = [("in_use", C.c_int32), ("center", C.c_float * 3), ("radius", C.c_float), ("margin", C.c_float),
                ("margin_far", C.c_float), ("n_rows", C.c_int32), ("n_big", C.c_int32), ("big", C.c_int32 * 8),
                ("n_table_rows", C.c_int32)]


class _FilterTablesInfo(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("scale", C.c_float), ("bound_scale", C.c_float * 3), ("n_rows", C.c_int32),
                ("n_blocks", C.c_int32), ("n_used", C.c_int32), ("n_bound_blocks", C.c_int32), ("n_big", C.c_int32),
                ("big", C.c_int32 * 8)]


_lib = None


def _proto(lib, name, field, value):
    """Declares a prototype; a library built from an older source (tuning comparisons through RT_AMD_LIB)
    may lack the newer entry points, which then fail when called, not when the library is loaded."""
    fn = getattr(lib, name, None)
    if fn is not None:
        setattr(fn, field, value)


def load_library(path: str | None = None) -> C.CDLL:
    """Load librt_amd.so and declare prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RtError(
            f"HIP library not found at {p}: build it first (python -c 'import __graft_entry__ as g; g.build()')"
        )
    lib = C.CDLL(p)
    vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
    _proto(lib, "rt_version", "restype", C.c_int)
    _proto(lib, "rt_rng4", "argtypes", [u32, u32, u32, u32, C.POINTER(u32)])
    _proto(lib, "rt_rng4", "restype", None)
    _proto(lib, "rt_rng4_device_host", "argtypes", [vp, C.c_int, vp, vp])
    _proto(lib, "rt_rng4_device_host", "restype", C.c_int)
    _proto(lib, "rt_filter_describe", "argtypes", [vp, C.POINTER(_FilterInfo)])
    _proto(lib, "rt_filter_describe", "restype", C.c_int)
    _proto(lib, "rt_filter_values_host", "argtypes", [vp, C.c_int, vp, vp, vp])
    _proto(lib, "rt_filter_values_host", "restype", C.c_int)
    _proto(lib, "rt_filter_tables_host", "argtypes", [C.c_int, vp, C.POINTER(_FilterTablesInfo), vp, vp, vp])
    _proto(lib, "rt_filter_tables_host", "restype", C.c_int)
    _proto(lib, "rt_output_rows", "argtypes", [C.POINTER(_Params)])
    _proto(lib, "rt_output_rows", "restype", C.c_int)
    _proto(lib, "rt_create", "argtypes", [C.c_int, C.POINTER(vp)])
    _proto(lib, "rt_create", "restype", C.c_int)
    _proto(lib, "rt_destroy", "argtypes", [vp])
    _proto(lib, "rt_destroy", "restype", None)
    _proto(lib, "rt_last_error", "argtypes", [vp])
    _proto(lib, "rt_last_error", "restype", C.c_char_p)
    _proto(lib, "rt_upload_scene", "argtypes", [vp, C.c_int, vp, vp, C.c_int, vp])
    _proto(lib, "rt_upload_scene", "restype", C.c_int)
    _proto(lib, "rt_upload_scene_ex", "argtypes", [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp])
    _proto(lib, "rt_upload_scene_ex", "restype", C.c_int)
    _proto(lib, "rt_upload_lights", "argtypes", [vp, C.c_int, vp])
    _proto(lib, "rt_upload_lights", "restype", C.c_int)
    _proto(lib, "rt_set_camera", "argtypes", [vp, vp])
    _proto(lib, "rt_set_camera", "restype", C.c_int)
    _proto(lib, "rt_render", "argtypes", [vp, C.POINTER(_Params), vp, vp, vp, vp])
    _proto(lib, "rt_render", "restype", C.c_int)
    _proto(lib, "rt_render_noalloc", "argtypes", [vp, C.POINTER(_Params), vp, vp, vp, vp])
    _proto(lib, "rt_render_noalloc", "restype", C.c_int)
    _proto(lib, "rt_render_host", "argtypes", [vp, C.POINTER(_Params), vp, vp, C.POINTER(C.c_uint64)])
    _proto(lib, "rt_render_host", "restype", C.c_int)
    _proto(lib, "rt_last_kernel_ms", "argtypes", [vp])
    _proto(lib, "rt_last_kernel_ms", "restype", C.c_float)
    _proto(lib, "rt_last_kernel_variant", "argtypes", [vp])
    _proto(lib, "rt_last_kernel_variant", "restype", C.c_int)
    _proto(lib, "rt_last_scans", "argtypes", [vp])
    _proto(lib, "rt_last_scans", "restype", C.c_uint64)
    _proto(lib, "rt_last_blocks", "argtypes", [vp])
    _proto(lib, "rt_last_blocks", "restype", C.c_uint64)
    _proto(lib, "rt_device_allocations", "argtypes", [vp])
    _proto(lib, "rt_device_allocations", "restype", C.c_uint64)
    _proto(lib, "rt_group_create", "argtypes", [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)])
    _proto(lib, "rt_group_create", "restype", C.c_int)
    _proto(lib, "rt_group_destroy", "argtypes", [vp])
    _proto(lib, "rt_group_destroy", "restype", None)
    _proto(lib, "rt_group_last_error", "argtypes", [vp])
    _proto(lib, "rt_group_last_error", "restype", C.c_char_p)
    _proto(lib, "rt_group_size", "argtypes", [vp])
    _proto(lib, "rt_group_size", "restype", C.c_int)
    _proto(lib, "rt_group_transport", "argtypes", [vp])
    _proto(lib, "rt_group_transport", "restype", C.c_char_p)
    _proto(lib, "rt_group_device_allocations", "argtypes", [vp])
    _proto(lib, "rt_group_device_allocations", "restype", C.c_uint64)
    _proto(lib, "rt_group_upload_scene_ex", "argtypes", [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp])
    _proto(lib, "rt_group_upload_scene_ex", "restype", C.c_int)
    _proto(lib, "rt_group_upload_lights", "argtypes", [vp, C.c_int, vp])
    _proto(lib, "rt_group_upload_lights", "restype", C.c_int)
    _proto(lib, "rt_group_set_camera", "argtypes", [vp, vp])
    _proto(lib, "rt_group_set_camera", "restype", C.c_int)
    _proto(lib, "rt_group_render", "argtypes", [vp, C.POINTER(_Params), C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(_GroupTiming)])
    _proto(lib, "rt_group_render", "restype", C.c_int)
    _proto(lib, "rt_group_render_host", "argtypes", [vp, C.POINTER(_Params), vp, C.POINTER(C.c_uint64), C.POINTER(_GroupTiming)])
    _proto(lib, "rt_group_render_host", "restype", C.c_int)
    _proto(lib, "rt_first_hit", "argtypes", [vp, C.c_int, vp, vp, C.c_float, u32, vp, vp, vp, vp, vp])
    _proto(lib, "rt_first_hit", "restype", C.c_int)
    _proto(lib, "rt_first_hit_host", "argtypes", [vp, C.c_int, vp, vp, C.c_float, u32, vp, vp, vp, vp])
    _proto(lib, "rt_first_hit_host", "restype", C.c_int)
    _proto(lib, "rt_first_hit_ex", "argtypes", [vp, C.c_int, vp, vp, C.c_float, u32, vp, vp, vp, vp, vp, vp])
    _proto(lib, "rt_first_hit_ex", "restype", C.c_int)
    _proto(lib, "rt_first_hit_ex_host", "argtypes", [vp, C.c_int, vp, vp, C.c_float, u32, vp, vp, vp, vp, vp])
    _proto(lib, "rt_first_hit_ex_host", "restype", C.c_int)
    _proto(lib, "rt_scatter_host", "argtypes", [vp, C.c_int] + [vp] * 10)
    _proto(lib, "rt_scatter_host", "restype", C.c_int)
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def rng4(pixel: int, sample: int, call: int, seed: int) -> np.ndarray:
    lib = load_library()
    out = (C.c_uint32 * 4)()
    lib.rt_rng4(pixel & 0xFFFFFFFF, sample & 0xFFFFFFFF, call & 0xFFFFFFFF, seed & 0xFFFFFFFF, out)
    return np.array(list(out), dtype=np.uint32)


def output_rows(p: RenderParams) -> int:
    lib = load_library()
    cp = p.to_c()
    return int(lib.rt_output_rows(C.byref(cp)))


class Renderer:
    """One rendering context on one GPU (rt_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.rt_create(device, C.byref(h))
        if rc != 0 or not h:
            raise RtError(f"rt_create(device={device}) failed with status {rc} (is a GPU visible?)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.rt_last_error(self.h)
            raise RtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # -- scene / camera -----------------------------------------------------
    def upload_scene(self, spheres: np.ndarray, mat_id: np.ndarray, mats: np.ndarray):
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        mat_id = np.ascontiguousarray(mat_id, dtype=np.uint32)
        mats = np.ascontiguousarray(mats, dtype=MATERIAL_DTYPE)
        if len(mat_id) != len(spheres):
            raise RtError("mat_id and spheres differ in length")
        self._check(
            self.lib.rt_upload_scene(self.h, len(spheres), _ptr(spheres), _ptr(mat_id), len(mats), _ptr(mats)),
            "rt_upload_scene",
        )
        self.n_spheres = len(spheres)

    def upload_scene_ex(self, prims: np.ndarray, verts: np.ndarray | None, mats: np.ndarray):
        """General primitive list (spheres, planes, cubes, polygons, selection flags)."""
        prims = np.ascontiguousarray(prims, dtype=PRIM_DTYPE)
        mats = np.ascontiguousarray(mats, dtype=MATERIAL_DTYPE)
        nv = 0 if verts is None else len(verts)
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3) if nv else None
        self._check(
            self.lib.rt_upload_scene_ex(self.h, len(prims), _ptr(prims) if len(prims) else None, nv,
                                        _ptr(v) if nv else None, len(mats), _ptr(mats)),
            "rt_upload_scene_ex",
        )
        self.n_spheres = int(np.count_nonzero(prims["type"] == PRIM_SPHERE))

    def upload_lights(self, lights: np.ndarray | None):
        n = 0 if lights is None else len(lights)
        arr = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE) if n else None
        self._check(self.lib.rt_upload_lights(self.h, n, _ptr(arr) if n else None), "rt_upload_lights")

    def set_camera(self, cam: np.ndarray):
        cam = np.ascontiguousarray(cam, dtype=CAMERA_DTYPE).reshape(1)
        self._check(self.lib.rt_set_camera(self.h, _ptr(cam)), "rt_set_camera")

    def set_scene(self, scene):
        """scene: a scenes.Scene"""
        if getattr(scene, "needs_general", False):
            self.upload_scene_ex(scene.prims, scene.verts, scene.mats)
        else:
            self.upload_scene(scene.spheres, scene.mat_id, scene.mats)
        self.upload_lights(getattr(scene, "lights", None))
        self.set_camera(scene.camera)

    # -- render ---------------------------------------------------------------
    def render_host(self, p: RenderParams, want_accum: bool = False, accum_in: np.ndarray | None = None):
        """Render into host arrays.  Returns (rgba[rows,W,4] u8, accum[rows,W,4] f32 or None, rays)."""
        rows = output_rows(p)
        rgba = np.zeros((rows, p.width, 4), dtype=np.uint8)
        accum = None
        if accum_in is not None:
            accum = np.ascontiguousarray(accum_in, dtype=np.float32).copy()
        elif want_accum:
            accum = np.zeros((rows, p.width, 4), dtype=np.float32)
        rays = C.c_uint64(0)
        cp = p.to_c()
        self._check(
            self.lib.rt_render_host(self.h, C.byref(cp), _ptr(rgba), _ptr(accum) if accum is not None else None,
                                    C.byref(rays)),
            "rt_render_host",
        )
        return rgba, accum, int(rays.value)

    def render_device(self, p: RenderParams, rgba, accum=None, rays=None, stream=None, noalloc: bool = False):
        """Asynchronous render into torch CUDA tensors (device pointers).

        rgba: uint8 [rows, W, 4]; accum: float32 [rows, W, 4] or None; rays: int64 [1] or None.
        noalloc: go through rt_render_noalloc (work counter in the library's static device array)."""
        cp = p.to_c()
        st = C.c_void_p(stream) if stream else None
        self._check(
            (self.lib.rt_render_noalloc if noalloc else self.lib.rt_render)(
                self.h, C.byref(cp), C.c_void_p(rgba.data_ptr()),
                C.c_void_p(accum.data_ptr()) if accum is not None else None,
                C.c_void_p(rays.data_ptr()) if rays is not None else None, st,
            ),
            "rt_render",
        )

    def rng4_device(self, keys: np.ndarray) -> np.ndarray:
        """The kernels' generator evaluated on the device for keys[n, 4] = (pixel, sample, call, seed)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint32).reshape(-1, 4)
        out = np.zeros_like(keys)
        self._check(self.lib.rt_rng4_device_host(self.h, len(keys), _ptr(keys), _ptr(out)), "rt_rng4_device_host")
        return out

    def filter_describe(self) -> dict:
        """Frame and membership of the matrix-core sphere filter of the uploaded scene (n_rows == 0: none)."""
        fi = _FilterInfo()
        self._check(self.lib.rt_filter_describe(self.h, C.byref(fi)), "rt_filter_describe")
        return {"in_use": bool(fi.in_use), "center": np.array(list(fi.center), dtype=np.float32), "radius": float(fi.radius),
                "margin": float(fi.margin), "margin_far": float(fi.margin_far), "n_rows": int(fi.n_rows),
                "big": [int(fi.big[i]) for i in range(fi.n_big)], "table_rows": int(fi.n_table_rows)}

    def filter_values(self, org: np.ndarray, dir: np.ndarray) -> np.ndarray:
        """Device-evaluated filter values [n_rays, n_rows]; negative = the sphere of that row passes."""
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((len(org), self.filter_describe()["n_rows"]), dtype=np.float32)
        self._check(self.lib.rt_filter_values_host(self.h, len(org), _ptr(org), _ptr(dir), _ptr(out)),
                    "rt_filter_values_host")
        return out

    def last_kernel_ms(self) -> float:
        return float(self.lib.rt_last_kernel_ms(self.h))

    def last_kernel_variant(self) -> int:
        """Sphere source of the last render's kernel (0 LDS copy, 1 LDS tiles, 2 scalar loads, 3 matrix-core filter
        resident, 4 streamed, 5 read from memory where a block's bound is met), + 8 for the general kernel; -1 before
        the first render."""
        return int(self.lib.rt_last_kernel_variant(self.h))

    def last_scans(self) -> int:
        """Scans (closest-sphere searches for the 64 rays of a wave) the last render_host's kernel ran."""
        return int(self.lib.rt_last_scans(self.h))

    def last_blocks(self) -> int:
        """Table blocks (32 rows x the 64 rays of a wave) the scans of the last render_host's kernel walked."""
        return int(self.lib.rt_last_blocks(self.h))

    def device_allocations(self) -> int:
        return int(self.lib.rt_device_allocations(self.h))

    # -- first hit / scatter --------------------------------------------------
    def first_hit(self, org: np.ndarray, dir: np.ndarray, tmin: float = 0.001, refine: bool = False,
                  expanded_state: bool = False, want_expanded: bool = False):
        """SceneManager::hitClosest for a batch of rays.  Returns (ids, t, normal, front) and, with
        want_expanded, HitRecord::hitExpanded as a fifth array."""
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
        n = len(org)
        ids = np.full(n, -2, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        nrm = np.zeros((n, 3), dtype=np.float32)
        front = np.zeros(n, dtype=np.uint8)
        flags = (FH_REFINE if refine else 0) | (FH_EXPANDED if expanded_state else 0)
        if not (expanded_state or want_expanded):
            self._check(
                self.lib.rt_first_hit_host(self.h, n, _ptr(org), _ptr(dir), float(tmin), flags,
                                           _ptr(ids), _ptr(t), _ptr(nrm), _ptr(front)),
                "rt_first_hit_host",
            )
            return ids, t, nrm, front
        exp = np.zeros(n, dtype=np.uint8)
        self._check(
            self.lib.rt_first_hit_ex_host(self.h, n, _ptr(org), _ptr(dir), float(tmin), flags,
                                          _ptr(ids), _ptr(t), _ptr(nrm), _ptr(front), _ptr(exp)),
            "rt_first_hit_ex_host",
        )
        return (ids, t, nrm, front, exp) if want_expanded else (ids, t, nrm, front)

    def scatter(self, mat, in_dir, point, normal, front, u):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)
        mat = np.ascontiguousarray(mat, dtype=np.uint32)
        in_dir, point, normal, u = f(in_dir), f(point), f(normal), f(u)
        front = np.ascontiguousarray(front, dtype=np.uint8)
        n = len(mat)
        att = np.zeros((n, 3), np.float32)
        oo = np.zeros((n, 3), np.float32)
        od = np.zeros((n, 3), np.float32)
        flag = np.zeros(n, np.uint8)
        self._check(
            self.lib.rt_scatter_host(self.h, n, _ptr(mat), _ptr(in_dir), _ptr(point), _ptr(normal), _ptr(front),
                                     _ptr(u), _ptr(att), _ptr(oo), _ptr(od), _ptr(flag)),
            "rt_scatter_host",
        )
        return att, oo, od, flag


class Group:
    """Several GPUs of this process rendering one frame (rt_group): member i renders strips i, i+n, ...;
    the strips are gathered on the first member's GPU.  A device may be listed more than once."""

    def __init__(self, devices):
        self.lib = load_library()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.lib.rt_group_create(len(devices), arr, C.byref(h))
        if rc != 0 or not h:
            raise RtError(f"rt_group_create({list(devices)}) failed with status {rc}")
        self.h = h
        self.devices = list(devices)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.rt_group_last_error(self.h)
            raise RtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    @property
    def transport(self) -> str:
        return self.lib.rt_group_transport(self.h).decode()

    def device_allocations(self) -> int:
        return int(self.lib.rt_group_device_allocations(self.h))

    def set_scene(self, scene):
        if getattr(scene, "needs_general", False):
            prims, verts = scene.prims, scene.verts
        else:
            prims = np.zeros(len(scene.spheres), dtype=PRIM_DTYPE)
            prims["type"] = PRIM_SPHERE
            prims["mat"] = scene.mat_id
            for k, f in enumerate(("cx", "cy", "cz", "r")):
                prims["p"][:, k] = scene.spheres[f]
            verts = None
        prims = np.ascontiguousarray(prims, dtype=PRIM_DTYPE)
        mats = np.ascontiguousarray(scene.mats, dtype=MATERIAL_DTYPE)
        nv = 0 if verts is None else len(verts)
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3) if nv else None
        self._check(self.lib.rt_group_upload_scene_ex(self.h, len(prims), _ptr(prims) if len(prims) else None, nv,
                                                      _ptr(v) if nv else None, len(mats), _ptr(mats)),
                    "rt_group_upload_scene_ex")
        lights = getattr(scene, "lights", None)
        n = 0 if lights is None else len(lights)
        arr = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE) if n else None
        self._check(self.lib.rt_group_upload_lights(self.h, n, _ptr(arr) if n else None), "rt_group_upload_lights")
        cam = np.ascontiguousarray(scene.camera, dtype=CAMERA_DTYPE).reshape(1)
        self._check(self.lib.rt_group_set_camera(self.h, _ptr(cam)), "rt_group_set_camera")

    def render_host(self, p: RenderParams):
        """-> (rgba[H,W,4] u8, rays, timing dict)"""
        rgba = np.zeros((p.height, p.width, 4), dtype=np.uint8)
        rays = C.c_uint64(0)
        t = _GroupTiming()
        cp = p.to_c()
        self._check(self.lib.rt_group_render_host(self.h, C.byref(cp), _ptr(rgba), C.byref(rays), C.byref(t)),
                    "rt_group_render_host")
        return rgba, int(rays.value), {f: getattr(t, f) for f, _ in _GroupTiming._fields_}


def filter_tables(spheres: np.ndarray):
    """The matrix-core filter's tables of a sphere list (SPHERE_DTYPE), built on the host as an upload builds them; no
    GPU is touched.  -> dict with the frames, `row_index` (sphere behind every table row), `table` (uint32
    [n_blocks + 1, 2, 64, 4]: block, k-step, lane, word) and `bounds` (uint32 [n_bound_blocks + 1, 3, 64, 4], or None
    when the table stays resident in LDS), or None for a list that gets no table."""
    lib = load_library()
    sp = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    info = _FilterTablesInfo()
    rc = lib.rt_filter_tables_host(len(sp), _ptr(sp), C.byref(info), None, None, None)
    if rc != 0:
        return None
    row_index = np.zeros(info.n_rows, dtype=np.int32)
    table = np.zeros((info.n_blocks + 1, 2, 64, 4), dtype=np.uint32)
    bounds = np.zeros((info.n_bound_blocks + 1, 3, 64, 4), dtype=np.uint32) if info.n_bound_blocks else None
    rc = lib.rt_filter_tables_host(len(sp), _ptr(sp), C.byref(info), _ptr(row_index), _ptr(table), _ptr(bounds) if bounds is not None else None)
    if rc != 0:
        raise RtError(f"rt_filter_tables_host failed with code {rc}")
    return {"center": np.array(list(info.center), dtype=np.float32), "scale": np.float32(info.scale),
            "bound_scale": np.array(list(info.bound_scale), dtype=np.float32), "n_blocks": int(info.n_blocks),
            "n_used": int(info.n_used), "n_bound_blocks": int(info.n_bound_blocks),
            "big": [int(info.big[i]) for i in range(info.n_big)], "row_index": row_index, "table": table, "bounds": bounds}
