"""MI355X (gfx950) path-trace backend for the I1Va/RayTracer per-pixel render path.

The product is the C-ABI shared library built from ``csrc/rt_amd.hip`` (declared in
``include/rt_amd.h``).  This package is the Python plumbing around it: a ctypes
binding (``api``), the build-defined benchmark scenes (``scenes``) and the
multi-GPU strip partition + gather (``multi_gpu``).
"""
from .api import Group, Renderer, RtError, load_library, RenderParams  # noqa: F401
from . import scenes  # noqa: F401

__all__ = ["Group", "Renderer", "RtError", "load_library", "RenderParams", "scenes"]
