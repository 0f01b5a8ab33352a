"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/rt_amd.h
declares, and its host-side helpers agree with the oracle.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from raytracer_amd.api import EXPORTS, load_library

    lib = load_library()
    names = declared_functions()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rt_amd.h but not exported"
    assert sorted(EXPORTS) == names
    assert lib.rt_version() >= 100


def test_struct_layouts_match_header():
    from raytracer_amd.api import CAMERA_DTYPE, LIGHT_DTYPE, MATERIAL_DTYPE, SPHERE_DTYPE, _Params

    assert SPHERE_DTYPE.itemsize == 16
    assert MATERIAL_DTYPE.itemsize == 48 and MATERIAL_DTYPE.fields["emission"][1] == 24
    assert MATERIAL_DTYPE.fields["diffuse"][1] == 36
    assert CAMERA_DTYPE.itemsize == 48 and LIGHT_DTYPE.itemsize == 64
    assert ctypes.sizeof(_Params) == 36


def test_rng_matches_oracle_and_kats(oracle64):
    from raytracer_amd.api import rng4

    rng = np.random.default_rng(0)
    for _ in range(200):
        key = [int(x) for x in rng.integers(0, 2 ** 32, 4)]
        assert np.array_equal(rng4(*key), oracle64.rng4(*key))


def test_output_rows():
    from raytracer_amd import RenderParams
    from raytracer_amd.api import output_rows
    from raytracer_amd.multi_gpu import local_rows

    for H in (1, 7, 8, 9, 675, 2160):
        assert output_rows(RenderParams(16, H)) == H
        for world in (2, 3, 8):
            rows = [output_rows(RenderParams(16, H, strip_first=r, strip_stride=world)) for r in range(world)]
            assert rows == [local_rows(H, r, world) for r in range(world)]
            assert sum(rows) == ((H + 7) // 8) * 8


def test_missing_library_fails_loudly(tmp_path):
    import pytest

    from raytracer_amd.api import RtError, load_library

    with pytest.raises(RtError, match="not found"):
        load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_amd/ may import, load or link it."""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|liborc|rt_oracle|orc_[a-z_]+\s*\()")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(text), f"{f} uses the oracle"
