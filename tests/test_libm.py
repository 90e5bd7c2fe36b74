"""sinf / cosf: the device evaluates the HOST libm's functions bit for bit (csrc/tdr_sincosf.h restates glibc's
double-precision algorithm; the reference calls std::sin / std::cos on floats at src/state_particle.cpp:58 and
src/top_down_map.cpp:381-385, and cell indices are rounded products of the results).

CPU: the restatement (tdr_sincosf_host, the variant tdr_libm_variant() selects) against this machine's libm on strided
sweeps of ALL float bit patterns plus every argument on which glibc's two builds differ.  GPU: the device against the
same host functions.  A full 2^32 sweep of both functions (12 s on 8 cores, 0 mismatches) was run once with the
program in the docstring of tools/libm_sweep.cpp."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

# arguments on which glibc's plain and FMA-contracted builds round differently (csrc/tdr_core.hip)
SIN_PROBE = [0x4255b0a9, 0x42a35c07, 0x42a35d44, 0x42a97360, 0x42cf5854, 0x42e87a55]
COS_PROBE = [0x418a3adb, 0x418a3adc, 0x418a3add, 0x418a3ade, 0x41bc76d9, 0x4202eb4b, 0x42687a55, 0x4280ce28,
             0x42870e40, 0x42c55faa, 0x42d8d23e]


def _host_libm(x):
    """sinf / cosf of the host's libm, element by element (numpy's own float32 sin/cos are SIMD kernels of their own)."""
    libm = C.CDLL(ctypes.util.find_library("m"))
    out = []
    for name in ("sinf", "cosf"):
        fn = getattr(libm, name)
        fn.restype, fn.argtypes = C.c_float, [C.c_float]
        out.append(np.fromiter((fn(float(v)) for v in x), np.float32, len(x)))
    return out


def _args(step, offset):
    bits = np.arange(offset, 1 << 32, step, dtype=np.uint64).astype(np.uint32)
    probes = np.asarray(SIN_PROBE + COS_PROBE, np.uint32)
    bits = np.concatenate([bits, probes, probes | np.uint32(0x80000000),
                           np.asarray([0, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0x3f490fdb, 0x42f00000,
                                       0x42efffff, 0x39800000, 0x397fffff, 0x00800000, 0x007fffff], np.uint32)])
    return bits.view(np.float32)


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or \
        bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_libm_variant_is_recognised_and_restatement_matches_host():
    from top_down_renderer_amd import _lib
    L = _lib.load()
    v = L.tdr_libm_variant()
    assert v in (0, 1), "the host libm's sinf / cosf is neither build of glibc's algorithm: indices from sin/cos unpinned"
    x = _args(step=(1 << 32) // 300_000 + 1, offset=12345)
    s, c = np.empty_like(x), np.empty_like(x)
    assert L.tdr_sincosf_host(x.ctypes.data_as(C.c_void_p), len(x), v, s.ctypes.data_as(C.c_void_p),
                              c.ctypes.data_as(C.c_void_p)) == 0
    hs, hc = _host_libm(x)
    assert _same(s, hs) and _same(c, hc)
    # the other build differs on the probe arguments (that is what makes the probe a probe)
    p = np.asarray(SIN_PROBE, np.uint32).view(np.float32)
    s0, s1 = np.empty_like(p), np.empty_like(p)
    L.tdr_sincosf_host(p.ctypes.data_as(C.c_void_p), len(p), 0, s0.ctypes.data_as(C.c_void_p), None)
    L.tdr_sincosf_host(p.ctypes.data_as(C.c_void_p), len(p), 1, s1.ctypes.data_as(C.c_void_p), None)
    assert np.all(s0.view(np.uint32) != s1.view(np.uint32))


@pytest.mark.gpu
def test_device_sincos_is_the_host_libm_bit_for_bit():
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    L = k.lib
    assert L.tdr_libm_variant() in (0, 1)
    # (a) against the host libm itself on 400k arguments spread over all bit patterns + the probes
    x = _args(step=(1 << 32) // 400_000 + 1, offset=777)
    xd, s, c = k.to_device(x), k.zeros((len(x),)), k.zeros((len(x),))
    assert L.tdr_k_selftest_sincos(C.c_void_p(xd.data_ptr()), len(x), C.c_void_p(s.data_ptr()), C.c_void_p(c.data_ptr()),
                                   k.stream()) == 0
    hs, hc = _host_libm(x)
    assert _same(s.cpu().numpy(), hs) and _same(c.cpu().numpy(), hc)
    # (b) against the host restatement (itself checked against libm above) on 2^26 arguments, both variants
    x = _args(step=64, offset=5)
    xd, s, c = k.to_device(x), k.zeros((len(x),)), k.zeros((len(x),))
    rs, rc = np.empty_like(x), np.empty_like(x)
    try:
        for v in (0, 1):
            assert L.tdr_libm_force_variant(v) == 0
            assert L.tdr_k_selftest_sincos(C.c_void_p(xd.data_ptr()), len(x), C.c_void_p(s.data_ptr()),
                                           C.c_void_p(c.data_ptr()), k.stream()) == 0
            assert L.tdr_sincosf_host(x.ctypes.data_as(C.c_void_p), len(x), v, rs.ctypes.data_as(C.c_void_p),
                                      rc.ctypes.data_as(C.c_void_p)) == 0
            assert _same(s.cpu().numpy(), rs) and _same(c.cpu().numpy(), rc), f"variant {v}"
    finally:
        L.tdr_libm_force_variant(-1)


def _host_logf(x):
    libm = C.CDLL(ctypes.util.find_library("m"))
    fn = libm.logf
    fn.restype, fn.argtypes = C.c_float, [C.c_float]
    with np.errstate(all="ignore"):
        return np.fromiter((fn(float(v)) for v in x), np.float32, len(x))


def test_logf_restatement_matches_the_host_libm():
    """csrc/tdr_logf.h against this machine's logf on 300 000 arguments spread over every bit pattern (zeros, subnormals,
    negatives, infinities, NaN included) and densely over (0, 1], where the polar method's r2 lives.  The sweep of all 2^32
    arguments (tools/logf_sweep.cpp) finds 0 mismatches, and glibc's plain and FMA builds agree everywhere."""
    from top_down_renderer_amd import _lib
    L = _lib.load()
    x = np.concatenate([_args(step=(1 << 32) // 200_000 + 1, offset=4321),
                        np.random.default_rng(7).random(100_000, dtype=np.float32),
                        np.asarray([1.0, np.nextafter(np.float32(1), np.float32(0)), 2.0 ** -48, 1e-38, 1e-45], np.float32)])
    out = np.empty_like(x)
    assert L.tdr_logf_host(x.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p)) == 0
    assert _same(out, _host_logf(x))


@pytest.mark.gpu
def test_device_logf_is_the_host_libm_bit_for_bit():
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    x = np.concatenate([_args(step=(1 << 32) // 300_000 + 1, offset=99),
                        np.random.default_rng(8).random(200_000, dtype=np.float32)])
    xd, o = k.to_device(x), k.zeros((len(x),))
    assert k.lib.tdr_k_selftest_logf(C.c_void_p(xd.data_ptr()), len(x), C.c_void_p(o.data_ptr()), k.stream()) == 0
    assert _same(o.cpu().numpy(), _host_logf(x))
    # and the host restatement on 2^26 arguments
    x = _args(step=64, offset=9)
    xd, o = k.to_device(x), k.zeros((len(x),))
    ref = np.empty_like(x)
    assert k.lib.tdr_logf_host(x.ctypes.data_as(C.c_void_p), len(x), ref.ctypes.data_as(C.c_void_p)) == 0
    assert k.lib.tdr_k_selftest_logf(C.c_void_p(xd.data_ptr()), len(x), C.c_void_p(o.data_ptr()), k.stream()) == 0
    assert _same(o.cpu().numpy(), ref)
