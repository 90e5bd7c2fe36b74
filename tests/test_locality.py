"""tdr_k_locality_order: the order scoring launches process particles in (csrc/tdr_filter.hip).  Any permutation gives the
same weights; what the order promises is locality — particles sorted by the Morton code of their half-cell position (the
key kernel + the device-wide radix sort; a one-workgroup form for small filters was measured and not kept).
Run with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ST_INIT_X, ST_INIT_Y, ST_DX, ST_DY, ST_THETA, ST_SCALE = 0, 1, 2, 3, 4, 5


@pytest.fixture(scope="module")
def k():
    import torch
    from top_down_renderer_amd.kernels import HipKernels

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return HipKernels()


def _spread(v):
    v = v.astype(np.uint32) & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def _half_cells(st, n, rows, cols):
    sc = st[ST_SCALE, :n]
    cx = st[ST_DX, :n] * sc + st[ST_INIT_X, :n]
    cy = st[ST_DY, :n] * sc + st[ST_INIT_Y, :n]
    cx = np.where(np.isnan(cx), np.float32(0), cx)
    cy = np.where(np.isnan(cy), np.float32(0), cy)
    hx = np.minimum(np.maximum(cx * np.float32(2), np.float32(0)), np.float32(2 * cols - 1)).astype(np.int64)
    hy = np.minimum(np.maximum(cy * np.float32(2), np.float32(0)), np.float32(2 * rows - 1)).astype(np.int64)
    return hx, hy


@pytest.mark.parametrize("n,kind", [(1, "cloud"), (63, "cloud"), (1000, "cloud"), (4097, "uniform"), (20000, "cloud"),
                                    (20000, "uniform"), (24577, "cloud"), (32768, "uniform"), (32769, "uniform"),
                                    (70001, "cloud")])
def test_order_is_a_permutation_sorted_by_the_morton_key(k, n, kind):
    import torch
    from top_down_renderer_amd.kernels import check
    rows, cols = 1500, 2000
    rng = np.random.default_rng(n)
    cap = (n + 63) // 64 * 64 + 64
    st = np.zeros((8, cap), np.float32)
    if kind == "cloud":
        st[ST_INIT_X, :n] = rng.normal(900.0, 15.0, n)
        st[ST_INIT_Y, :n] = rng.normal(400.0, 9.0, n)
        far = rng.random(n) < 0.1   # ... and a tenth of the particles anywhere (the bench mix)
        st[ST_INIT_X, :n][far] = rng.uniform(-50, cols + 50, int(far.sum()))
        st[ST_INIT_Y, :n][far] = rng.uniform(-50, rows + 50, int(far.sum()))
    else:
        st[ST_INIT_X, :n] = rng.uniform(-50, cols + 50, n)
        st[ST_INIT_Y, :n] = rng.uniform(-50, rows + 50, n)
    st[ST_DX, :n] = rng.normal(0, 2, n)
    st[ST_DY, :n] = rng.normal(0, 2, n)
    st[ST_SCALE, :n] = rng.uniform(0.8, 1.2, n)
    if n > 10:
        st[ST_INIT_X, 3] = np.nan
        st[ST_DY, 7] = np.nan
    dev = k.to_device(st)
    perm = k.zeros((cap,), torch.int32)
    perm.fill_(-5)
    tmp = k.zeros((int(k.lib.tdr_locality_tmp_ints(n, rows, cols)),), torch.int32)
    check(k.lib.tdr_k_locality_order(C.c_void_p(dev.data_ptr()), cap, n, rows, cols, C.c_void_p(perm.data_ptr()),
                                     C.c_void_p(tmp.data_ptr()), k.stream()))
    k.synchronize()
    got = perm.cpu().numpy()
    assert (got[n:] == -5).all()
    assert np.array_equal(np.sort(got[:n]), np.arange(n))
    hx, hy = _half_cells(st, n, rows, cols)
    key = _spread(hx) | (_spread(hy) << 1)
    along = key[got[:n]].astype(np.int64)
    assert (np.diff(along) >= 0).all()
