"""Soak test of the running-sum kernels: random weight vectors (dynamic range, dyadic values that force rounding ties,
zero runs, rare negatives / NaN / inf) of random length, all four implementations against numpy's sequential float32
cumsum.  usage: PYTHONPATH=. python tools/soak_prefix.py [cases=300] (GPU box)"""
import ctypes as C
import sys

import numpy as np

from top_down_renderer_amd.kernels import HipKernels

k = HipKernels()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(12345)
f32 = np.float32
bad = 0
for case in range(cases):
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 7000), rng.integers(7000, 32769), rng.integers(32769, 70000),
                        rng.integers(70000, 400000)]))
    kind = rng.integers(0, 6)
    if kind == 0:
        w = np.exp(rng.normal(0, rng.uniform(0.1, 6), n))
    elif kind == 1:
        w = rng.integers(0, 1 << int(rng.integers(1, 12)), n) * 2.0 ** -int(rng.integers(8, 30))
    elif kind == 2:
        w = rng.random(n) ** int(rng.integers(1, 8))
        w[rng.random(n) < rng.uniform(0, 0.9)] = 0
    elif kind == 3:
        w = np.where(rng.random(n) < 0.5, 2.0 ** rng.integers(-40, 3, n), rng.random(n))
    elif kind == 4:
        w = rng.random(n) - rng.uniform(0, 0.3)
    else:
        w = rng.random(n) * 10.0 ** rng.uniform(-30, 20)
    w = w.astype(f32)
    if rng.random() < 0.5 and kind != 4:
        s = w.sum(dtype=np.float64)
        if s > 0:
            w = (w / s).astype(f32)
    if rng.random() < 0.1:
        w[rng.integers(0, n)] = rng.choice([np.nan, np.inf, -1.0, 1e30])
    with np.errstate(all="ignore"):
        ref = np.cumsum(w, dtype=f32)
        refmax = np.maximum.accumulate(np.where(np.isnan(ref), -np.inf, ref)).astype(f32)
    wd = k.to_device(w)
    ws = k.prefix_workspace(n)
    for mode in (0, 1, 2, 3):   # (2: the chunk walk, its first 32 768 weights through the one-launch kernel; 3: that kernel)
        if mode == 3 and n > 32768:
            continue
        rm, pf = k.zeros((n,)), k.zeros((n,))
        rc = k.lib.tdr_k_prefix_mode(C.c_void_p(wd.data_ptr()), n, mode, C.c_void_p(rm.data_ptr()),
                                     C.c_void_p(pf.data_ptr()) if mode else None, C.c_void_p(ws.data_ptr()), k.stream())
        assert rc == 0
        ok = np.array_equal(rm.cpu().numpy(), refmax) and (mode == 0 or np.array_equal(pf.cpu().numpy(), ref, equal_nan=True))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} n={n} kind={kind} mode={mode}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
