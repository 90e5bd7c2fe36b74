#!/usr/bin/env python3
"""Phase times of uw_small_kernel (the statistics of ParticleFilter::update for n <= 32768).  Needs the diagnostic build:
    python -c "from top_down_renderer_amd import build; build.build(force=True, extra_flags=['-DTDR_UW_TIMELINE'])"
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from top_down_renderer_amd.kernels import HipKernels

k = HipKernels()
lib = k.lib
names = ["count valid", "chain sum: head", "chain sum: chunks", "count under", "chain stddev: head", "chain stddev: chunks",
         "fill + sum", "normalise 1 + sum", "normalise 2 + argmax"]
for n in (1000, 5000, 20000, 32768):
    rng = np.random.default_rng(n)
    raw = rng.uniform(0.01, 1.0, n).astype(np.float32)
    raw[rng.integers(0, n, n // 50)] = np.nan
    ld = rng.uniform(0, 0.5, n).astype(np.float32)
    raw_d, ld_d = k.to_device(raw), k.to_device(ld)
    w, info = k.empty((n,)), k.empty((8,))
    for _ in range(3):
        k.update_weights(raw_d, ld_d, n, w, info)
    k.synchronize()
    tl = (C.c_ulonglong * 16)()
    assert lib.tdr_debug_read_uw_timeline(tl) == 0
    t = [tl[i] for i in range(10)]
    print(f"n={n}: total {(t[9] - t[0]) / 100:.1f} us")
    for i, nm in enumerate(names):
        print(f"   {nm:24s} {(t[i + 1] - t[i]) / 100:7.1f} us")
    print(f"   walk iterations: sum chain {tl[10]}, stddev chain {tl[11]}; chunks (both chains): predicted {tl[12]}, "
          f"crossing chunks lane by lane {tl[13]}, carried through {tl[14]}")
