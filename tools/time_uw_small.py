#!/usr/bin/env python3
"""Launch time of the one-workgroup weight statistics (uw_small_kernel) and of the running sum of the resample, at the
reference's operating point and around it; both evaluations of the chains (tdr_config_uw_waves).
usage: python tools/time_uw_small.py (GPU box)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from top_down_renderer_amd.kernels import HipKernels

k = HipKernels()
for n in (300, 1000, 5000, 20000, 32768):
    rng = np.random.default_rng(n)
    raw = (1.0 / (rng.random(n) * 20 + 0.15)).astype(np.float32)
    raw[rng.integers(0, n, n // 50)] = np.nan
    ld = rng.uniform(0, 0.5, n).astype(np.float32)
    raw_d, ld_d = k.to_device(raw), k.to_device(ld)
    w, info, runmax = k.empty((n,)), k.empty((65536,)), k.empty((n,))
    line = f"n={n:6d}:"
    for on in (0, 1):
        k.lib.tdr_config_uw_waves(on)
        for _ in range(5):
            k.update_weights(raw_d, ld_d, n, w, info)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps):
            k.update_weights(raw_d, ld_d, n, w, info)
        e1.record()
        k.synchronize()
        line += f"  statistics ({'waves' if on else 'workgroup'}) {e0.elapsed_time(e1) / reps * 1e3:7.1f} us"
    for on in (0, 1):
        k.lib.tdr_config_prefix_small(on)
        for _ in range(5):
            k.prefix(w, n, runmax)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            k.prefix(w, n, runmax)
        e1.record()
        k.synchronize()
        line += f"  running sum ({'one launch' if on else 'as before'}) {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us"
    print(line)
