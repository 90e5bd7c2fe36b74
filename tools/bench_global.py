"""Times the global-N stages every rank runs after the all-gather (statistics, exact prefix, resample slice, gather)
for particle counts up to the 8-GPU configurations.  GPU only; prints one line per n."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from top_down_renderer_amd.kernels import HipKernels  # noqa: E402


def timeit(fn, k, reps=10):
    fn(); k.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    k.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    k = HipKernels()
    rng = np.random.default_rng(0)
    world = 8
    for n in [100_000, 200_000, 400_000, 800_000, 1_000_000, 2_000_000]:
        raw = np.exp(rng.normal(0, 1.5, n)).astype(np.float32)
        raw[rng.random(n) < 0.05] = np.nan
        ld = np.abs(rng.normal(0, 1, n)).astype(np.float32)
        raw_d, ld_d = k.to_device(raw), k.to_device(ld)
        w, runmax, info = k.zeros((n,)), k.zeros((n,)), k.zeros((65536,))
        nl = n // world
        idx = k.zeros((nl,), torch.int32)
        src = k.to_device(rng.random((world * 7 * nl,)).astype(np.float32))
        dst = k.zeros((7, nl))
        t_uw = timeit(lambda: k.update_weights(raw_d, ld_d, n, w, info), k)
        t_px = timeit(lambda: k.prefix(w, n, runmax), k)
        t_rs = timeit(lambda: k.resample(runmax, n, n, 0.37, 0, nl, idx), k)
        t_ga = timeit(lambda: k.gather_states(src, idx, nl, dst, src_shard=nl), k)
        print(f"n={n:8d} update_weights {t_uw:.3f} ms  prefix {t_px:.3f} ms  resample(n/8) {t_rs:.3f} ms  "
              f"gather(n/8) {t_ga:.3f} ms", flush=True)


if __name__ == "__main__":
    sys.exit(main())
