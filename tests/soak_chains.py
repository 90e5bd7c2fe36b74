"""Soak test of the weight statistics: random raw-weight vectors (dynamic range, dyadic values that force rounding ties,
zero runs, NaNs, tiny sums) of random length on both sides of the one-workgroup / multi-workgroup switch; `sum`, `mean`
and `bottom_stddev` must equal the oracle's serial float chains bit for bit, the weights to 3e-6.
usage: PYTHONPATH=. python tests/soak_chains.py [cases=300] (GPU box)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import c_oracle as oracle  # noqa: E402
from top_down_renderer_amd.kernels import HipKernels  # noqa: E402

oracle.build()
k = HipKernels()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(4321)
f32 = np.float32
bad = 0
for case in range(cases):
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 5000), rng.integers(5000, 32769),
                        rng.integers(32769, 120000)]))
    kind = rng.integers(0, 5)
    if kind == 0:
        raw = np.exp(rng.normal(0, rng.uniform(0.1, 5), n))
    elif kind == 1:
        raw = rng.integers(0, 1 << int(rng.integers(1, 12)), n) * 2.0 ** -int(rng.integers(2, 20))
    elif kind == 2:
        raw = rng.random(n) ** int(rng.integers(1, 6))
        raw[rng.random(n) < rng.uniform(0, 0.9)] = 0
    elif kind == 3:
        raw = rng.random(n) * 10.0 ** -int(rng.integers(0, 36))
    else:
        raw = np.where(rng.random(n) < 0.5, 2.0 ** rng.integers(-30, 6, n), rng.random(n) * 8)
    raw = raw.astype(f32)
    raw[rng.random(n) < rng.choice([0.0, 0.02, 0.5, 0.98])] = np.nan
    ld = rng.random(n).astype(f32)
    w, info = k.zeros((n,)), k.zeros((65536,))
    k.update_weights(k.to_device(raw), k.to_device(ld), n, w, info)
    ref, best, stats = oracle.update_weights(raw, ld)
    got = info[1:4].cpu().numpy()
    ok = np.array_equal(got, np.asarray(stats[:3], f32), equal_nan=True) and \
        np.allclose(w.cpu().numpy(), ref, rtol=3e-6, atol=0, equal_nan=True)
    if not ok:
        bad += 1
        print(f"case {case}: n={n} kind={kind} stats {got} vs {stats[:3]}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
