"""Per-chunk timing of the walking workgroup of the multi-workgroup running sum.  Needs a diagnostic build:
   python -c "from top_down_renderer_amd import build as b; b.OUT='/root/repo/top_down_renderer_amd/libtdr_hip_dbg.so'; b.build(force=True, extra_flags=['-DTDR_PFX_TIMING'])"
   TDR_LIB_PATH=.../libtdr_hip_dbg.so PYTHONPATH=. python tools/diag_prefix.py      (GPU box)"""
import numpy as np, torch
from top_down_renderer_amd.kernels import HipKernels
k = HipKernels()
rng = np.random.default_rng(0)
for n in [100_000, 800_000]:
    raw = np.exp(rng.normal(0, 1.5, n)).astype(np.float32)
    raw[rng.random(n) < 0.05] = np.nan
    ld = np.abs(rng.normal(0, 1, n)).astype(np.float32)
    w, runmax, info = k.zeros((n,)), k.zeros((n,)), k.zeros((65536,))
    k.update_weights(k.to_device(raw), k.to_device(ld), n, w, info)
    for rep in range(3):
        k.prefix(w, n, runmax)
        k.synchronize()
    ws = k._pws.cpu().numpy()
    dt = np.dtype([("t", "<i8"), ("re", "<i4"), ("d0", "<u4"), ("d1", "<u4"), ("r0", "<f4"), ("c0", "<f4"), ("acc", "<i4")])
    nch = (n + 4095) // 4096
    h = ws[: nch * 32].view(dt)
    t = h["t"].astype(np.int64)
    d = np.diff(t) * 10   # ns
    acc = h["acc"][1:]
    print(n, "total walk ns (from chunk0 end)", (t[-1] - t[0]) * 10)
    print(" fast chunks: mean ns", d[acc == 1].mean(), "max", d[acc == 1].max(), " slow chunks ns:", d[acc == 0])
