"""Times the three running-sum implementations (tdr_k_prefix_mode 0/1/2) over particle counts (GPU box)."""
import ctypes as C
import time

import numpy as np

from top_down_renderer_amd.kernels import HipKernels

k = HipKernels()
rng = np.random.default_rng(0)
for n in [300, 1000, 2000, 4000, 8000, 12000, 16000, 20000, 32000, 50000, 100000]:
    w = rng.random(n).astype(np.float32) ** 3
    w = (w / w.sum()).astype(np.float32)
    wd, rm = k.to_device(w), k.zeros((n,))
    ws = k.prefix_workspace(n)
    out = []
    for mode in (0, 1, 2):
        def run():
            assert k.lib.tdr_k_prefix_mode(C.c_void_p(wd.data_ptr()), n, mode, C.c_void_p(rm.data_ptr()), None,
                                           C.c_void_p(ws.data_ptr()), k.stream()) == 0
        run(); k.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            run()
        k.synchronize()
        out.append((time.perf_counter() - t) / 20 * 1e6)
    print(f"n={n:7d}  serial {out[0]:8.1f} us   one-workgroup {out[1]:8.1f} us   multi {out[2]:8.1f} us", flush=True)
