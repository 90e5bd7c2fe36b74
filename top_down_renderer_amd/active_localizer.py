"""ActiveLocalizer — Python mirror of the reference class (include/top_down_render/active_localizer.h:6-17,
src/active_localizer.cpp).  Every candidate displacement of getBestRelPos is one workgroup of ONE launch
(csrc/tdr_active.hip); the host keeps the reference's candidate loops and its sequential choice."""
import ctypes as C

import numpy as np
import torch

from ._lib import check


class ActiveLocalizer:
    def __init__(self, map_):                                    # src/active_localizer.cpp:3-5
        self.map_ = map_
        self.k = map_.k
        self.last_best_diff = 0.0
        self.last_diffs = None    # [distances][17] of the last getBestRelPos (NaN where the reference's loops did not go)

    def getBestRelPos(self, preds):
        """:44-82.  preds: (K, 3) = {x, y, theta} of the pose hypotheses; returns (distance, direction)."""
        preds = np.ascontiguousarray(preds, np.float32).reshape(-1, 3)
        K = len(preds)
        self.last_best_diff = 0.0
        if K == 0:
            return np.zeros(2, np.float32)
        m, k, lib = self.map_.dev, self.k, self.k.lib
        ncand = 4 * 17
        centres = np.zeros((ncand, K, 2), np.float32)
        dists = np.zeros(ncand, np.float32)
        thetas = np.zeros(ncand, np.float32)
        shifts = np.zeros(K, np.int32)
        nt, nd = C.c_int(0), C.c_int(0)
        check(lib.tdr_active_candidates_host(preds.ctypes.data, K, m.nb, centres.ctypes.data, dists.ctypes.data,
                                             thetas.ctypes.data, shifts.ctypes.data, C.byref(nt), C.byref(nd)))
        d_c, d_s = k.to_device(centres), k.to_device(shifts)
        sums = k.zeros((ncand,), torch.float64)
        check(lib.tdr_k_active_diffs(C.byref(m.desc), C.c_void_p(m.tab.data_ptr()), m.nb, m.nr, C.c_float(2.0),
                                     C.c_void_p(d_c.data_ptr()), C.c_void_p(d_s.data_ptr()), K, ncand,
                                     C.c_void_p(sums.data_ptr()), k.stream()))
        sums = sums.cpu().numpy()
        cnt = np.float32(K * (K - 1) // 2 * m.ncls)                                  # :15
        with np.errstate(all="ignore"):
            diffs = (sums.astype(np.float32) / cnt).reshape(4, 17)                   # :19
        diffs[:, nt.value:] = np.nan
        best, out = np.float32(0), np.zeros(2, np.float32)
        seen = np.full((4, 17), np.nan, np.float32)
        for di in range(nd.value):
            if not best < 6000:                                                       # :58
                break
            for t in range(nt.value):
                seen[di, t] = diffs[di, t]
                if diffs[di, t] > best:                                               # :70-73 (NaN never wins)
                    best = diffs[di, t]
                    out[:] = (dists[di * 17 + t], thetas[di * 17 + t])
        self.last_best_diff, self.last_diffs = float(best), seen
        return out
