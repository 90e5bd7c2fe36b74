#!/usr/bin/env python3
"""Times the Cartesian scoring kernel (config c4 shapes) under different particle processing orders (GPU box only).
usage: PYTHONPATH=. python tools/tune_cart.py [n_particles=50000]"""
import ctypes as C
import sys

import numpy as np
import torch

import top_down_renderer_amd as pkg
from top_down_renderer_amd import synth
from top_down_renderer_amd.kernels import HipKernels


def morton(parts, bits=12):
    key = np.zeros_like(parts[0])
    d = len(parts)
    for b in range(bits):
        for a, v in enumerate(parts):
            key |= ((v >> b) & 1) << (d * b + a)
    return key


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    k = HipKernels()
    cfg = synth.CONFIGS["c4"]
    sc = synth.make_scene(cfg, n_particles=n)
    rows, cols = cfg.nb, cfg.nr
    m = pkg.TopDownMap(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.setWindow(rows, cols)
    r = pkg.ScanRenderer(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, rows, cols)
    r.renderSemanticTopDown(sc.pts, cfg.res)
    scan = r.last_scan()[1]
    st_h = sc.states
    fp = pkg.FilterParams(fixed_scale=1.0).to_c(cfg.ncls)
    st, raw, perm = k.zeros((7, n)), k.zeros((n,)), k.zeros((n,), torch.int32)
    k.states_to_device(st_h, st, n)
    cx = (st_h["dx_m"] * st_h["scale"] + st_h["init_x_px"]).astype(np.float64)
    cy = (st_h["dy_m"] * st_h["scale"] + st_h["init_y_px"]).astype(np.float64)
    th = np.mod(st_h["theta"].astype(np.float64), 2 * np.pi)
    ix = lambda v, q: np.floor(v / q).astype(np.int64).clip(0)  # noqa: E731
    orders = {"none": None, "device (x,y) half px": "dev"}
    for q in (0.5, 2.0, 4.0, 8.0):
        for rref in (64.0, 128.0, 256.0):
            orders[f"(x,y,theta) q={q:g}px r={rref:g}"] = np.argsort(
                morton([ix(cx, q), ix(cy, q), np.floor(th / (q / rref)).astype(np.int64)]), kind="stable")
    for name, o in orders.items():
        if isinstance(o, str):
            k.locality_order(st, n, m.rows, m.cols, perm)
        elif o is not None:
            perm.copy_(torch.from_numpy(o.astype(np.int32)))
        ts = []
        for rep in range(3):
            k.lib.tdr_profile_enable(1)
            k.score_cart(m.dev, scan, rows, cols, cfg.res, fp, st, n, raw, perm=None if o is None else perm)
            tot, cnt = C.c_double(0), C.c_int64(0)
            k.lib.tdr_profile_score_ms(C.byref(tot), C.byref(cnt))
            ts.append(tot.value)
        k.lib.tdr_profile_enable(0)
        print(f"{name:34s} score_cart ms: " + " ".join(f"{t:8.2f}" for t in ts) + f"   min {min(ts):8.2f}", flush=True)


if __name__ == "__main__":
    main()
