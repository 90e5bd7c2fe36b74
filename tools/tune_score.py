#!/usr/bin/env python3
"""Times the scoring kernel of config c2 under different particle distributions / processing orders (GPU box only)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import top_down_renderer_amd as pkg  # noqa: E402
from top_down_renderer_amd import synth  # noqa: E402
from top_down_renderer_amd.kernels import HipKernels  # noqa: E402


USC = float(os.environ.get("TDR_USCALE", "1.0"))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    k = HipKernels()
    cfg = synth.CONFIGS[name]
    sc = synth.make_scene(cfg)
    rng = np.random.default_rng(99)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
    scan = r.last_scan()[1]
    img = r.last_images().cpu().numpy()
    print("scan: nonzero bins %.3f, max count %d" % ((img.sum(0) > 0).mean(), img.max()))
    n = cfg.n_particles
    sets = {
        "mix(90g+10u)": sc.states,
        "gauss30": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0),
        "gauss5": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, sigma_px=5.0, sigma_deg=2.0, uniform_frac=0.0),
        "uniform": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=1.0),
    }
    fp = pkg.FilterParams(fixed_scale=1.0).to_c(cfg.ncls)
    st = k.zeros((7, n))
    raw = k.zeros((n,))
    perm = k.zeros((n,), torch.int32)
    for sname, states in sets.items():
        k.states_to_device(states, st, n)
        cx = (states["dx_m"] * states["scale"] + states["init_x_px"]).astype(np.float64)
        cy = (states["dy_m"] * states["scale"] + states["init_y_px"]).astype(np.float64)
        col = np.clip(np.floor(cx), 0, m.cols - 1).astype(np.int64)
        row = np.clip(np.floor(cy), 0, m.rows - 1).astype(np.int64)

        def morton(r, c):
            key = np.zeros_like(r)
            for b in range(12):
                key |= ((c >> b) & 1) << (2 * b)
                key |= ((r >> b) & 1) << (2 * b + 1)
            return key
        host_perms = {
            "rowmajor1px": np.lexsort((cx, col, row)),
            "morton1px": np.argsort(morton(row, col), kind="stable"),
            "morton.5px": np.argsort(morton(np.clip(np.floor(cy * 2), 0, 8191).astype(np.int64), np.clip(np.floor(cx * 2), 0, 8191).astype(np.int64)), kind="stable"),
        }
        th = np.mod(states["theta"].astype(np.float64), 2 * np.pi)

        def morton3(a, b, c):
            key = np.zeros_like(a)
            for bit in range(12):
                key |= ((a >> bit) & 1) << (3 * bit)
                key |= ((b >> bit) & 1) << (3 * bit + 1)
                key |= ((c >> bit) & 1) << (3 * bit + 2)
            return key
        for q in (1.0, 2.0, 4.0):
            for rref in (64.0, 128.0):
                tb = np.floor(th / (q / rref)).astype(np.int64)
                host_perms[f"m3 q={q:g} r={rref:g}"] = np.argsort(
                    morton3(np.floor(cx / q).astype(np.int64).clip(0), np.floor(cy / q).astype(np.int64).clip(0), tb), kind="stable")
        for deg in (1.0, 3.0):
            tb = np.floor(th / np.deg2rad(deg)).astype(np.int64)
            host_perms[f"theta{deg:g}deg+morton.5"] = np.lexsort((morton(np.floor(cy * 2).astype(np.int64).clip(0, 8191), np.floor(cx * 2).astype(np.int64).clip(0, 8191)), tb))
        if "m3" in sys.argv:
            locs = (True,) + tuple(x for x in host_perms if x.startswith(("m3", "theta")))
        elif "sorted" in sys.argv:
            locs = (True,)
        elif "quick" in sys.argv:
            locs = (False, True)
        else:
            locs = (False, True, "rowmajor1px", "morton1px", "morton.5px")
        for loc in locs:
            if loc is True:
                k.locality_order(st, n, m.rows, m.cols, perm)
            elif isinstance(loc, str):
                perm.copy_(torch.from_numpy(host_perms[loc].astype(np.int32)))
            ts = []
            for rep in range(5):
                k.lib.tdr_profile_enable(1)
                k.score(m.dev, scan, cfg.res, fp, st, n, raw, perm=perm if loc else None, uniform_scale=USC)
                tot, cnt = C.c_double(0), C.c_int64(0)
                k.lib.tdr_profile_score_ms(C.byref(tot), C.byref(cnt))
                ts.append(tot.value)
            k.lib.tdr_profile_enable(0)
            print(f"{sname:14s} locality={str(loc):12s}  score kernel ms: " + " ".join(f"{t:7.2f}" for t in ts) + f"   min {min(ts):7.2f}", flush=True)


if __name__ == "__main__":
    main()
