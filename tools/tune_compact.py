#!/usr/bin/env python3
"""A/B of the record form tdr_k_score_polar reads, on one device, config-2 scene: ms per call for several particle
distributions with the dense-record kernel and the compact-record kernel.
    python3 tools/tune_compact.py [config] [distribution-name filter] [axis]
axis "compact" (default): dense records vs compact records; axis "su": lane-shift kernel vs shift-uniform kernel (both on
the compact records); axis "su-only" / "lane-only": one kernel (counter passes)."""
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


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    only = sys.argv[2] if len(sys.argv) > 2 else ""      # substring filter on the distribution name (profiling runs)
    axis = sys.argv[3] if len(sys.argv) > 3 else "compact"

    k = HipKernels()
    cfg = synth.CONFIGS[name]
    n = cfg.n_particles // 8 if name in ("c3", "c5") else cfg.n_particles
    t0 = time.time()
    sc = synth.make_scene(cfg, n_particles=n)
    print(f"scene {time.time() - t0:.1f} s", flush=True)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    print("compact words", m.dev.desc.cwords, "dict", m.dev.desc.dict_n, flush=True)
    r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
    scan = r.last_scan()[1]
    rng = np.random.default_rng(1)
    dists = {
        "bench mix (90% Gaussian 30 px + 10% uniform)": sc.states,
        "100% Gaussian 30 px": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0),
        "100% Gaussian 5 px": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0, sigma_px=5.0),
        "100% uniform": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=1.0),
        "8 clusters 40 px": synth.make_cluster_particles(cfg, sc.lab, rng, per_cluster=n // 8),
        "Gaussian 30 px, one heading": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0, sigma_deg=0.0),
        "Gaussian 30 px, headings uniform": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0,
                                                                 sigma_deg=1e4),
    }
    for pct in (1, 2, 3, 5):
        dists[f"Gaussian 30 px + {pct}% uniform"] = synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n,
                                                                         uniform_frac=pct / 100.0)
    modes = {"compact": [("dense", 0), ("compact", 1)], "su": [("lane-shift", 0), ("shift-uniform", 1)],
             "su-only": [("shift-uniform", 1)], "lane-only": [("lane-shift", 0)]}[axis]
    setter = k.lib.tdr_config_compact if axis == "compact" else k.lib.tdr_config_shift_uniform
    fp = pkg.FilterParams(fixed_scale=1.0).to_c(cfg.ncls)
    ref = {}
    for dname, states in dists.items():
        if only and only not in dname:
            continue
        states = states.copy()
        states["have_init"] = 1
        st = k.zeros((7, n))
        k.states_to_device(states, st, n)
        perm = k.zeros((n,), torch.int32)
        k.locality_order(st, n, m.rows, m.cols, perm)
        raw = k.zeros((n,))
        line = []
        for mname, on in modes:
            setter(on)
            for _ in range(2):
                k.score(m.dev, scan, cfg.res, fp, st, n, raw, perm=perm, uniform_scale=1.0)
            k.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                k.score(m.dev, scan, cfg.res, fp, st, n, raw, perm=perm, uniform_scale=1.0)
            e1.record()
            k.synchronize()
            ms = e0.elapsed_time(e1) / reps
            got = raw.cpu().numpy()
            if dname not in ref:
                ref[dname] = got
            same = np.array_equal(ref[dname], got, equal_nan=True)
            line.append(f"{mname}: {ms:7.2f} ms{'' if same else ' MISMATCH'}")
        print(f"{dname:48s} " + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
