#!/usr/bin/env python3
"""Times one polar scoring call of a config (default c2) in its integer form under several particle distributions and
splits between the two kernels (GPU box only):
    span 0       every particle through the shift-uniform kernel
    span 8 / 16  the mixed launch (dense particles shift-uniform, scattered ones ray-mapped); "+ctx": with a caller context
                 that holds the table's factors (the ray-mapped kernel multiplies the offsets out, block-major rows)
    all ray      every particle through the ray-mapped kernel (with and without the factors)
    float        the float kernel (tdr_config_shift_uniform(0))
Usage: python3 tools/time_int_form.py [config] [quick] [only=<distribution>] [ray|su]
   ray / su: only the all-ray / all-shift-uniform launch, three times (counter passes: tools/pmc_ray_bound.sh)
   rayctx [patch=0|1]: the all-ray launch with a caller context (block-major / patch order)"""
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
    name = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in synth.CONFIGS else "c2"
    quick = "quick" in sys.argv
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
    n = cfg.n_particles
    sets = {"mix(90g+10u)": sc.states,
            "uniform": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=1.0),
            "gauss5": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, sigma_px=5.0, sigma_deg=2.0, uniform_frac=0.0)}
    if not quick:
        sets["gauss30"] = synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0)
    only = [x[5:] for x in sys.argv if x.startswith("only=")]
    if only:
        sets = {k_: v for k_, v in sets.items() if only[0] in k_}
    fp = pkg.FilterParams(fixed_scale=1.0).to_c(cfg.ncls)
    st = k.zeros((7, n))
    raw = k.zeros((n,))
    perm = k.zeros((n,), torch.int32)
    ctx = k.score_ctx_create()
    lib = k.lib

    def timed(reps, **kw):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k.score(m.dev, scan, cfg.res, fp, st, n, raw, perm=perm, uniform_scale=1.0, **kw)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts[1:]) if len(ts) > 1 else ts[0]

    for sname, states in sets.items():
        k.states_to_device(states, st, n)
        k.locality_order(st, n, m.rows, m.cols, perm)
        row = []
        lib.tdr_config_shift_uniform(1)
        lib.tdr_config_ray_split(0)
        if "mixed" in sys.argv:   # the mixed launch only (kernel traces)
            lib.tdr_config_shift_uniform_span(16.0)
            if "splits" in sys.argv:
                for span in (8.0, 16.0, 24.0):
                    lib.tdr_config_shift_uniform_span(span)
                    for split in (1, 2, 4, 8):
                        lib.tdr_config_ray_split(split)
                        print(f"{sname:14s} span {span:g} split {split}: no ctx {timed(5):6.2f}   +ctx {timed(5, ctx=ctx):6.2f}", flush=True)
                lib.tdr_config_ray_split(0)
                continue
            print(f"{sname:14s} span 16 +ctx: {timed(6, ctx=ctx):6.2f}   one stream: {timed(6):6.2f}", flush=True)
            lib.tdr_config_shift_uniform_span(-2.0)
            continue
        if "rayctx" in sys.argv:   # every particle through the ray-mapped kernel, the caller's context holding the table's factors
            for a_ in sys.argv:
                if a_.startswith("patch="):
                    lib.tdr_config_tuning(b"ray_patch", int(a_[6:]))
            lib.tdr_config_shift_uniform_span(1e-6)
            print(f"{sname:14s} all ray +ctx (ray_patch {lib.tdr_config_tuning(b'ray_patch', -1)}): {timed(4, ctx=ctx):6.2f}", flush=True)
            lib.tdr_config_shift_uniform_span(-2.0)
            continue
        if "ray" in sys.argv or "su" in sys.argv:
            lib.tdr_config_shift_uniform_span(1e-6 if "ray" in sys.argv else 0.0)
            print(f"{sname:14s} {'all ray' if 'ray' in sys.argv else 'span 0'}: {timed(3):6.2f}", flush=True)
            lib.tdr_config_shift_uniform_span(-2.0)
            continue
        for label, span, c in (("span 0", 0.0, None), ("span 8 +ctx", 8.0, ctx), ("span 16 +ctx", 16.0, ctx),
                               ("span 16 no ctx", 16.0, None), ("span 40 +ctx", 40.0, ctx), ("all ray +ctx", 1e-6, ctx),
                               ("all ray no ctx", 1e-6, None)):
            lib.tdr_config_shift_uniform_span(span)
            row.append((label, timed(4, ctx=c)))
        if not quick:
            for split in (1, 2, 4):
                lib.tdr_config_ray_split(split)
                lib.tdr_config_shift_uniform_span(1e-6)
                row.append((f"all ray +ctx, split {split}", timed(3, ctx=ctx)))
            lib.tdr_config_ray_split(0)
        ref = raw[:n].clone()
        lib.tdr_config_shift_uniform(0)
        row.append(("float kernel", timed(3)))
        diff = (raw[:n] - ref).abs() / ref.abs().clamp_min(1e-30)
        diff = diff[~torch.isnan(diff)]
        lib.tdr_config_shift_uniform(1)
        lib.tdr_config_shift_uniform_span(-2.0)
        print(f"{sname:14s} " + "   ".join(f"{a}: {b:6.2f}" for a, b in row) +
              f"   | int vs float max rel {float(diff.max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
