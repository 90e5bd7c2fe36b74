#!/usr/bin/env python3
"""40-rotation search for maps of 8-15 classes: matrix-core kernel (score_init_mfma_wide_kernel) against the vector-unit
kernel, same device, same particles.   python3 tools/time_init_wide.py [ncls] [particles]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import top_down_renderer_amd as pkg  # noqa: E402
from top_down_renderer_amd import synth  # noqa: E402
from top_down_renderer_amd.kernels import HipKernels  # noqa: E402


def main():
    ncls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
    k = HipKernels()
    cfg = synth.Config("initwide", 100_000, ncls, 256, 256, 2000, n, seed=5)
    sc = synth.make_scene(cfg)
    st = synth.make_cluster_particles(cfg, sc.lab, np.random.default_rng(2), n_clusters=8, per_cluster=n // 8)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    r.set_output_shape(ncls, cfg.nb, cfg.nr)
    r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
    scan = r.last_scan()[1]
    f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False)
    thetas = {}
    for on in (0, 1):
        k.lib.tdr_config_init_mfma(on)
        ms = []
        for rep in range(3):
            f.set_states(st)
            k.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            k.score(m.dev, scan, cfg.res, f.fp_c, f.st, len(st), f.raw_w, init_search=True, uniform_scale=f._uniform_scale)
            e1.record()
            k.synchronize()
            ms.append(e0.elapsed_time(e1))
        thetas[on] = k.states_to_host(f.st, len(st), pkg.STATE_DTYPE)["theta"]
        print(f"{ncls} classes, {len(st)} particles without a heading, 256 x 256 window: search + scoring pass "
              f"{'matrix cores' if on else 'vector units '}: {min(ms):.1f} ms", flush=True)
    print(f"rotations that differ between the two: {(thetas[0] != thetas[1]).mean():.2%}")
    k.lib.tdr_config_init_mfma(1)


if __name__ == "__main__":
    main()
