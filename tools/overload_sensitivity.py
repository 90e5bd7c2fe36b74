#!/usr/bin/env python3
"""What the oracle's UNPINNED overload choices can cost (CPU only; test infrastructure — uses oracle/).

The reference calls unqualified atan2 / sqrt on floats in the polar raster (src/scan_renderer_polar.cpp:97-98) and unqualified
cos / sin / atan2 in meanLikelihood (src/particle_filter.cpp:198-202).  Whether those are the float overloads or the C
library's double functions depends on the include graph of the reference's translation units (Eigen / PCL / ROS / OpenCV
headers, absent here).  The oracle — and the HIP path — take the float overloads; `orc_set_overload_mode` switches the
oracle to the other reading.  This script reports, on the synthetic c1 / c2 scans, how many raster bins move, how far the
raw weights of a particle sample move, and how far meanLikelihood / computeMeanCov move.  It pins nothing: it bounds what
"unpinned" can cost.

    PYTHONPATH=. python tools/overload_sensitivity.py [c1 c2 ref] > profiles/r05_overload_sensitivity.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle as o  # noqa: E402
from top_down_renderer_amd import synth  # noqa: E402


def one(name, n_particles):
    sc = synth.make_scene(name, n_particles=n_particles)
    cfg = sc.cfg
    print(f"== {name}: {len(sc.pts)} points, {cfg.ncls} classes, {cfg.nb} x {cfg.nr} polar image, res {cfg.res}, "
          f"{len(sc.states)} particles")
    o.set_overload_mode(0)
    scan_f = o.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    o.set_overload_mode(1)
    scan_d = o.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    o.set_overload_mode(0)
    moved_bins = int((scan_f != scan_d).sum())
    moved_pts = int(np.abs(scan_f - scan_d).sum() // 2)
    print(f"raster, float atan2f/sqrtf vs (float)atan2((double)..)/sqrt: {moved_bins} of {scan_f.size} bins differ "
          f"({moved_pts} of {int(scan_f.sum())} counted points land in another bin)")
    # per-point view: which of the two functions moves them
    x, y = sc.pts[:, 0].astype(np.float32), sc.pts[:, 1].astype(np.float32)
    th_f = np.arctan2(x, y, dtype=np.float32)
    th_d = np.arctan2(x.astype(np.float64), y.astype(np.float64)).astype(np.float32)
    print(f"  numpy's view of the same question: atan2 results differ on {int((th_f != th_d).sum())} of {len(x)} points "
          f"(a bin moves only when the difference straddles a rounding boundary of theta / ang_res)")
    om = o.OracleMap(sc.class_maps, sc.class_mask, cfg.map_resolution)
    tab = o.polar_table(cfg.nb, cfg.nr, cfg.ang_res, cfg.map_resolution)
    fp = o.make_params(cfg.ncls)
    st = sc.states.copy()
    w_f = o.compute_weights(om, tab, cfg.nb, cfg.nr, scan_f, cfg.res, fp, st.copy())
    w_d = o.compute_weights(om, tab, cfg.nb, cfg.nr, scan_d, cfg.res, fp, st.copy())
    ok = ~np.isnan(w_f) & ~np.isnan(w_d)
    rel = np.abs(w_f[ok] - w_d[ok]) / np.abs(w_f[ok])
    print(f"raw weights of {int(ok.sum())} particles scored with either raster: max relative difference {rel.max():.3e}, "
          f"median {np.median(rel):.3e}, above 1e-5: {int((rel > 1e-5).sum())}; NaN pattern equal: "
          f"{bool(np.array_equal(np.isnan(w_f), np.isnan(w_d)))}")
    last = np.zeros(len(st), np.float32)
    wn_f, best_f, _ = o.update_weights(w_f, last)
    wn_d, best_d, _ = o.update_weights(w_d, last)
    idx_f = o.resample_prefix(wn_f, len(st), 0.37)
    idx_d = o.resample_prefix(wn_d, len(st), 0.37)
    print(f"  normalised weights max rel diff {np.max(np.abs(wn_f - wn_d) / np.abs(wn_f)):.3e}; arg-max {best_f} vs {best_d}; "
          f"resample indices that differ: {int((idx_f != idx_d).sum())} of {len(idx_f)}")
    # meanLikelihood / computeMeanCov
    mean_f, cov_f = o.mean_cov(st)
    o.set_overload_mode(2)
    mean_d, cov_d = o.mean_cov(st)
    o.set_overload_mode(0)
    print(f"meanLikelihood, cosf/sinf/atan2f vs double cos/sin/atan2 accumulated into float: mean theta {mean_f[2]!r} vs "
          f"{mean_d[2]!r} (diff {abs(float(mean_f[2]) - float(mean_d[2])):.3e} rad); cov(2,2) {cov_f[2, 2]!r} vs {cov_d[2, 2]!r}; "
          f"max |cov diff| {np.abs(cov_f - cov_d).max():.3e}")


def main():
    names = sys.argv[1:] or ["c1", "c2", "ref"]
    sizes = {"c1": 1000, "c2": 4000, "ref": 4000}
    for n in names:
        one(n, sizes.get(n, 2000))
        print()


if __name__ == "__main__":
    main()
