"""Generates the committed golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference ships no fixtures for this path and cannot be built or imported here, so these vectors
come from the C++ restatement (oracle/oracle.cpp) and are cross-checked, before being written, against the
independent NumPy statement (oracle/np_oracle.py).  PARITY UNPINNED by the reference itself.

Fixtures are plain .npz files (numeric arrays only, loadable with allow_pickle=False).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import c_oracle as co  # noqa: E402
from oracle import np_oracle as no  # noqa: E402
from top_down_renderer_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def states_to_dict(st):
    return {"init_x": st["init_x_px"].copy(), "init_y": st["init_y_px"].copy(), "dx": st["dx_m"].copy(),
            "dy": st["dy_m"].copy(), "theta": st["theta"].copy(), "scale": st["scale"].copy(),
            "have_init": st["have_init"].copy()}


def edge_states(sc):
    """The micro particle set plus hand-placed edge cases."""
    st = sc.states.copy()
    H, W = sc.lab.shape
    st["dx_m"] = np.linspace(-1, 1, len(st)).astype(np.float32)      # exercise centre = d*scale + init
    st["dy_m"] = np.linspace(0.5, -0.5, len(st)).astype(np.float32)
    st["init_x_px"][0], st["init_y_px"][0], st["theta"][0] = sc.pose  # true pose
    st["dx_m"][:9] = 0
    st["dy_m"][:9] = 0
    st["init_x_px"][1], st["init_y_px"][1] = -500.0, -500.0            # fully out of bounds -> NaN
    st["init_x_px"][2], st["init_y_px"][2] = 0.4, 0.4                  # corner: >half the window unknown
    st["init_x_px"][3], st["init_y_px"][3] = W - 0.5, H / 2            # edge
    st["init_x_px"][4], st["init_y_px"][4] = W + 0.4, H / 2            # just off the map (force_on_map gate)
    st["theta"][5] = 7.5                                               # shift normalisation > 2pi
    st["theta"][6] = -9.25                                             # negative rotation
    st["scale"][7] = 0.5                                               # scale != 1
    st["scale"][8] = 12.0                                              # outside [10^-0.1, 10^1] when scale unknown
    st["have_init"][9:14] = 0                                          # 40-rotation init search
    st["init_x_px"][13], st["init_y_px"][13] = -500.0, 20.0            # un-initialised AND all-NaN
    return st


def main():
    sc = synth.make_scene("micro")
    cfg = sc.cfg
    m = co.OracleMap(sc.class_maps, sc.class_mask, cfg.map_resolution)
    out = {"pts": sc.pts, "class_maps": sc.class_maps, "class_mask": sc.class_mask, "lut": sc.lut,
           "pose": np.asarray(sc.pose, np.float64),
           "shape": np.asarray([cfg.ncls, cfg.nb, cfg.nr, cfg.map_size], np.int32),
           "res": np.float32(cfg.res), "ang_res": np.float32(cfg.ang_res)}

    # A1 / A2 raster
    scan = co.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    assert np.array_equal(scan, no.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr))
    cart = co.raster_cart(sc.pts, 0.5, sc.lut, cfg.ncls, 12, 10)
    assert np.array_equal(cart, no.raster_cart(sc.pts, 0.5, sc.lut, cfg.ncls, 12, 10))
    out["scan"], out["scan_cart_12x10_res0p5"] = scan, cart

    # A4 table
    tab = co.polar_table(cfg.nb, cfg.nr, cfg.ang_res, cfg.map_resolution)
    assert np.array_equal(tab.T, no.polar_table(cfg.nb, cfg.nr, cfg.ang_res, cfg.map_resolution))
    out["table"] = tab

    # A5 gather at the true pose
    d, k = co.local_map_polar(m, tab, sc.pose[0], sc.pose[1], 1.0, cfg.res)
    d2, k2 = no.local_map_polar(sc.class_maps, sc.class_mask, 1.0, tab.T, sc.pose[0], sc.pose[1], 1.0, cfg.res)
    assert np.array_equal(d, d2) and np.array_equal(k, k2)
    out["window_dists"], out["window_mask"] = d, k

    # A8/A9 weights for three parameter sets
    st0 = edge_states(sc)
    out["states_in"] = st0.view(np.uint8).reshape(len(st0), 28)
    variants = {
        "default": dict(),
        "force_on_map": dict(force_on_map=True),
        "scale_unknown": dict(fixed_scale=-1.0, class_weights=[1.0, 0.5, 2.0], regularization=0.7),
    }
    for name, kw in variants.items():
        fp = co.make_params(cfg.ncls, **kw)
        st = st0.copy()
        w = co.compute_weights(m, tab, cfg.nb, cfg.nr, scan, cfg.res, fp, st, nthreads=2)
        sd = states_to_dict(st0)
        fpd = dict(regularization=fp.regularization, force_on_map=bool(fp.force_on_map), fixed_scale=fp.fixed_scale,
                   scale_log_min=fp.scale_log_min, scale_log_max=fp.scale_log_max,
                   class_weights=[fp.class_weights[i] for i in range(cfg.ncls)])
        wn = no.compute_weights(sc.class_maps, sc.class_mask, 1.0, tab.T, cfg.nb, cfg.nr, scan, cfg.res, fpd, sd)
        assert np.array_equal(np.isnan(w), np.isnan(wn)), name
        ok = ~np.isnan(w)
        assert np.allclose(w[ok], wn[ok], rtol=2e-6, atol=0), (name, w, wn)
        assert np.array_equal(st["theta"], sd["theta"]) and np.array_equal(st["have_init"], sd["have_init"])
        out[f"weights_{name}"] = w
        out[f"theta_after_{name}"] = st["theta"].copy()

    # A10/A11 propagate with the shared mt19937 (seed 7), and the normals it consumed
    fp = co.make_params(cfg.ncls)
    for freeze in (0, 1):
        st = st0.copy()
        last = co.propagate(st, 1.0, 0.25, 0.01, bool(freeze), fp, co.Rng(7))
        out[f"prop_states_freeze{freeze}"] = st.view(np.uint8).reshape(len(st), 28)
        out[f"prop_last_dist_freeze{freeze}"] = last
        out[f"prop_normals_freeze{freeze}"] = co.propagate_normals(len(st), bool(freeze), co.Rng(7))
    out["rng_seed7_uniform3"] = np.asarray([(lambda r: [r.uniform() for _ in range(3)])(co.Rng(7))], np.float32)[0]

    # A12 weight statistics / A14 resample
    raw = out["weights_default"].copy()
    last = out["prop_last_dist_freeze1"]
    w, best, stats = co.update_weights(raw, last)
    wn, bestn = no.update_weights(raw, last)
    assert best == bestn and np.allclose(w, wn, rtol=1e-6, atol=0)
    out["upd_weights"], out["upd_argmax"], out["upd_stats"] = w, np.int64(best), stats
    allnan = np.full(8, np.nan, np.float32)
    w2, best2, stats2 = co.update_weights(allnan, np.full(8, 0.1, np.float32))
    out["upd_weights_allnan"] = w2
    for n_new in (len(w), 20, 50):
        idx = co.resample_literal(w, n_new, 0.37)
        assert np.array_equal(idx, co.resample_prefix(w, n_new, 0.37))
        assert np.array_equal(idx, no.resample(w, n_new, 0.37))
        out[f"resample_idx_{n_new}"] = idx
    # negative-weight case (NaN fill below zero): prefix is non-monotone
    wneg = np.asarray([0.3, -0.2, 0.25, 0.05, -0.1, 0.4, 0.3], np.float32)
    idx = co.resample_literal(wneg, 9, 0.5)
    assert np.array_equal(idx, co.resample_prefix(wneg, 9, 0.5)) and np.array_equal(idx, no.resample(wneg, 9, 0.5))
    out["resample_neg_w"], out["resample_neg_idx"] = wneg, idx

    # A16 statistics
    mean, cov = co.mean_cov(st0)
    out["mean_state"], out["mean_cov"] = mean, cov
    stf = st0.copy()
    stf["scale"] = np.linspace(0.8, 1.3, len(stf)).astype(np.float32)
    out["freeze_scale_in"] = stf["scale"].copy()
    out["freeze_scale_geo_mean"] = np.float32(co.freeze_scale(stf))

    np.savez_compressed(os.path.join(OUT, "micro.npz"), **out)
    print("wrote", os.path.join(OUT, "micro.npz"), {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
