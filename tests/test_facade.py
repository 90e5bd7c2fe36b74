"""The C++ drop-in surface (include/top_down_render/*.h over the handle layer of include/tdr.h).

CPU part: the reference-named classes compile with plain g++ against nothing but tdr.h (no Eigen / PCL / ROS in this
image) and link against libtdr_hip.so; constructing them without a GPU fails loudly.
GPU part: tests/cpp/facade_step.cpp replays TopDownRender::initialize + takeStep (src/top_down_render.cpp) through
those classes and its outputs are checked against the CPU oracle.
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "top_down_renderer_amd")


@pytest.fixture(scope="module")
def facade_exe():
    from top_down_renderer_amd import build
    build.build()
    exe = os.path.join(tempfile.mkdtemp(prefix="tdr_facade_"), "facade_step")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "facade_step.cpp"), "-o", exe, "-L", PKG, "-ltdr_hip",
           f"-Wl,-rpath,{PKG}"]
    subprocess.run(cmd, check=True)
    return exe


def test_facade_compiles_and_fails_loudly_without_gpu(facade_exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = tempfile.mkdtemp()
    open(os.path.join(d, "meta.txt"), "w").write("3 8 8 16 8 0 4 1.0 0.39 1 1 0 0 0\n")
    np.zeros(3 * 64, np.float32).tofile(os.path.join(d, "maps.bin"))
    np.zeros(64, np.uint8).tofile(os.path.join(d, "mask.bin"))
    np.zeros(0, np.float32).tofile(os.path.join(d, "pts.bin"))
    np.zeros(4 * 28, np.uint8).tofile(os.path.join(d, "states.bin"))
    r = subprocess.run([facade_exe, d], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("device_scan,labels_mode", [(0, 0), (1, 0), (1, 1)])
def test_facade_take_step_matches_oracle(facade_exe, oracle, device_scan, labels_mode):
    from top_down_renderer_amd import synth
    sc = synth.make_scene("c1", n_particles=2048)
    cfg = sc.cfg
    d = tempfile.mkdtemp(prefix="tdr_facade_run_")
    seed, tx, ty, omega = 17, 1.0, 0.25, 0.01
    open(os.path.join(d, "meta.txt"), "w").write(
        f"{cfg.ncls} {cfg.map_size} {cfg.map_size} {cfg.nb} {cfg.nr} {len(sc.pts)} {len(sc.states)} {cfg.res} "
        f"{float(cfg.ang_res)!r} {seed} {tx} {ty} {omega} {device_scan} {labels_mode}\n")
    # the same map as a class-index image (cv::Mat layout: row 0 = top), raw id 200 = unlabelled
    np.where(sc.lab >= 0, sc.lab, 200).astype(np.uint8)[::-1].copy().tofile(os.path.join(d, "labels.bin"))
    np.ascontiguousarray(np.transpose(sc.class_maps, (0, 2, 1)), np.float32).tofile(os.path.join(d, "maps.bin"))
    np.ascontiguousarray(sc.class_mask.T, np.uint8).tofile(os.path.join(d, "mask.bin"))
    pcl = np.zeros((len(sc.pts), 8), np.float32)
    pcl[:, :3], pcl[:, 3], pcl[:, 4] = sc.pts[:, :3], 1.0, sc.pts[:, 3]
    pcl.tofile(os.path.join(d, "pts.bin"))
    sc.states.tofile(os.path.join(d, "states.bin"))
    r = subprocess.run([facade_exe, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    n = len(sc.states)
    scan = np.fromfile(os.path.join(d, "out_scan.bin"), np.float32).reshape(cfg.ncls, -1)
    w = np.fromfile(os.path.join(d, "out_weights.bin"), np.float32)
    idx = np.fromfile(os.path.join(d, "out_idx.bin"), np.int32)
    st = np.fromfile(os.path.join(d, "out_states.bin"), oracle.STATE_DTYPE)
    stats = np.fromfile(os.path.join(d, "out_stats.bin"), np.float32)
    # the oracle's step, same mt19937 seed
    fpo = oracle.make_params(cfg.ncls)
    st_o = sc.states.copy()
    rng = oracle.Rng(seed)
    last = oracle.propagate(st_o, tx, ty, omega, True, fpo, rng)
    scan_o = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0)
    tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, 1.0)
    raw_o = oracle.compute_weights(om, tab, cfg.nb, cfg.nr, scan_o, cfg.res, fpo, st_o)
    w_o, best_o, _ = oracle.update_weights(raw_o, last)
    idx_o = oracle.resample_prefix(w_o, n, rng.uniform())
    new_o = oracle.gather_states(st_o, idx_o)
    assert np.array_equal(scan, scan_o)   # raster: exact
    assert np.allclose(w, w_o, rtol=2e-5, atol=0)
    assert (idx != idx_o).sum() <= 2 + n // 200
    same = idx == idx_o
    for name in ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta", "scale"):
        assert np.allclose(st[name][same], new_o[name][same], rtol=2e-6, atol=2e-6), name
    mean_o, cov_o = oracle.mean_cov(new_o)
    assert np.allclose(stats[:4], mean_o, rtol=1e-4, atol=1e-3)
    assert np.allclose(stats[4:20].reshape(4, 4), cov_o, rtol=2e-3, atol=1e-2)
    s = st_o[best_o]
    ml_o = np.asarray([s["dx_m"] * s["scale"] + s["init_x_px"], s["dy_m"] * s["scale"] + s["init_y_px"], s["theta"],
                       s["scale"]], np.float32)
    assert np.allclose(stats[20:24], ml_o, rtol=1e-5, atol=1e-4)
    assert np.allclose(stats[24:40].reshape(4, 4), oracle.cov_about(new_o, ml_o), rtol=2e-3, atol=1e-2)
    assert stats[40] == 1.0 and stats[41] == n and stats[42] == 1.0
    # getGMM after computeGMM on the resampled set (src/particle_filter.cpp:238-318; deterministic fit, parity unpinned)
    from oracle import np_oracle as no
    gmm = np.fromfile(os.path.join(d, "out_gmm.bin"), np.float32).reshape(-1, 12)
    idx_s = np.minimum(n - 1, np.arange(min(1000, n)) * n // min(1000, n))
    sx = (st["dx_m"] * st["scale"] + st["init_x_px"]).astype(np.float32)[idx_s]
    sy = (st["dy_m"] * st["scale"] + st["init_y_px"]).astype(np.float32)[idx_s]
    sth = st["theta"][idx_s]
    x = np.column_stack([sx, sy, np.float32(50) * np.cos(sth), np.float32(50) * np.sin(sth)]).astype(np.float64)
    kk, means_o, covs_o = no.gmm_select(x, n, 1)
    assert len(gmm) == kk
    assert np.allclose(gmm[:, :3], means_o, atol=2e-3) and np.allclose(gmm[:, 3:].reshape(-1, 3, 3), covs_o, rtol=2e-3, atol=2e-3)
