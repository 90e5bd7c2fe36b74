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


@pytest.fixture(scope="module")
def session_exe():
    from top_down_renderer_amd import build
    build.build()
    exe = os.path.join(tempfile.mkdtemp(prefix="tdr_facade_"), "facade_session")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_session.cpp"), "-o", exe, "-L", PKG, "-ltdr_hip",
                    f"-Wl,-rpath,{PKG}"], check=True)
    return exe


def test_reference_call_sites_compile_unchanged():
    """tests/cpp/call_sites.cpp writes out every call TopDownRender makes into the hot-path classes (src/top_down_render.cpp
    :81,115-117,173-177,333-359,423-431,508,529-540,591) with the node's argument types — among them the three the
    round-1 surface lacked: `params.color_lut = ...` (:173), `filter_->visualize(img)` (:431) and the two-argument
    `filter_->updateMap(img, centre)` (:591).  -Wall -Werror, linked against libtdr_hip.so."""
    from top_down_renderer_amd import build
    build.build()
    exe = os.path.join(tempfile.mkdtemp(prefix="tdr_facade_"), "call_sites")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "call_sites.cpp"), "-o", exe, "-L", PKG, "-ltdr_hip",
                    f"-Wl,-rpath,{PKG}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "call sites compile" in r.stdout


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
    assert np.allclose(w, w_o, rtol=1e-5, atol=0)
    assert (idx != idx_o).sum() <= 2 + n // 200
    same = idx == idx_o
    for name in ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta", "scale"):
        assert np.array_equal(st[name][same], new_o[name][same]), name     # propagate is bit for bit
    # pose statistics of the particle set the GPU path ended with (summation order is all that differs)
    mean_o, cov_o = oracle.mean_cov(st)
    assert np.allclose(stats[:4], mean_o, rtol=2e-5, atol=2e-5)
    assert np.allclose(stats[4:20].reshape(4, 4), cov_o, rtol=1e-4, atol=1e-4)
    s = st_o[best_o]
    ml_o = np.asarray([s["dx_m"] * s["scale"] + s["init_x_px"], s["dy_m"] * s["scale"] + s["init_y_px"], s["theta"],
                       s["scale"]], np.float32)
    assert np.allclose(stats[20:24], ml_o, rtol=1e-5, atol=1e-4)
    assert np.allclose(stats[24:40].reshape(4, 4), oracle.cov_about(st, ml_o), rtol=1e-4, atol=1e-4)
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


@pytest.mark.gpu
def test_facade_session_matches_oracle(session_exe, oracle):
    """tests/cpp/facade_session.cpp: constructor-time initializeParticles with an unknown scale, getClassesAtPoint, two
    steps, freezeScale, one more step, computeGMM + adaptive particle count, updateMap with a moved centre — through the
    C++ classes, each stage against the oracle (later stages on the states the GPU path itself produced)."""
    from oracle import np_oracle as no
    from top_down_renderer_amd import synth
    sc = synth.make_scene("c1", with_particles=False)
    cfg = sc.cfg
    n, seed = 1500, 23
    d = tempfile.mkdtemp(prefix="tdr_session_")
    img = np.where(sc.lab >= 0, sc.lab, 200).astype(np.uint8)[::-1].copy()      # cv::Mat layout, 200 = unlabelled
    img.tofile(os.path.join(d, "labels.bin"))
    kw = dict(fixed_scale=-1.0, init_pos_px_x=float(sc.pose[0]), init_pos_px_y=float(sc.pose[1]), init_pos_px_cov=12.0,
              init_pos_deg_theta=float(np.rad2deg(sc.pose[2])), init_pos_deg_cov=4.0)
    open(os.path.join(d, "meta.txt"), "w").write(
        f"{cfg.ncls} {cfg.map_size} {cfg.map_size} {cfg.nb} {cfg.nr} {len(sc.pts)} {n} {cfg.res} {float(cfg.ang_res)!r} "
        f"{seed} {kw['init_pos_px_x']!r} {kw['init_pos_px_y']!r} {kw['init_pos_px_cov']} {kw['init_pos_deg_theta']!r} "
        f"{kw['init_pos_deg_cov']}\n")
    pcl = np.zeros((len(sc.pts), 8), np.float32)
    pcl[:, :3], pcl[:, 3], pcl[:, 4] = sc.pts[:, :3], 1.0, sc.pts[:, 3]
    pcl.tofile(os.path.join(d, "pts.bin"))
    r = subprocess.run([session_exe, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rd = lambda name, dt: np.fromfile(os.path.join(d, name), dt)  # noqa: E731
    # the map the label image turns into (bit-exact ingest, tests/test_ingest.py) and the oracle's filter on it
    lut = -np.ones(256, np.int32)
    lut[: cfg.ncls] = np.arange(cfg.ncls)
    maps, mask = no.load_compressed_raster_map(img, lut, cfg.ncls, 1.0)
    om = oracle.OracleMap(maps, mask, 1.0)
    fpo = oracle.make_params(cfg.ncls, **kw)
    rng = oracle.Rng(seed)
    st_o = oracle.initialize_particles(om, fpo, n, rng)
    st0 = rd("out_init_states.bin", oracle.STATE_DTYPE)
    assert st0.tobytes() == st_o.tobytes()                      # same std::mt19937 stream, same rejections
    assert len(np.unique(st0["scale"])) >= 5                    # unknown scale: a ladder of scales (:40-58)
    cls = rd("out_classes.bin", np.int32).reshape(-1, cfg.ncls)
    for i in range(len(cls)):
        bits = oracle.classes_at_point(om, int(st0["init_x_px"][i]), int(st0["init_y_px"][i]))
        assert [c for c in cls[i] if c >= 0] == [c for c in range(cfg.ncls) if bits >> c & 1]
    # getLocalMap of the first particle (src/top_down_map_polar.cpp:21-53): the oracle's window, value for value
    tab0 = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, 1.0)
    s0 = st0[0]
    d_o, k_o = oracle.local_map_polar(om, tab0, np.float32(s0["dx_m"] * s0["scale"] + s0["init_x_px"]),
                                      np.float32(s0["dy_m"] * s0["scale"] + s0["init_y_px"]), s0["scale"], cfg.res)
    win = rd("out_window.bin", np.float32).reshape(cfg.ncls + 1, -1)
    assert np.array_equal(win[: cfg.ncls], d_o) and np.array_equal(win[cfg.ncls], k_o.astype(np.float32))
    # ActiveLocalizer::getBestRelPos (src/active_localizer.cpp:44-82) through the C++ class: the oracle's choice (or one
    # that ties with it to rounding) and the same mean difference
    act = rd("out_active.bin", np.float32)
    o_best, o_diff, o_all = oracle.active_best_rel_pos(om, tab0, cfg.nb, cfg.nr, act[3:12].reshape(3, 3))
    assert np.isclose(act[2], o_diff, rtol=1e-5)
    if not np.array_equal(act[:2], o_best):
        cand = np.nanmax(o_all)
        assert np.isclose(cand, o_diff, rtol=1e-5)   # a tie within rounding
    assert np.isclose(act[12], act[2], rtol=1e-5)
    # Cartesian render + Cartesian getLocalMap through the C++ base classes
    wr, wc = 40, 56
    cart = rd("out_cartesian.bin", np.float32).reshape(2 * cfg.ncls + 1, wr * wc)
    assert np.array_equal(cart[: cfg.ncls], oracle.raster_cart(sc.pts, cfg.res, sc.lut, cfg.ncls, wr, wc))
    dc_o, kc_o = oracle.local_map_cart(om, float(st0["init_x_px"][0]), float(st0["init_y_px"][0]), 0.6, 1.5, wr, wc)
    nbad = int((cart[cfg.ncls: 2 * cfg.ncls] != dc_o).any(0).sum()) + int((cart[2 * cfg.ncls] != kc_o).sum())
    assert nbad == 0                         # cos / sin of the rotation are the host libm's bit for bit (tests/test_libm.py)
    # StateParticle: two particles on one shared generator (constructor draw, propagate with / without scale freeze,
    # computeWeight, weight, lastDist, mlState, setScale)
    sp = rd("out_state_particles.bin", np.float32)
    g2 = oracle.Rng(seed + 1)
    pa, pb = oracle.init_particle(om, fpo, g2), oracle.init_particle(om, fpo, g2)
    fields = ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta", "scale")
    for j, pp in enumerate((pa, pb)):
        assert np.array_equal(sp[7 * j: 7 * j + 6], np.asarray([pp[f][0] for f in fields], np.float32))
        assert sp[7 * j + 6] == float(pp["have_init"][0])
    la = oracle.propagate(pa, 1.0, 0.25, 0.01, False, fpo, g2)
    lb = oracle.propagate(pb, 1.0, 0.25, 0.01, True, fpo, g2)
    scan_sp = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    for j, (pp, ll) in enumerate(((pa, la), (pb, lb))):
        o = 14 + 13 * j
        assert np.array_equal(sp[o: o + 6], np.asarray([pp[f][0] for f in fields], np.float32))   # bit for bit
        w_ref = oracle.compute_weights(om, tab0, cfg.nb, cfg.nr, scan_sp, cfg.res, fpo, pp.copy())[0]
        assert (np.isnan(w_ref) and np.isnan(sp[o + 7])) or np.isclose(sp[o + 7], w_ref, rtol=1e-5)
        assert sp[o + 8] == ll[0]
        s1 = pp[0]
        assert np.allclose(sp[o + 9: o + 13], [s1["dx_m"] * s1["scale"] + s1["init_x_px"], s1["dy_m"] * s1["scale"] + s1["init_y_px"],
                                              s1["theta"], s1["scale"]], rtol=2e-6, atol=2e-5)
    assert sp[14 + 26] == 2.5
    assert sp[14 + 13 + 5] == sp[7 + 5]          # scale_freeze = true: b's scale did not move (:70-73)
    # step 1, oracle in lock step (same generator continues)
    last = oracle.propagate(st_o, 1.0, 0.25, 0.01, False, fpo, rng)
    scan_o = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, 1.0)
    raw_o = oracle.compute_weights(om, tab, cfg.nb, cfg.nr, scan_o, cfg.res, fpo, st_o)
    w_o, _, _ = oracle.update_weights(raw_o, last)
    idx_o = oracle.resample_prefix(w_o, n, rng.uniform())
    new_o = oracle.gather_states(st_o, idx_o)
    w1, idx1, st1 = rd("out_weights_step1.bin", np.float32), rd("out_idx_step1.bin", np.int32), rd("out_states_step1.bin", oracle.STATE_DTYPE)
    assert np.isnan(raw_o).any()                                # the scale gate fired for some particles
    assert np.allclose(w1, w_o, rtol=1e-5, atol=0)
    assert (idx1 != idx_o).sum() <= 2 + n // 200
    same = idx1 == idx_o
    for name in ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta", "scale"):
        assert np.array_equal(st1[name][same], new_o[name][same]), name
    # freezeScale (src/particle_filter.cpp:343-357) on the states the GPU path had after step 2
    st2, st3 = rd("out_states_step2.bin", oracle.STATE_DTYPE), rd("out_states_frozen.bin", oracle.STATE_DTYPE)
    misc = rd("out_misc.bin", np.float32)
    assert misc[0] == -1.0 and misc[1] == 0.0                   # scale() before freezing: unknown (:359-367)
    ref2 = st2.copy()
    gm = oracle.freeze_scale(ref2)
    assert np.allclose(st3["scale"], gm, rtol=2e-6) and len(np.unique(st3["scale"])) == 1
    for name in ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta"):
        assert np.array_equal(st3[name], st2[name])
    assert np.isclose(misc[2], gm, rtol=2e-6) and misc[3] == 1.0
    # after freezing, propagate leaves the scale alone (state_particle.cpp:70-73)
    st4 = rd("out_states_step3.bin", oracle.STATE_DTYPE)
    assert np.all(st4["scale"] == st3["scale"][0])
    # adaptive particle count (:151-157) from computeGMM on the step-3 states
    idx_s = np.minimum(n - 1, np.arange(min(1000, n)) * n // min(1000, n))
    sx = (st4["dx_m"] * st4["scale"] + st4["init_x_px"]).astype(np.float32)[idx_s]
    sy = (st4["dy_m"] * st4["scale"] + st4["init_y_px"]).astype(np.float32)[idx_s]
    x = np.column_stack([sx, sy, np.float32(50) * np.cos(st4["theta"][idx_s]), np.float32(50) * np.sin(st4["theta"][idx_s])]).astype(np.float64)
    _, _, covs = no.gmm_select(x, n, 1)
    assert int(misc[4]) == oracle.adaptive_count(covs[:, :2, :2], n, n)
    # updateMap with a moved centre (:325-334)
    assert misc[5] == 7.0 and misc[6] == -3.0 and misc[7] == 6997.0
    # the two-argument updateMap(cv::Mat, centre) of the reference, update()'s shape check, visualize() not throwing,
    # renderGeometricTopDown against the oracle, getLocalGeoMap after the dynamic-map path (constant 1 inside the map)
    extra = rd("out_extra.bin", np.float32)
    assert extra[0] == 2.0 and extra[1] == 2.0 and extra[2] == 1.0 and extra[3] == 1.0
    # the sharded constructor on a one-rank RCCL communicator == the plain filter, bit for bit; the opt-in geometric
    # cost with zero geometric images == the plain cost
    assert extra[4] == 1.0 and extra[6] == 768.0
    assert extra[5] <= 1e-6
    assert extra[7] == 0.5     # visualize() handed every particle and a best state to the host's hook
    geo = np.stack([rd("out_geo_render.bin", np.float32), rd("out_geo_render1.bin", np.float32)])
    pts4 = np.ascontiguousarray(sc.pts[:, :4])
    assert np.array_equal(geo, oracle.raster_geo_polar(pts4, len(pts4), 1, cfg.res, cfg.ang_res, cfg.nb, cfg.nr))


def test_default_drawing_of_visualize_compiles_and_draws_with_an_opencv_stand_in():
    """ParticleFilter::visualize without a hook draws like the reference (src/particle_filter.cpp:373-423) when OpenCV is
    present (include/top_down_render/particle_viz.h).  This image has no OpenCV: tests/cpp/viz_compile.cpp builds the drawing
    against a minimal stand-in that records the primitives (tests/cpp/opencv_stub — test infrastructure only) and checks
    them on a hand-made snapshot; no GPU involved."""
    from top_down_renderer_amd import build
    build.build()
    exe = os.path.join(tempfile.mkdtemp(prefix="tdr_facade_"), "viz_compile")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "opencv_stub"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "viz_compile.cpp"), "-o", exe,
                    "-L", PKG, "-ltdr_hip", f"-Wl,-rpath,{PKG}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "as expected" in r.stdout, r.stdout + r.stderr
