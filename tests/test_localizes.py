"""Functional check that does not lean on the oracle being right: a scan rendered from a known pose must score highest
at that pose.  A grid of candidate poses around the truth (offsets in x, y and heading) is scored once; the best
weight has to sit on the true pose (or a direct grid neighbour — the map is piecewise constant).  Run for the CPU
oracle here and, on the GPU box, for the HIP path (same answer required)."""
import numpy as np
import pytest

from top_down_renderer_amd import synth


def _pose_grid(sc, step_px=6.0, step_deg=6.0, half=3):
    cx, cy, th = sc.pose
    offs = np.arange(-half, half + 1)
    gx, gy, gt = np.meshgrid(offs, offs, offs, indexing="ij")
    st = np.zeros(gx.size, synth.STATE_DTYPE)
    st["init_x_px"] = cx + step_px * gx.ravel()
    st["init_y_px"] = cy + step_px * gy.ravel()
    st["theta"] = th + np.deg2rad(step_deg) * gt.ravel()
    st["scale"] = 1.0
    st["have_init"] = 1
    return st, np.stack([gx.ravel(), gy.ravel(), gt.ravel()], 1)


def _check_peak(w, grid):
    best = grid[int(np.nanargmax(w))]
    assert np.abs(best).max() <= 1, f"likelihood peaks at grid offset {best}, not at the true pose"
    centre = w[np.all(grid == 0, 1)][0]
    far = w[np.abs(grid).max(1) == np.abs(grid).max()]
    assert centre > np.nanmax(far), "the true pose does not beat the far corners of the pose grid"


@pytest.mark.parametrize("name", ["c1"])
def test_oracle_likelihood_peaks_at_the_true_pose(oracle, name):
    sc = synth.make_scene(name, with_particles=False)
    cfg = sc.cfg
    st, grid = _pose_grid(sc)
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, 1.0)
    w = oracle.compute_weights(om, tab, cfg.nb, cfg.nr, scan, cfg.res, oracle.make_params(cfg.ncls), st)
    _check_peak(w, grid)


@pytest.mark.gpu
@pytest.mark.parametrize("name,search", [("c1", False), ("c1", True), ("ref", False)])
def test_gpu_likelihood_peaks_at_the_true_pose(name, search):
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    sc = synth.make_scene(name, with_particles=False)
    cfg = sc.cfg
    st, grid = _pose_grid(sc)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
    if search:
        # heading unknown: the 40-rotation search must land on one of the two candidates (9 degrees apart) that bracket the
        # true heading, give or take half a theta bin of the image
        st = st[np.all(grid[:, :2] == 0, 1)][:1].copy()
        st["have_init"] = 0
        st["theta"] = 0
    f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False)
    f.set_states(st)
    f.update(r.last_scan(), None, cfg.res)
    if search:
        got = k.states_to_host(f.st_new, 1, st.dtype)
        err = np.angle(np.exp(1j * (float(got["theta"][0]) - sc.pose[2])))
        assert abs(err) <= np.deg2rad(9.0) + 0.5 * float(cfg.ang_res), f"init search chose a heading {np.rad2deg(err):.1f} deg off"
    else:
        _check_peak(f.raw_weights(), grid)


@pytest.mark.gpu
def test_gpu_filter_converges_on_the_true_pose():
    """Closed loop through the class surface: propagate -> update (score, statistics, running sum, resample, gather)
    repeated on a static scene.  The particle cloud (sigma 30 px / 10 deg + 10 % uniform over the 1000 px map) must
    collapse around the pose the scan was rendered from.  The likelihood is flat along the road the pose sits on and
    the weights are regularised (1 / (cost + 0.15)), so where exactly the cloud condenses is up to resampling noise:
    the bounds are those of the basin, not of the peak (test_gpu_likelihood_peaks_at_the_true_pose pins the peak)."""
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    sc = synth.make_scene("c1", n_particles=4000)
    cfg = sc.cfg
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
    f = pkg.ParticleFilter(len(sc.states), m, pkg.FilterParams(fixed_scale=1.0), seed=5, kernels=k, init_particles=False)
    f.set_states(sc.states)
    cov0 = f.computeMeanCov()
    # a standing robot is not updated at all (weights are blended with 1/N by min(5 * distance travelled, 1),
    # particle_filter.cpp:137-141): move 0.2 m per step, which saturates the blend and drifts 8 px in total.  Weights are
    # 1 / (cost + 0.15), at most ~2:1 between good and bad particles here, so the cloud condenses over tens of steps
    for _ in range(40):
        f.propagate((0.2, 0.0), 0.0)
        f.update(r.last_scan(), None, cfg.res)
        assert f.weights().max() > 1.05 / len(sc.states)
    mean, cov = f.meanLikelihood(), f.computeMeanCov()
    cx, cy, th = sc.pose
    assert np.hypot(mean[0] - cx, mean[1] - cy) < 30, (mean, sc.pose)
    assert abs(np.angle(np.exp(1j * (float(mean[2]) - th)))) < np.deg2rad(4)
    assert cov[0, 0] + cov[1, 1] < 0.1 * (cov0[0, 0] + cov0[1, 1])
    ml = f.maxLikelihood()
    assert np.hypot(ml[0] - cx, ml[1] - cy) < 40
