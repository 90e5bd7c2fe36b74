"""ActiveLocalizer (src/active_localizer.cpp) — one launch for all candidate displacements against the CPU oracle's
restatement of the reference's loops.  Run with `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tdr():
    import torch
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return pkg, HipKernels()


def _setup(pkg, k, oracle, seed, ncls=6, size=500, shape=(100, 25)):
    from top_down_renderer_amd import synth
    cfg = synth.Config("act", 2000, ncls, shape[0], shape[1], size, 64, seed=seed)
    sc = synth.make_scene(cfg)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    ang = np.float32(2 * np.pi / shape[0])
    m.samplePtsPolar(shape, ang)                                      # src/top_down_render.cpp:115
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0)
    tab = oracle.polar_table(shape[0], shape[1], ang, 1.0)
    return sc, m, om, tab


@pytest.mark.parametrize("K,seed", [(2, 1), (3, 2), (4, 3), (7, 4)])
def test_best_rel_pos_matches_the_oracle(tdr, oracle, K, seed):
    pkg, k = tdr
    sc, m, om, tab = _setup(pkg, k, oracle, 6000 + seed)
    rng = np.random.default_rng(seed)
    preds = np.column_stack([rng.uniform(60, 440, K), rng.uniform(60, 440, K), rng.uniform(-4, 4, K)]).astype(np.float32)
    preds[0, :2] = (5.0, 495.0)                                       # a hypothesis in a corner: windows leave the map
    al = pkg.ActiveLocalizer(m)
    got = al.getBestRelPos(preds)
    ref, ref_diff, ref_all = oracle.active_best_rel_pos(om, tab, 100, 25, preds)
    # every candidate the reference's loops visit, to 1e-5 (sums of |a - b| over 2500 x pairs x classes floats)
    seen = ~np.isnan(ref_all)
    assert np.array_equal(seen, ~np.isnan(al.last_diffs))
    assert np.allclose(al.last_diffs[seen], ref_all[seen], rtol=1e-5, atol=0)
    assert np.isclose(al.last_best_diff, ref_diff, rtol=1e-5)
    if not np.array_equal(got, ref):                                  # another candidate may win only by a rounding tie
        assert np.isclose(np.nanmax(al.last_diffs), ref_diff, rtol=1e-5)


def test_one_hypothesis_and_none(tdr, oracle):
    """One hypothesis: the reference divides 0 by 0 and no candidate ever beats 0 (:15-19, 70) — {0, 0}, all four
    distances visited; no hypothesis: the same."""
    pkg, k = tdr
    sc, m, om, tab = _setup(pkg, k, oracle, 6100)
    al = pkg.ActiveLocalizer(m)
    preds = np.array([[200.0, 210.0, 0.3]], np.float32)
    assert np.array_equal(al.getBestRelPos(preds), np.zeros(2, np.float32)) and al.last_best_diff == 0.0
    ref, ref_diff, _ = oracle.active_best_rel_pos(om, tab, 100, 25, preds)
    assert np.array_equal(ref, np.zeros(2, np.float32)) and ref_diff == 0.0
    assert np.array_equal(al.getBestRelPos(np.zeros((0, 3), np.float32)), np.zeros(2, np.float32))


def test_the_search_stops_at_the_first_distance_that_reaches_6000(tdr, oracle):
    """Two hypotheses far apart on a map whose classes differ strongly: the mean difference passes 6000 at the first
    distance and the loops stop there (:58)."""
    pkg, k = tdr
    sc, m, om, tab = _setup(pkg, k, oracle, 6200, ncls=3, size=700)
    preds = np.array([[80.0, 90.0, 0.0], [600.0, 610.0, 2.0], [90.0, 600.0, -1.0]], np.float32)
    al = pkg.ActiveLocalizer(m)
    got = al.getBestRelPos(preds)
    ref, ref_diff, ref_all = oracle.active_best_rel_pos(om, tab, 100, 25, preds)
    assert np.array_equal(~np.isnan(ref_all), ~np.isnan(al.last_diffs))
    assert np.isclose(al.last_best_diff, ref_diff, rtol=1e-5)
    if ref_diff >= 6000:
        assert np.isnan(ref_all[1:]).all() or np.isnan(ref_all[-1]).all()
    if not np.array_equal(got, ref):
        assert np.isclose(np.nanmax(al.last_diffs), ref_diff, rtol=1e-5)


def test_handle_call_equals_the_python_host(tdr, oracle):
    """tdr_map_best_rel_pos (what the C++ class calls) is exercised by tests/test_facade.py on the GPU; here the candidate
    generator it shares with the Python host: 16 directions per distance, the float loop's angles."""
    import ctypes as C
    pkg, k = tdr
    preds = np.array([[10.0, 20.0, 0.5], [30.0, 40.0, -2.0]], np.float32)
    centres = np.zeros((68, 2, 2), np.float32)
    dists, thetas, shifts = np.zeros(68, np.float32), np.zeros(68, np.float32), np.zeros(2, np.int32)
    nt, nd = C.c_int(0), C.c_int(0)
    assert k.lib.tdr_active_candidates_host(preds.ctypes.data, 2, 100, centres.ctypes.data, dists.ctypes.data,
                                            thetas.ctypes.data, shifts.ctypes.data, C.byref(nt), C.byref(nd)) == 0
    th, t = [], np.float32(0)
    while t < 2 * np.pi:
        th.append(t)
        t = np.float32(t + np.pi / 8)
    assert nt.value == len(th) and nd.value == 4
    assert np.array_equal(thetas[: nt.value], np.array(th, np.float32)) and np.array_equal(dists[::17], [50, 75, 100, 125])
    assert list(shifts) == [oracle.rot_shift(0.5, 100), oracle.rot_shift(-2.0, 100)]
    ang = np.float32(thetas[3] + preds[1, 2])
    assert np.isclose(centres[17 + 3, 1, 0], preds[1, 0] + 75.0 * np.cos(float(ang)), rtol=1e-6)
