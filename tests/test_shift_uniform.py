"""The integer form of a polar launch — the shift-uniform kernel (csrc/tdr_score_su.hip) for dense particles, the ray-mapped
kernel (csrc/tdr_score_ray.hip) for scattered ones — against itself (array equality: the integer sums are exact, whichever
kernel forms them), against the float kernel (score_polar_kernel: equal to rounding) and against the CPU oracle (1e-5,
BASELINE.json north_star).  Run with `pytest -m gpu`.

Small launches are scored as SHARDS of a large filter (n_total = 10^6): the ring groups then have the size a large
filter gets, which is what the kernel needs (groups of a multiple of 4 rings), and tdr_config_shift_uniform(2) takes the
path whatever the padding costs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_TOTAL = 1_000_000


@pytest.fixture(scope="module")
def tdr():
    import torch
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return pkg, HipKernels()


def _assert_weights(w, ref, rtol=1e-5):
    assert np.array_equal(np.isnan(w), np.isnan(ref)), (np.isnan(w).sum(), np.isnan(ref).sum())
    ok = ~np.isnan(ref)
    err = np.abs(w[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-30)
    assert err.max(initial=0.0) <= rtol, f"max rel err {err.max():.3e}"


ALL_RAY = 1e-6   # a span no two distinct particles fit in: every particle counts as scattered


def _assert_float_form_agrees(raw_float, raw_int, rtol=3e-6):
    """The float kernel adds a window's products up in float, group by group; the integer form is exact: same NaN / zero
    pattern, values equal to the float kernel's rounding."""
    assert np.array_equal(np.isnan(raw_float), np.isnan(raw_int))
    ok = ~np.isnan(raw_int)
    assert np.array_equal(raw_float[ok] == 0, raw_int[ok] == 0)
    err = np.abs(raw_float[ok] - raw_int[ok]) / np.maximum(np.abs(raw_int[ok]), 1e-30)
    assert err.max(initial=0.0) <= rtol, f"float form off by {err.max():.3e}"


def _score_both(pkg, k, m, scan, res, st, params, locality=1, init_search=False, span=0.0, ctx=None):
    """Raw weights (and the states after scoring) of the float kernel (mode 0) and of the integer form (mode 2).
    span 0: every particle through the shift-uniform kernel; > 0: particles farther than that from their neighbours in
    the locality order go through the ray-mapped kernel (the mixed launch); ALL_RAY: all of them."""
    before = k.lib.tdr_config_shift_uniform(-1)
    out = []
    try:
        k.lib.tdr_config_shift_uniform_span(span)
        for mode in (0, 2):
            k.lib.tdr_config_shift_uniform(mode)
            f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(**params), kernels=k, init_particles=False,
                                   locality_every=locality)
            f.set_states(st)
            n = len(st)
            perm = None
            if locality:
                perm = k.zeros((f.cap_local,), __import__("torch").int32)
                k.locality_order(f.st, n, m.rows, m.cols, perm)
            launches = int(k.lib.tdr_shift_uniform_launches())
            k.score(m.dev, m.scan_handle(scan), float(res), f.fp_c, f.st, n, f.raw_w, perm=perm, init_search=init_search,
                    uniform_scale=f._uniform_scale, n_total=N_TOTAL, ctx=ctx)
            k.synchronize()
            took = int(k.lib.tdr_shift_uniform_launches()) - launches
            assert took == (1 if mode == 2 else 0), f"mode {mode}: {took} shift-uniform launches"
            out.append((f.raw_w[:n].cpu().numpy(), k.states_to_host(f.st, n, pkg.STATE_DTYPE)))
    finally:
        k.lib.tdr_config_shift_uniform(before)
        k.lib.tdr_config_shift_uniform_span(-2.0)   # back to the default: tuned while running
    return out


CASES = [  # ncls, nb, nr, n, fixed scale, locality
    (1, 32, 16, 700, True, 1),
    (2, 64, 32, 1500, True, 0),
    (3, 36, 24, 1500, True, 1),
    (5, 48, 20, 2000, False, 1),
    (6, 64, 24, 3000, True, 1),
    (6, 50, 28, 1500, False, 0),
    (7, 32, 16, 1000, True, 1),
    (9, 40, 24, 1200, False, 1),
    (11, 24, 12, 600, True, 1),
    (6, 100, 25, 2500, True, 1),      # the reference node's own image (src/top_down_render.cpp:115): 25 rings, a ragged last group
    (6, 100, 50, 2500, False, 1),     # the map's default table (src/top_down_map_polar.cpp:4)
    (3, 40, 13, 900, True, 0),
]


@pytest.mark.parametrize("ncls,nb,nr,n,scale_fixed,locality", CASES)
def test_shift_uniform_equals_lane_shift_kernel_and_oracle(tdr, oracle, ncls, nb, nr, n, scale_fixed, locality):
    from top_down_renderer_amd import synth
    pkg, k = tdr
    cfg = synth.Config("su", 5000, ncls, nb, nr, 260, n, seed=4100 + 17 * ncls + nb)
    sc = synth.make_scene(cfg)
    st = sc.states.copy()
    rng = np.random.default_rng(ncls * 100 + nb)
    if not scale_fixed:
        st["scale"] = rng.uniform(0.6, 1.7, n).astype(np.float32)
    far = rng.random(n) < 0.1
    st["init_x_px"][far] = rng.uniform(-300, 600, int(far.sum())).astype(np.float32)   # partly outside the map
    wild = rng.random(n) < 0.05
    st["theta"][wild] = rng.uniform(-40, 40, int(wild.sum())).astype(np.float32)       # many turns, negative headings
    params = dict(fixed_scale=1.0 if scale_fixed else -1.0)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    assert m.dev.desc.cwords > 0
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    ref = oracle.compute_weights(oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0),
                                 oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res), cfg.nb, cfg.nr, scan, cfg.res,
                                 oracle.make_params(cfg.ncls, **params), st.copy())
    (raw0, _), (raw2, _) = _score_both(pkg, k, m, scan, cfg.res, st, params, locality)
    _assert_float_form_agrees(raw0, raw2)
    _assert_weights(raw2, ref)
    # mixed launch: the particles with neighbours within 3 cells through the shift-uniform kernel, the others through the
    # ray-mapped kernel (here on a stream of its own: a caller's context); then every particle through the ray-mapped kernel
    (_, _), (raw3, _) = _score_both(pkg, k, m, scan, cfg.res, st, params, locality, span=3.0, ctx=k.score_ctx_create())
    assert np.array_equal(raw2, raw3, equal_nan=True)
    (_, _), (raw4, _) = _score_both(pkg, k, m, scan, cfg.res, st, params, locality, span=ALL_RAY)
    assert np.array_equal(raw2, raw4, equal_nan=True)


@pytest.mark.parametrize("holes", ["none", "one far corner"])
def test_shift_uniform_on_a_fully_known_map(tdr, oracle, holes):
    """A map without unknown cells: every sector of every workgroup whose windows stay inside it takes the loop without
    clamp and without known-mask lookups (`allknown`); windows that reach the border, or the one unknown corner, do not."""
    from top_down_renderer_amd import synth
    pkg, k = tdr
    cfg = synth.Config("suk", 6000, 6, 64, 24, 420, 4096, seed=4700)
    sc = synth.make_scene(cfg)
    mask = np.zeros_like(sc.class_mask)
    maps = sc.class_maps.copy()
    if holes != "none":
        mask[:40, :40] = 1
        maps[:, :40, :40] = 0
    st = sc.states.copy()
    rng = np.random.default_rng(9)
    st["init_x_px"] = rng.normal(210, 12, len(st)).astype(np.float32)      # one tight cluster in the middle
    st["init_y_px"] = rng.normal(210, 12, len(st)).astype(np.float32)
    st["dx_m"] = 0
    st["dy_m"] = 0
    st["theta"] = rng.normal(0.3, 0.05, len(st)).astype(np.float32)
    st["init_x_px"][:64] = rng.uniform(0, 420, 64).astype(np.float32)       # and a few anywhere, borders included
    params = dict(fixed_scale=1.0)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), maps, mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    ref = oracle.compute_weights(oracle.OracleMap(maps, mask, 1.0), oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res),
                                 cfg.nb, cfg.nr, scan, cfg.res, oracle.make_params(cfg.ncls, **params), st.copy())
    (raw0, _), (raw2, _) = _score_both(pkg, k, m, scan, cfg.res, st, params)
    _assert_float_form_agrees(raw0, raw2)
    _assert_weights(raw2, ref)
    (_, _), (raw4, _) = _score_both(pkg, k, m, scan, cfg.res, st, params, span=ALL_RAY)
    assert np.array_equal(raw2, raw4, equal_nan=True)


@pytest.mark.parametrize("kind", ["empty", "dense", "fractional", "one bin"])
def test_shift_uniform_scan_contents(tdr, oracle, kind):
    """Descriptor classes: every bin empty; most bins holding several classes; counts that are not integers and negative
    ones (ParticleFilter::update is handed images, not counts: such a scan has no integer form, the device notices and the
    float kernel does the launch); a single occupied bin."""
    from top_down_renderer_amd import synth
    pkg, k = tdr
    cfg = synth.Config("suc", 3000, 6, 40, 16, 200, 900, seed=4300)
    sc = synth.make_scene(cfg)
    rng = np.random.default_rng(5)
    P = cfg.nb * cfg.nr
    if kind == "empty":
        scan = np.zeros((cfg.ncls, P), np.float32)
    elif kind == "dense":
        scan = rng.integers(0, 4, (cfg.ncls, P)).astype(np.float32)
    elif kind == "fractional":
        scan = np.where(rng.random((cfg.ncls, P)) < 0.2, rng.normal(0, 2, (cfg.ncls, P)), 0).astype(np.float32)
        scan[0, ::7] = -0.0
    else:
        scan = np.zeros((cfg.ncls, P), np.float32)
        scan[3, 5 + cfg.nb * 7] = 9
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    st = sc.states.copy()
    with np.errstate(all="ignore"):
        ref = oracle.compute_weights(oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0),
                                     oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res), cfg.nb, cfg.nr, scan, cfg.res,
                                     oracle.make_params(cfg.ncls), st.copy())
    (raw0, _), (raw2, _) = _score_both(pkg, k, m, scan, cfg.res, st, dict(fixed_scale=1.0))
    (_, _), (raw3, _) = _score_both(pkg, k, m, scan, cfg.res, st, dict(fixed_scale=1.0), span=4.0)
    (_, _), (raw4, _) = _score_both(pkg, k, m, scan, cfg.res, st, dict(fixed_scale=1.0), span=ALL_RAY)
    assert np.array_equal(raw2, raw3, equal_nan=True) and np.array_equal(raw2, raw4, equal_nan=True)
    if kind == "fractional":
        assert np.array_equal(raw0, raw2, equal_nan=True)   # the float kernel ran in both modes
    else:
        _assert_float_form_agrees(raw0, raw2)
    if kind != "fractional":   # sums of mixed signs cancel: the 1e-5 bound is for counts
        fin = np.isfinite(ref)
        _assert_weights(raw2[fin], ref[fin])
        assert np.array_equal(np.isfinite(raw2), fin)


def test_shift_uniform_keeps_zero_times_infinity(tdr, oracle):
    """A map value of +inf makes 0 * inf = NaN in the reference's products (state_particle.cpp:136-139): a dictionary holding
    such a value has no integer form, the float kernel does the launch in either mode and multiplies every product out."""
    from top_down_renderer_amd import synth
    pkg, k = tdr
    cfg = synth.Config("sui", 3000, 6, 32, 16, 160, 800, seed=4400)
    sc = synth.make_scene(cfg)
    maps = sc.class_maps.copy()
    maps[2, 40:90, 30:100] = np.inf
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), maps, sc.class_mask, kernels=k)
    assert m.dev.desc.cwords > 0
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    st = sc.states.copy()
    (raw0, _), (raw2, _) = _score_both(pkg, k, m, scan, cfg.res, st, dict(fixed_scale=1.0))
    assert np.isnan(raw0).any() and not np.isnan(raw0).all()
    assert np.array_equal(raw0, raw2, equal_nan=True)
    with np.errstate(all="ignore"):
        ref = oracle.compute_weights(oracle.OracleMap(maps, sc.class_mask, 1.0),
                                     oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res), cfg.nb, cfg.nr, scan, cfg.res,
                                     oracle.make_params(cfg.ncls), st.copy())
    assert np.array_equal(np.isnan(raw2), np.isnan(ref))


def test_shift_uniform_after_the_init_search(tdr, oracle):
    """Un-initialised particles get their heading from the 40-rotation search first; the order is built from the headings
    the search wrote."""
    from top_down_renderer_amd import synth
    pkg, k = tdr
    cfg = synth.Config("sus", 4000, 6, 40, 24, 220, 1200, seed=4500)
    sc = synth.make_scene(cfg)
    st = sc.states.copy()
    st["have_init"][::3] = 0
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    (raw0, st0), (raw2, st2) = _score_both(pkg, k, m, scan, cfg.res, st, dict(fixed_scale=1.0), init_search=True)
    assert np.array_equal(st0["theta"], st2["theta"]) and st2["have_init"].all()
    _assert_float_form_agrees(raw0, raw2)


def test_shift_uniform_c2_all_particles_vs_oracle(tdr, oracle):
    """BASELINE configs[1] at full size: one step (propagate, render, update) with the default kernel choice — the
    integer form at this size, dense and scattered particles side by side — against the float kernel (to rounding) and
    against the oracle over ALL 100 000 particles: raw weights to 1e-5, the NaN pattern, the normalised weights, the
    arg-max and the resample indices."""
    from top_down_renderer_amd import synth
    pkg, k = tdr
    sc = synth.make_scene("c2")
    cfg = sc.cfg
    n = len(sc.states)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
    res = []
    before = k.lib.tdr_config_shift_uniform(-1)
    try:
        for mode in (0, 1):
            k.lib.tdr_config_shift_uniform(mode)
            launches = int(k.lib.tdr_shift_uniform_launches())
            f = pkg.ParticleFilter(n, m, pkg.FilterParams(fixed_scale=1.0), seed=5, kernels=k, init_particles=False,
                                   locality_every=1)   # parity RNG: the reference's mt19937 stream
            f.set_states(sc.states)
            f.propagate((1.0, 0.0), 0.01)
            f.update(r.last_scan(), None, cfg.res, shift=0.37)
            assert int(k.lib.tdr_shift_uniform_launches()) - launches == mode
            res.append((f.raw_weights(), f.weights(), f.resample_indices(), f._argmax()))
    finally:
        k.lib.tdr_config_shift_uniform(before)
    raw, w, idx, _ = res[1]
    _assert_float_form_agrees(res[0][0], raw)
    assert np.allclose(res[0][1], w, rtol=1e-5, atol=0) and int((res[0][2] != idx).sum()) <= 2 + n // 200
    # the oracle's step on the same inputs
    fpo = oracle.make_params(cfg.ncls)
    st = sc.states.copy()
    last = oracle.propagate(st, 1.0, 0.0, 0.01, True, fpo, oracle.Rng(5))
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    ref_raw = oracle.compute_weights(oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0),
                                     oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, 1.0), cfg.nb, cfg.nr, scan, cfg.res,
                                     fpo, st)
    _assert_weights(raw, ref_raw)
    ref_w, ref_argmax, _ = oracle.update_weights(ref_raw, last)
    assert np.allclose(w, ref_w, rtol=2e-5, atol=0)
    assert res[1][3] == ref_argmax or abs(w[ref_argmax] - w.max()) <= 2e-5 * w.max()
    # resampling the ORACLE's weights on the device reproduces the oracle's indices bit for bit (same weight inputs);
    # the device's own weights differ in the last bits, which may move a handful of boundaries by one particle
    ref_idx = oracle.resample_prefix(ref_w, n, 0.37)
    import torch
    wd = k.to_device(ref_w)
    runmax = k.empty((n,))
    k.prefix(wd, n, runmax)
    out = k.zeros((n,), torch.int32)
    k.resample(runmax, n, n, 0.37, 0, n, out)
    assert np.array_equal(out.cpu().numpy(), ref_idx)
    mism = int((idx != ref_idx).sum())
    assert mism <= 2 + n // 200, f"{mism} resample indices differ"


def test_span_tuned_while_running_never_changes_the_weights(tdr, oracle):
    """A caller that brings a context has the span that routes particles between the two kernels of a mixed launch tuned
    while the filter runs (five candidates, two timed scoring calls each, repeated now and then): 52 calls on the same
    particles — through the skipped calls, every candidate and the settled state — give the same raw weights, bit for bit,
    and two contexts on one device (two filters of different shapes, called in turn) each settle for themselves."""
    from top_down_renderer_amd import synth
    import torch
    pkg, k = tdr
    sc = synth.make_scene("c1", n_particles=512)
    cfg = sc.cfg
    st = synth.make_particles(cfg, sc.lab, sc.pose, np.random.default_rng(12), n=24_576)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
    before = k.lib.tdr_config_shift_uniform(-1)
    try:
        k.lib.tdr_config_shift_uniform_span(-2.0)   # tuning (the default)
        k.lib.tdr_config_shift_uniform(2)
        filters = []
        for n_f in (len(st), len(st) - 4096):   # two filters of different sizes on one device, a context each
            f = pkg.ParticleFilter(n_f, m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False,
                                   locality_every=1)
            f.set_states(st[:n_f])
            perm = k.zeros((f.cap_local,), torch.int32)
            k.locality_order(f.st, n_f, m.rows, m.cols, perm)
            filters.append((f, perm, n_f, k.score_ctx_create(), [None], set()))
        for _ in range(52):   # 30 calls before the first trial, 10 trial calls, the settled state
            for f, perm, n_f, ctx, ref, spans in filters:   # in turn: neither disturbs the other's tuner
                launches = int(k.lib.tdr_shift_uniform_launches())
                k.score(m.dev, m.scan_handle(r.last_scan()), float(cfg.res), f.fp_c, f.st, n_f, f.raw_w, perm=perm,
                        uniform_scale=f._uniform_scale, n_total=n_f, ctx=ctx)
                k.synchronize()
                assert int(k.lib.tdr_shift_uniform_launches()) - launches == 1
                got = f.raw_w[:n_f].cpu().numpy()
                if ref[0] is None:
                    ref[0] = got
                assert np.array_equal(got, ref[0], equal_nan=True)
                spans.add(ctx.span())
        for f, perm, n_f, ctx, ref, spans in filters:
            assert ctx.span() in (2.0, 8.0, 16.0, 24.0, 40.0)
            assert len(spans) >= 1
        # the same particles in both filters: the same weights
        assert np.array_equal(filters[0][4][0][: filters[1][2]], filters[1][4][0], equal_nan=True)
    finally:
        k.lib.tdr_config_shift_uniform(before)
        k.lib.tdr_config_shift_uniform_span(-2.0)
