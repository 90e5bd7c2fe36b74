"""What the integer form of a polar launch reads and promises (csrc/tdr_cmap.hip, tdr_score_ray.hip, tdr_score_su.hip):
the class planes and the integer dictionary decode to the map's own values (from the layout include/tdr.h documents), and a
particle's weight is the same bits whatever order its products were added in — any split of a window over waves, any
particle order, either kernel, one caller context or none.  Run with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_TOTAL = 1_000_000
ALL_RAY = 1e-6


@pytest.fixture(scope="module")
def tdr():
    import torch
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return pkg, HipKernels()


def _scene(ncls=6, nb=64, nr=32, size=300, n=3000, seed=5100, pts=6000):
    from top_down_renderer_amd import synth
    cfg = synth.Config("ray", pts, ncls, nb, nr, size, n, seed=seed)
    return synth.make_scene(cfg)


@pytest.mark.parametrize("ncls,rows,cols", [(6, 300, 300), (3, 97, 131), (9, 64, 200)])
def test_class_planes_and_integer_dictionary_decode_to_the_map(tdr, ncls, rows, cols):
    """Read the device buffers back and decode them on the host with nothing but the layout of include/tdr.h."""
    pkg, k = tdr
    rng = np.random.default_rng(rows)
    d2 = rng.integers(0, 60, (ncls, rows, cols)) ** 2 + rng.integers(0, 12, (ncls, rows, cols))   # < 1000 distinct values
    maps = np.minimum(50.0, np.sqrt(d2.astype(np.float64))).astype(np.float32)
    mask = (rng.random((rows, cols)) < 0.15).astype(np.uint8)
    maps[:, mask == 1] = 0
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), maps, mask, kernels=k)
    lib = k.lib
    assert m.dev.desc.cwords > 0
    off = int(lib.tdr_cmap_plane_offset_words(ncls, rows, cols))
    pw = int(lib.tdr_cmap_plane_words(ncls, rows, cols))
    cmw = int(lib.tdr_cmap_cmask_words(ncls, rows, cols))
    assert pw > 0 and cmw > 0 and int(lib.tdr_cmap_words_total(ncls, rows, cols)) == off + pw * ncls + cmw
    crec = m.dev.crec.cpu().numpy().view(np.uint32)
    dic = m.dev.dict.cpu().numpy()
    n = int(m.dev.desc.dict_n)
    fdict = dic[:1024].view(np.float32)
    idict = dic[1024:2048].view(np.uint32)
    q, ok = int(dic[2048:2049].view(np.uint32)[0]), int(dic[2049:2050].view(np.uint32)[0])
    assert ok == 1 and 0 <= q <= 23
    assert np.array_equal(idict[:n].astype(np.float64), fdict[:n].astype(np.float64) * 2.0 ** q)   # exact: integers
    # planes: cell (r, c) of class k at 16-bit element  k * 2 pw + ((c' >> 3) * trows + (r' >> 3)) * 64 + (r' & 7) * 8 + (c' & 7)
    planes = crec[off: off + pw * ncls].view(np.uint16)
    trows = (rows >> 3) + 2
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    rp, cp = r + 8, c + 8
    el = ((cp >> 3) * trows + (rp >> 3)) * 64 + (rp & 7) * 8 + (cp & 7)
    for kk in range(ncls):
        v = planes[kk * 2 * pw + el]
        assert np.array_equal(v >> 15, 1 - mask)
        assert np.array_equal(fdict[(v >> 2) & 0x3FF], maps[kk])
    # the guard band: unknown, distance 0
    gr = np.array([-1, -1, rows, rows, 5]); gc = np.array([-1, cols, -1, cols, -1])
    el = (((gc + 8) >> 3) * trows + ((gr + 8) >> 3)) * 64 + ((gr + 8) & 7) * 8 + ((gc + 8) & 7)
    assert (planes[el] == 0).all()
    # the coarse mask plane behind the class planes: cell (r >> 2, c >> 2), bit (r & 3) * 4 + (c & 3), the planes' stride
    cm = crec[off + pw * ncls: off + pw * ncls + cmw].view(np.uint16)
    Rp, Cp = (r >> 2) + 8, (c >> 2) + 8
    el = ((Cp >> 3) * trows + (Rp >> 3)) * 64 + (Rp & 7) * 8 + (Cp & 7)
    assert np.array_equal((cm[el] >> ((r & 3) * 4 + (c & 3))) & 1, 1 - mask)
    gR, gC = (gr >> 2) + 8, (gc >> 2) + 8
    assert ((cm[((gC >> 3) * trows + (gR >> 3)) * 64 + (gR & 7) * 8 + (gC & 7)] >> ((gr & 3) * 4 + (gc & 3))) & 1 == 0).all()


def test_a_map_without_an_integer_form_says_so(tdr, oracle):
    """Distance values are multiples of 2^-q with value 2^q below 2^32 for every resolution a distance map is built at
    (0.05 m per cell included: q = 28).  A map that also holds 1e-6 is not: its dictionary's flag stays 0, the device raises
    `inexact` and the float kernel scores the launch (equal to the float kernel on its own)."""
    pkg, k = tdr
    sc = _scene(n=1500)
    cfg = sc.cfg
    fine = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), (sc.class_maps * np.float32(0.05)).astype(np.float32),
                               sc.class_mask, kernels=k)
    if fine.dev.desc.cwords and fine.dev.desc.dict_n <= 1024:
        tail = fine.dev.dict.cpu().numpy()[2048:2050].view(np.uint32)
        assert int(tail[1]) == 1 and 24 <= int(tail[0]) <= 30
    maps = sc.class_maps.copy()
    maps[1, 100:140, 90:160] = np.float32(1e-6)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), maps, sc.class_mask, kernels=k)
    assert m.dev.desc.cwords > 0
    dic = m.dev.dict.cpu().numpy()
    assert int(dic[2049:2050].view(np.uint32)[0]) == 0
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    got = []
    before = k.lib.tdr_config_shift_uniform(-1)
    try:
        for mode in (0, 2):
            k.lib.tdr_config_shift_uniform(mode)
            f = pkg.ParticleFilter(len(sc.states), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False)
            f.set_states(sc.states)
            k.score(m.dev, m.scan_handle(scan), float(cfg.res), f.fp_c, f.st, len(sc.states), f.raw_w, uniform_scale=1.0,
                    n_total=N_TOTAL)
            k.synchronize()
            got.append(f.raw_w[: len(sc.states)].cpu().numpy())
    finally:
        k.lib.tdr_config_shift_uniform(before)
    assert np.array_equal(got[0], got[1], equal_nan=True)


@pytest.mark.parametrize("ncls,nb,nr,scale_fixed", [(6, 64, 32, True), (4, 48, 20, False), (2, 100, 24, True), (11, 24, 12, True)])
def test_weights_do_not_depend_on_the_order_of_the_additions(tdr, oracle, ncls, nb, nr, scale_fixed):
    """One set of particles scored (a) by the shift-uniform kernel, (b)-(e) by the ray-mapped kernel with a window split
    over 1, 2, 4 and 8 waves, (f) in a mixed launch, (g) in another particle order, (h) in a mixed launch with a caller's
    context (sample offsets multiplied out of the table's factors, block-major rows): eight different orders of the same products, one set of bits — and the oracle's
    weights to 1e-5."""
    import torch
    pkg, k = tdr
    sc = _scene(ncls=ncls, nb=nb, nr=nr, seed=5100 + ncls)
    cfg = sc.cfg
    st = sc.states.copy()
    n = len(st)
    rng = np.random.default_rng(ncls)
    if not scale_fixed:
        st["scale"] = rng.uniform(0.7, 1.5, n).astype(np.float32)
    st["init_x_px"][::9] = rng.uniform(-100, 400, len(st[::9])).astype(np.float32)   # borders, outside
    params = dict(fixed_scale=1.0 if scale_fixed else -1.0)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    ref = oracle.compute_weights(oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0),
                                 oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res), cfg.nb, cfg.nr, scan, cfg.res,
                                 oracle.make_params(cfg.ncls, **params), st.copy())
    f = pkg.ParticleFilter(n, m, pkg.FilterParams(**params), kernels=k, init_particles=False, locality_every=1)
    f.set_states(st)
    loc = k.zeros((f.cap_local,), torch.int32)
    k.locality_order(f.st, n, m.rows, m.cols, loc)
    shuffled = k.to_device(rng.permutation(n).astype(np.int32))
    pk = m.scan_handle(scan)

    def run(span, split=0, perm=loc, ctx=None):
        k.lib.tdr_config_shift_uniform_span(span)
        k.lib.tdr_config_ray_split(split)
        f.raw_w.fill_(-7.0)
        k.score(m.dev, pk, float(cfg.res), f.fp_c, f.st, n, f.raw_w, perm=perm, uniform_scale=f._uniform_scale,
                n_total=N_TOTAL, ctx=ctx)
        k.synchronize()
        return f.raw_w[:n].cpu().numpy()

    before = k.lib.tdr_config_shift_uniform(-1)
    try:
        k.lib.tdr_config_shift_uniform(2)
        a = run(0.0)
        assert not (a == -7.0).any()
        for split in (1, 2, 4, 8):
            assert np.array_equal(a, run(ALL_RAY, split), equal_nan=True), f"ray kernel, {split} waves per particle"
        assert np.array_equal(a, run(3.0), equal_nan=True)
        assert np.array_equal(a, run(3.0, perm=shuffled), equal_nan=True)
        assert np.array_equal(a, run(ALL_RAY, 2, perm=None), equal_nan=True)
        assert np.array_equal(a, run(5.0, ctx=k.score_ctx_create()), equal_nan=True)
        # the re-routing pass (tdr_config_tuning("su_wave_span"): waves whose own particles spread too far go to the ray-mapped
        # kernel after all; off by default — measured, it does not pay): some, then every wave re-routed
        for cells in (6, 1):
            k.lib.tdr_config_tuning(b"su_wave_span", cells)
            assert np.array_equal(a, run(40.0), equal_nan=True), f"waves wider than {cells} cells re-routed"
    finally:
        k.lib.tdr_config_shift_uniform(before)
        k.lib.tdr_config_shift_uniform_span(-2.0)
        k.lib.tdr_config_ray_split(0)
        k.lib.tdr_config_tuning(b"su_wave_span", 0)
    assert np.array_equal(np.isnan(a), np.isnan(ref))
    ok = ~np.isnan(ref)
    err = np.abs(a[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-30)
    assert err.max(initial=0.0) <= 1e-5, err.max()


@pytest.mark.parametrize("nb,nr,scale_fixed", [(64, 32, True), (100, 100, True), (40, 200, False), (12, 300, True),
                                               (9, 530, False), (32, 130, False), (256, 40, True), (48, 70, True), (16, 300, True), (96, 17, False),
                                               (240, 64, True), (128, 129, False)])
def test_offsets_multiplied_out_of_the_tables_factors(tdr, oracle, nb, nr, scale_fixed):
    """A context that holds the table's factors (tdr_polar_factors_host): the ray-mapped kernel multiplies a direction's
    pair with a ring's radius itself instead of reading the product — the same float products (checked here on the host,
    and by the kernel's preparation on the device), so the same bits as the kernel that reads the table; one, two and four
    rings per lane, one block and several.  With factors that are NOT the table's (another angular resolution) the
    preparation notices and the table is read: the same bits again.  Direction counts that are multiples of 16 take the
    PATCH order (a step = 4 directions x 16 rings, a lane's four descriptors of a unit in one load, radii per ring block;
    ragged ring counts pad the last block) — switched off and on here, like the borrowed class planes of empty bins
    (tdr_config_tuning "ray_patch" / "ray_borrow"): the same bits every way."""
    import torch
    pkg, k = tdr
    sc = _scene(ncls=5, nb=nb, nr=nr, size=400, n=1200, seed=5300 + nr, pts=9000)
    cfg = sc.cfg
    st = sc.states.copy()
    n = len(st)
    rng = np.random.default_rng(nr)
    if not scale_fixed:
        st["scale"] = rng.uniform(0.4, 1.2, n).astype(np.float32)
    st["init_x_px"][::7] = rng.uniform(-100, 500, len(st[::7])).astype(np.float32)
    params = dict(fixed_scale=1.0 if scale_fixed else -1.0)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((nb, nr), cfg.ang_res)
    fac = m.dev.fac.cpu().numpy()
    prod = np.stack([np.outer(fac[2 * nb:], fac[0:2 * nb:2]), np.outer(fac[2 * nb:], fac[1:2 * nb:2])], axis=-1)
    assert np.array_equal(prod.reshape(-1, 2).view(np.uint32), m.dev.tab_host.view(np.uint32))
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, nb, nr)
    f = pkg.ParticleFilter(n, m, pkg.FilterParams(**params), kernels=k, init_particles=False, locality_every=1)
    f.set_states(st)
    pk = m.scan_handle(scan)
    ctx = k.score_ctx_create()

    def run(span, split, ctx):
        k.lib.tdr_config_shift_uniform_span(span)
        k.lib.tdr_config_ray_split(split)
        f.raw_w.fill_(-7.0)
        k.score(m.dev, pk, float(cfg.res), f.fp_c, f.st, n, f.raw_w, uniform_scale=f._uniform_scale, n_total=N_TOTAL, ctx=ctx)
        k.synchronize()
        return f.raw_w[:n].cpu().numpy()

    before = k.lib.tdr_config_shift_uniform(-1)
    good = m.dev.fac
    try:
        k.lib.tdr_config_shift_uniform(2)
        a = run(ALL_RAY, 1, None)
        assert not (a == -7.0).any()
        for patch, borrow in ((1, 1), (0, 1), (1, 0), (0, 0)):
            k.lib.tdr_config_tuning(b"ray_patch", patch)
            k.lib.tdr_config_tuning(b"ray_borrow", borrow)
            for split in (1, 2, 8):
                assert np.array_equal(a, run(ALL_RAY, split, ctx), equal_nan=True), \
                    f"factors, {split} waves per particle, patch order {patch}, borrowed planes {borrow}"
            assert np.array_equal(a, run(3.0, 0, ctx), equal_nan=True)
        k.lib.tdr_config_tuning(b"ray_patch", 1)
        k.lib.tdr_config_tuning(b"ray_borrow", 1)
        other = np.empty(2 * nb + nr, np.float32)
        rc = k.lib.tdr_polar_factors_host(nb, nr, C.c_float(cfg.ang_res * 1.01), C.c_float(1.0), other.ctypes.data_as(C.c_void_p))
        assert rc == 0
        m.dev.fac = k.to_device(other)
        assert np.array_equal(a, run(ALL_RAY, 2, ctx), equal_nan=True), "factors of another table"
    finally:
        m.dev.fac = good
        k.lib.tdr_config_shift_uniform(before)
        k.lib.tdr_config_shift_uniform_span(-2.0)
        k.lib.tdr_config_ray_split(0)
        k.lib.tdr_config_tuning(b"ray_patch", 1)
        k.lib.tdr_config_tuning(b"ray_borrow", 1)
    ref = oracle.compute_weights(oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0), oracle.polar_table(nb, nr, cfg.ang_res),
                                 nb, nr, scan, cfg.res, oracle.make_params(cfg.ncls, **params), st.copy())
    assert np.array_equal(np.isnan(a), np.isnan(ref))
    ok = ~np.isnan(ref)
    err = np.abs(a[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-30)
    assert err.max(initial=0.0) <= 1e-5, err.max()


def test_large_counts_and_many_points_in_one_bin(tdr, oracle):
    """Counts far above what a LiDAR bin holds (up to 2^24 - 1 in a bin, 3 10^7 in the image): the integer sums need their
    64 bits, and 2^24 itself has no integer form any more (the float kernel takes over)."""
    pkg, k = tdr
    sc = _scene(n=800, seed=5300)
    cfg = sc.cfg
    P = cfg.nb * cfg.nr
    rng = np.random.default_rng(3)
    scan = np.zeros((cfg.ncls, P), np.float32)
    hot = rng.choice(P, 40, replace=False)
    scan[rng.integers(0, cfg.ncls, 40), hot] = rng.integers(1 << 18, 1 << 20, 40).astype(np.float32)
    scan[:, hot[0]] = 0
    scan[2, hot[0]] = float((1 << 24) - 1)
    scan[4, hot[1]] += 77.0                     # a bin that (most likely) holds two classes
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    st = sc.states.copy()
    ref = oracle.compute_weights(oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0),
                                 oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res), cfg.nb, cfg.nr, scan, cfg.res,
                                 oracle.make_params(cfg.ncls), st.copy())
    f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False)
    f.set_states(st)
    before = k.lib.tdr_config_shift_uniform(-1)
    try:
        k.lib.tdr_config_shift_uniform(2)
        out = []
        for span in (0.0, ALL_RAY):
            k.lib.tdr_config_shift_uniform_span(span)
            k.score(m.dev, m.scan_handle(scan), float(cfg.res), f.fp_c, f.st, len(st), f.raw_w, uniform_scale=1.0,
                    n_total=N_TOTAL)
            k.synchronize()
            out.append(f.raw_w[: len(st)].cpu().numpy())
        assert np.array_equal(out[0], out[1], equal_nan=True)
        ok = ~np.isnan(ref)
        assert np.array_equal(np.isnan(out[0]), ~ok)
        assert (np.abs(out[0][ok] - ref[ok]) <= 1e-5 * np.abs(ref[ok])).all()
        scan[2, hot[0]] = float(1 << 24)            # no integer form: the float kernel does the launch in either mode
        k.score(m.dev, m.scan_handle(scan), float(cfg.res), f.fp_c, f.st, len(st), f.raw_w, uniform_scale=1.0, n_total=N_TOTAL)
        k.lib.tdr_config_shift_uniform(0)
        big = f.raw_w[: len(st)].cpu().numpy().copy()
        k.score(m.dev, m.scan_handle(scan), float(cfg.res), f.fp_c, f.st, len(st), f.raw_w, uniform_scale=1.0, n_total=N_TOTAL)
        k.synchronize()
        assert np.array_equal(big, f.raw_w[: len(st)].cpu().numpy(), equal_nan=True)
    finally:
        k.lib.tdr_config_shift_uniform(before)
        k.lib.tdr_config_shift_uniform_span(-2.0)


@pytest.mark.parametrize("ncls,rows,cols,kind", [(6, 50, 64, "scan"), (3, 33, 21, "scan"), (9, 18, 140, "scan"),
                                                 (6, 37, 300, "dense"), (6, 24, 24, "empty"), (6, 20, 36, "fractional")])
def test_cartesian_integer_form(tdr, oracle, ncls, rows, cols, kind):
    """The Cartesian score's integer form (csrc/tdr_score_cart.hip): the skipping kernel with integer accumulators for
    dense particles, score_cart_ray_kernel (one wave per particle, lanes = consecutive window columns) for scattered ones.
    All dense, all scattered and a mixed launch give the same bits; the float kernels agree to rounding, the oracle to 1e-5.
    Window widths of 1, 2 and 4 steps per block and more than one block; a scan with several classes in every bin (the
    list); an empty scan; a scan with fractional counts (no integer form: the float kernel scores it in every mode)."""
    from top_down_renderer_amd import synth
    pkg, k = tdr
    cfg = synth.Config("cartint", 6000, ncls, rows, cols, 500, 900, polar=False, seed=177 + ncls + rows, res=0.75)
    sc = synth.make_scene(cfg)
    st = sc.states.copy()
    rng = np.random.default_rng(15)
    st["scale"] = rng.uniform(0.8, 1.25, len(st)).astype(np.float32)
    st["init_x_px"][:6] = np.asarray([-50, 5, 500, 495, 250, 0.5], np.float32)          # off / at the border
    st["init_y_px"][:6] = np.asarray([250, 250, 250, 250, -40, 0.5], np.float32)
    scan = oracle.raster_cart(sc.pts, cfg.res, sc.lut, ncls, rows, cols)
    if kind == "dense":
        scan = rng.integers(0, 3, scan.shape).astype(np.float32)
    elif kind == "empty":
        scan = np.zeros_like(scan)
    elif kind == "fractional":
        scan = (scan * np.float32(0.5)).astype(np.float32)
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0)
    with np.errstate(all="ignore"):
        ref = oracle.compute_weights_cart(om, rows, cols, scan, cfg.res, oracle.make_params(ncls), st.copy())
    m = pkg.TopDownMap(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.setWindow(rows, cols)
    before = k.lib.tdr_config_shift_uniform(-1)
    got = {}
    try:
        for name, mode, span in (("float", 0, 16.0), ("dense", 2, 0.0), ("ray", 2, ALL_RAY), ("mixed", 2, 6.0)):
            k.lib.tdr_config_shift_uniform(mode)
            k.lib.tdr_config_shift_uniform_span(span)
            f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False,
                                   locality_every=1)
            f.set_states(st)
            f.update(np.ascontiguousarray(scan, np.float32), None, cfg.res)
            got[name] = f.raw_weights()
    finally:
        k.lib.tdr_config_shift_uniform(before)
        k.lib.tdr_config_shift_uniform_span(-2.0)
    assert np.array_equal(got["dense"], got["ray"], equal_nan=True)
    assert np.array_equal(got["dense"], got["mixed"], equal_nan=True)
    if kind == "fractional":
        assert np.array_equal(got["float"], got["dense"], equal_nan=True)   # the float kernel ran in every mode
    else:
        assert np.array_equal(np.isnan(got["float"]), np.isnan(got["dense"]))
        ok = ~np.isnan(got["dense"])
        err = np.abs(got["float"][ok] - got["dense"][ok]) / np.maximum(np.abs(got["dense"][ok]), 1e-30)
        assert err.max(initial=0.0) <= 3e-6
    assert np.array_equal(np.isnan(got["dense"]), np.isnan(ref))
    ok = ~np.isnan(ref)
    err = np.abs(got["dense"][ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-30)
    assert err.max(initial=0.0) <= 1e-5, err.max()
