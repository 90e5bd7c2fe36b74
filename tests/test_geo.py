"""renderGeometricTopDown: ground and obstacle images of a scan, polar
(src/scan_renderer_polar.cpp:6-81) and Cartesian (src/scan_renderer.cpp:7-53).

CPU: the oracle's restatement on hand-derived cases (every expected count below is worked out from the reference's
statements in the comments).  GPU: the HIP kernels (csrc/tdr_geo.hip) against the oracle, array equality, on organised
and unorganised clouds, with range ties, origin points, non-finite points and empty clouds.  Parity unpinned like the
rest of the path: the reference holds no fixture for these functions (and its node never calls them)."""
import numpy as np
import pytest


def _lidar(rng, width=512, height=32, bumps=0.1):
    """An organised spinning-LiDAR cloud in PCL order (element idy*width + idx): beams (idy) x azimuth steps (idx) over a
    ground plane 2 m below the sensor with random vertical obstacles."""
    az = np.linspace(-np.pi, np.pi, width, endpoint=False)
    el = np.linspace(-0.45, 0.15, height)
    pts = np.zeros((height, width, 4), np.float32)
    for j in range(height):
        r = np.where(el[j] < 0, 2.0 / np.maximum(1e-3, -np.tan(el[j])), 35.0) + rng.normal(0, 0.05, width)
        r = np.minimum(r, 45.0)
        pts[j, :, 0] = r * np.sin(az)
        pts[j, :, 1] = r * np.cos(az)
        pts[j, :, 2] = r * np.tan(el[j]) + (rng.random(width) < bumps) * rng.uniform(0, 4, width)
    pts[rng.random((height, width)) < 0.02] = 0          # dropped returns: (0, 0, 0)
    return pts.reshape(-1, 4), width, height


# ---- oracle, hand-derived ---------------------------------------------------------------------------------------------
def test_oracle_geo_polar_hand_case(oracle):
    nb, nr = 4, 8
    ang = np.float32(np.pi / 2)
    # theta = atan2(x, y) = 0 -> bin round(0) + nb/2 = 2 for all three.  Sorted by range descending: A (r=5), B (2), C (1).
    #  A: from the origin, dist 5, slope 0.5/5 = 0.1 < 0.3 -> ground cells last_r_ind(0)..5 of row 2 (:67-72)
    #  B: from A, dist 3, slope 2.5/3 = 0.83: neither -> flag cleared (:73-75)
    #  C: from B, dist 1, slope 2/1 = 2 > 1 -> obstacle at range bin 1 (:62-66)
    pts = np.asarray([[0, 2, 3, 0], [0, 5, 0.5, 0], [0, 1, 5, 0]], np.float32)
    g = oracle.raster_geo_polar(pts, 3, 1, 1.0, ang, nb, nr).reshape(2, nr, nb)     # [img][range][theta]
    exp_ground = np.zeros((nr, nb), np.float32)
    exp_ground[0:6, 2] = 1
    exp_obst = np.zeros((nr, nb), np.float32)
    exp_obst[1, 2] = 1
    assert np.array_equal(g[0], exp_ground) and np.array_equal(g[1], exp_obst)
    # a steep first return sets the flag, so the flat return right behind it is NOT ground (:67 `last_high_grad == false`)
    pts = np.asarray([[0, 6, 7, 0], [0, 3, 7.1, 0]], np.float32)
    g = oracle.raster_geo_polar(pts, 2, 1, 1.0, ang, nb, nr).reshape(2, nr, nb)
    assert g[0].sum() == 0 and g[1][6, 2] == 1 and g[1].sum() == 1
    # theta is clamped into the image (:36-37): atan2(-1e-3, -1) ~ -pi -> round(-2) + 2 = 0; +pi -> 4 -> clamped to 3
    pts = np.asarray([[-1e-3, -1, 0, 0], [1e-3, -1, 0, 0]], np.float32)
    g = oracle.raster_geo_polar(pts, 2, 1, 1.0, ang, nb, nr).reshape(2, nr, nb)
    assert g[0][:2, 0].tolist() == [1, 1] and g[0][:2, 3].tolist() == [1, 1] and g[0].sum() == 4


def test_oracle_geo_cart_hand_case(oracle):
    rows, cols = 9, 9
    # one scan line (width 1, height 3), res 1, image centre (4, 4):
    #  P0 (3, 0, 0.1): x_ind 7, y_ind 4; from the origin dist 3, slope 0.033 -> ground along (4,4)->(7,4): diff (3,0),
    #     norm 3, i = 0, 1/3, 2/3 -> cells x = 4, 5, 6 at y = 4 (the end point itself is not drawn, `i < 1`)
    #  P1 (3, 2, 0.2): (7, 6); dist 2, slope 0.05 -> ground (7,4)->(7,6): norm 2, i = 0, 0.5 -> (7,4), (7,5)
    #  P2 (3, 2.5, 4): (7, 7)  [round(2.5) = 3]; dist 0.5, slope 7.6 > 1 -> obstacle at (y 7, x 7)
    pts = np.asarray([[3, 0, 0.1, 0], [3, 2, 0.2, 0], [3, 2.5, 4, 0]], np.float32)
    g = oracle.raster_geo_cart(pts, 1, 3, 1.0, rows, cols).reshape(2, cols, rows)     # [img][x][y]
    exp = np.zeros((cols, rows), np.float32)
    for x, y in ((4, 4), (5, 4), (6, 4), (7, 4), (7, 5)):
        exp[x, y] += 1
    assert np.array_equal(g[0], exp)
    assert g[1][7, 7] == 1 and g[1].sum() == 1
    # a return in the very cell of the previous one: diff (0,0), norm 0 -> 1./0 = inf -> exactly one iteration (i = 0)
    pts = np.asarray([[1, 1, 0.0, 0], [1.2, 1.1, 0.01, 0]], np.float32)
    g = oracle.raster_geo_cart(pts, 1, 2, 1.0, rows, cols).reshape(2, cols, rows)
    assert g[0][5, 5] == 1 and g[0][4, 4] == 1 and g[0].sum() == 2


def test_oracle_geo_skips_and_drops(oracle):
    rng = np.random.default_rng(3)
    pts, w, h = _lidar(rng, 64, 8)
    a = oracle.raster_geo_polar(pts, w, h, 1.0, np.float32(2 * np.pi / 32), 32, 40)
    bad = pts.copy()
    extra = np.asarray([[np.nan, 1, 0, 0], [1, np.inf, 0, 0], [0, 0, 5, 0]], np.float32)
    bad = np.concatenate([bad, np.repeat(extra, w, axis=0)[: 3 * w]])          # three more full "beams" of junk
    with np.errstate(all="ignore"):
        b = oracle.raster_geo_polar(bad, w, h + 3, 1.0, np.float32(2 * np.pi / 32), 32, 40)
    assert np.array_equal(a, b)                                                # junk returns change nothing
    assert not oracle.raster_geo_polar(np.zeros((0, 4), np.float32), 0, 0, 1.0, np.float32(0.1), 8, 8).any()


# ---- GPU --------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tdr():
    import torch
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels
    assert torch.cuda.is_available()
    return pkg, HipKernels()


def _imgs(n, rows, cols):
    return [np.full((rows, cols), 7.0, np.float32, order="F") for _ in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_geo_render_matches_oracle(tdr, oracle, seed):
    pkg, k = tdr
    rng = np.random.default_rng(100 + seed)
    width, height = int(rng.choice([64, 300, 1024])), int(rng.choice([1, 16, 64]))
    pts, w, h = _lidar(rng, width, height, bumps=float(rng.choice([0.0, 0.1, 0.5])))
    if seed % 2:          # range ties (duplicated returns) and junk: NaN / inf coordinates, far returns
        dup = rng.integers(0, len(pts), len(pts) // 4)
        pts[rng.integers(0, len(pts), len(dup))] = pts[dup]
        junk = rng.integers(0, len(pts), 40)
        pts[junk[:10], 0] = np.nan
        pts[junk[10:20], 1] = np.inf
        pts[junk[20:30], :2] = rng.normal(0, 1e5, (10, 2)).astype(np.float32)
        pts[junk[30:], 2] = np.nan
    lut = -np.ones(256, np.int32)
    res = float(rng.choice([0.5, 1.0, 2.0]))
    # polar
    nb, nr = int(rng.choice([36, 100, 256])), int(rng.choice([25, 64, 128]))
    ang = np.float32(2 * np.pi / nb)
    with np.errstate(all="ignore"):
        ref = oracle.raster_geo_polar(pts, w, h, res, ang, nb, nr)
    r = pkg.ScanRendererPolar(lut, kernels=k)
    imgs = _imgs(3, nb, nr)
    dev = r.renderGeometricTopDown(pts, res, ang, imgs, width=w, height=h)
    assert np.array_equal(dev.cpu().numpy(), ref)
    assert np.array_equal(np.stack([im.ravel(order="F") for im in imgs[:2]]), ref) and not imgs[2].any()
    assert ref[0].sum() > 0
    # Cartesian
    rows, cols = int(rng.choice([50, 128])), int(rng.choice([64, 100]))
    with np.errstate(all="ignore"):
        refc = oracle.raster_geo_cart(pts, w, h, res, rows, cols)
    rc = pkg.ScanRenderer(lut, kernels=k)
    imgs = _imgs(2, rows, cols)
    dev = rc.renderGeometricTopDown(pts, res, imgs, width=w, height=h)
    assert np.array_equal(dev.cpu().numpy(), refc)
    assert np.array_equal(np.stack([im.ravel(order="F") for im in imgs]), refc)
    # pcl::PointXYZI-strided input (8 floats per point) gives the same images
    pcl = np.zeros((len(pts), 8), np.float32)
    pcl[:, :3] = pts[:, :3]
    assert np.array_equal(rc.renderGeometricTopDown(pcl, res, _imgs(2, rows, cols), width=w, height=h).cpu().numpy(), refc)


@pytest.mark.gpu
def test_geo_render_edge_cases(tdr, oracle):
    pkg, k = tdr
    lut = -np.ones(256, np.int32)
    r = pkg.ScanRendererPolar(lut, kernels=k)
    imgs = _imgs(2, 16, 8)
    r.renderGeometricTopDown(np.zeros((0, 4), np.float32), 1.0, np.float32(0.4), imgs)         # empty cloud: zeros
    assert not np.stack(imgs).any()
    one = _imgs(1, 16, 8)
    assert r.renderGeometricTopDown(np.ones((5, 4), np.float32), 1.0, np.float32(0.4), one) is None   # < 2 images (:8)
    assert (one[0] == 7.0).all()                                                               # ... untouched
    # the hand cases of the CPU tests, through the kernels
    pts = np.asarray([[0, 2, 3, 0], [0, 5, 0.5, 0], [0, 1, 5, 0]], np.float32)
    ang = np.float32(np.pi / 2)
    dev = r.renderGeometricTopDown(pts, 1.0, ang, _imgs(2, 4, 8))
    assert np.array_equal(dev.cpu().numpy(), oracle.raster_geo_polar(pts, 3, 1, 1.0, ang, 4, 8))
    rc = pkg.ScanRenderer(lut, kernels=k)
    pts = np.asarray([[3, 0, 0.1, 0], [3, 2, 0.2, 0], [3, 2.5, 4, 0]], np.float32)
    dev = rc.renderGeometricTopDown(pts, 1.0, _imgs(2, 9, 9), width=1, height=3)
    assert np.array_equal(dev.cpu().numpy(), oracle.raster_geo_cart(pts, 1, 3, 1.0, 9, 9))
    with pytest.raises(ValueError):
        rc.renderGeometricTopDown(pts, 1.0, _imgs(2, 9, 9), width=2, height=3)


# ---- N4: geometric layers of the map, getLocalGeoMap, the geometric score term -------------------------------------------
def _geo_scene(seed=0, ncls=6, nb=48, nr=32, size=260, n=600):
    from top_down_renderer_amd import synth
    cfg = synth.Config("geo", 4000, ncls, nb, nr, size, n, seed=4100 + seed)
    return cfg, synth.make_scene(cfg)


def test_oracle_geo_maps_definition(oracle):
    """geo_maps_ (src/top_down_map.cpp:48-58, 410-427): layer 1 = distance to the nearest cell of a flattened class >= 3,
    layer 0 = distance to the nearest cell without one, truncated at 50, nothing masked."""
    maps = np.full((5, 6, 9), 50.0, np.float32)
    mask = np.zeros((6, 9), np.uint8)
    maps[3][2, 4] = 0.0                 # one cell of geometric class 3 ...
    maps[1][0, 0] = 0.0                 # ... and a road cell (class 1 is not geometric)
    maps[4][5, 8] = 0.0
    mask[5, 8] = 1                      # class 4 "present" on an unknown cell does not count (its 0 is the mask's)
    g = oracle.geo_maps_from_class_maps(maps, mask, 1.0)
    g0, g1 = g.maps_cm[0].T, g.maps_cm[1].T           # back to [row, col]
    assert g1[2, 4] == 0 and g1[2, 5] == 1 and np.isclose(g1[0, 0], np.hypot(2, 4))
    assert g0[2, 4] == 1 and g0[0, 0] == 0 and g0[5, 8] == 0
    assert not g.mask_cm.any()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,resolution", [(0, 1.0), (1, 0.5), (2, 2.0)])
def test_geo_layers_and_local_geo_map_match_oracle(tdr, oracle, seed, resolution):
    pkg, k = tdr
    from top_down_renderer_amd import synth
    cfg = synth.Config("geo", 3000, 6, 40, 24, 220, 8, seed=4200 + seed, map_resolution=resolution)
    sc = synth.make_scene(cfg, with_particles=False)
    rows, cols = 200, 170
    maps = np.ascontiguousarray(sc.class_maps[:, :rows, :cols])
    mask = np.ascontiguousarray(sc.class_mask[:rows, :cols])
    geo_o = oracle.geo_maps_from_class_maps(maps, mask, resolution)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=resolution), maps, mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    gm, gk = k.unpack_map(m.geo_dev())                       # (2, cols, rows) column-major like the oracle's
    assert np.array_equal(gm, geo_o.maps_cm) and not gk.any()
    tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, resolution)
    rng = np.random.default_rng(seed)
    for _ in range(6):
        cx, cy = rng.uniform(-20, cols * resolution + 20), rng.uniform(-20, rows * resolution + 20)
        scale, res = float(rng.uniform(0.5, 2.5)), float(rng.choice([0.5, 1.0, 1.5]))
        d_o, _ = oracle.local_map_polar(geo_o, tab, cx, cy, scale, res)
        d = m.getLocalGeoMap((cx, cy), scale, res)
        assert np.array_equal(np.stack([x.T.ravel() for x in d]), d_o)
    mc = pkg.TopDownMap(pkg.Params(resolution=resolution), maps, mask, kernels=k)
    d_o, _ = oracle.local_map_cart(geo_o, 60.0, 70.0, 0.7, 1.0, 21, 17)
    d = mc.getLocalGeoMap((60.0, 70.0), 0.7, 1.0, (21, 17))
    assert np.array_equal(np.stack([x.T.ravel() for x in d]), d_o)
    # the dynamic-map path (updateMap with a label image) leaves both layers at 1 (src/top_down_map.cpp:126-133)
    lab = np.where(sc.lab[:rows, :cols] >= 0, sc.lab[:rows, :cols], 200).astype(np.uint8)[::-1].copy()
    md = pkg.TopDownMapPolar(pkg.Params(resolution=1.0, flatten_lut=list(sc.lut), num_classes=6), kernels=k)
    md.updateMap(lab, None, (0, 0))
    md.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    d = md.getLocalGeoMap((cols / 2, rows / 2), 1.0, 1.0)
    inside = np.stack(d)[:, :, :8]
    assert (inside == 1.0).all() and set(np.unique(np.stack(d))) <= {0.0, 1.0}


@pytest.mark.gpu
@pytest.mark.parametrize("seed,uninit", [(0, False), (1, True), (2, True)])
def test_geometric_cost_term_matches_oracle(tdr, oracle, seed, uninit):
    """getCostForRot with its geometric block (src/state_particle.cpp:145-152) switched on: weights within 1e-5 of the
    oracle; with un-initialised particles the 40-rotation search prices the geometric term too."""
    pkg, k = tdr
    cfg, sc = _geo_scene(seed, ncls=[6, 5, 7][seed])
    rng = np.random.default_rng(seed)
    st = sc.states.copy()
    st["dx_m"] = rng.normal(0, 2, len(st)).astype(np.float32)
    far = rng.random(len(st)) < 0.1
    st["init_x_px"][far] = rng.uniform(-300, 600, int(far.sum())).astype(np.float32)
    if uninit:
        st["have_init"][rng.random(len(st)) < 0.4] = 0
    # geometric images of an organised scan of the same scene
    pts, w, h = _lidar(rng, 256, 16, bumps=0.3)
    ang = cfg.ang_res
    geo = oracle.raster_geo_polar(pts, w, h, cfg.res, ang, cfg.nb, cfg.nr)
    assert geo[0].sum() > 0 and geo[1].sum() > 0
    scan = oracle.raster_polar(sc.pts, cfg.res, ang, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0)
    geo_o = oracle.geo_maps_from_class_maps(sc.class_maps, sc.class_mask, 1.0)
    tab = oracle.polar_table(cfg.nb, cfg.nr, ang, 1.0)
    cw = [float(x) for x in rng.uniform(0.3, 2.0, cfg.ncls)]
    params = dict(fixed_scale=1.0, class_weights=cw, regularization=0.3)
    st_o = st.copy()
    ref = oracle.compute_weights_geo(om, geo_o, tab, cfg.nb, cfg.nr, scan, geo, cfg.res, oracle.make_params(cfg.ncls, **params),
                                     st_o)
    plain = oracle.compute_weights(om, tab, cfg.nb, cfg.nr, scan, cfg.res, oracle.make_params(cfg.ncls, **params), st.copy())
    ok = ~np.isnan(ref)
    assert np.abs(ref[ok] - plain[ok]).max() > 1e-3            # the term does change the weights
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), ang)
    geo_imgs = [geo[i].reshape(cfg.nr, cfg.nb).T.copy() for i in range(2)]
    f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(**params), kernels=k, init_particles=False, locality_every=1,
                           use_geometric_cost=True)
    f.set_states(st)
    f.update(scan, geo_imgs, cfg.res)
    got = f.raw_weights()
    pre = k.states_to_host(f.st_new, len(st), pkg.STATE_DTYPE)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    same = pre["theta"] == st_o["theta"]   # (no floor on how many agree: every mismatch must be a tie, below)
    err = np.abs(got[ok & same] - ref[ok & same]) / np.abs(ref[ok & same])
    assert err.max(initial=0.0) <= 1e-5, err.max()
    if not same.all():   # another candidate of the search: the weight is the oracle's at that rotation, which is a near-tie
        st2 = st_o.copy()
        st2["theta"], st2["have_init"] = pre["theta"], 1
        ref2 = oracle.compute_weights_geo(om, geo_o, tab, cfg.nb, cfg.nr, scan, geo, cfg.res,
                                          oracle.make_params(cfg.ncls, **params), st2)
        d = ok & ~same
        assert (np.abs(got[d] - ref2[d]) / np.abs(ref2[d])).max() <= 1e-5
        assert (np.abs(ref2[d] - ref[d]) / np.abs(ref[d])).max() <= 2e-5
    # without the switch top_down_geo is ignored like in the reference
    f0 = pkg.ParticleFilter(len(st), m, pkg.FilterParams(**params), kernels=k, init_particles=False)
    f0.set_states(st)
    f0.update(scan, geo_imgs, cfg.res)
    w0 = f0.raw_weights()
    assert np.allclose(w0[ok], plain[ok], rtol=1e-5, atol=0)


@pytest.mark.gpu
def test_map_cache_round_trip_in_the_reference_format(tdr, oracle, tmp_path):
    """saveCachedMaps / loadCachedMaps (src/top_down_map.cpp:226-286) through the C ABI: cached_data.txt + class_map<i>.eig
    + geo_map<i>.eig + class_mask.eig, readable by the Python reader of the same format; a cache for other parameters is
    not loaded."""
    import ctypes as C
    pkg, k = tdr
    from top_down_renderer_amd import eig_io, synth
    from top_down_renderer_amd._lib import check
    L = k.lib
    cfg = synth.Config("cache", 100, 4, 16, 8, 90, 4, seed=77)
    sc = synth.make_scene(cfg, with_particles=False)
    ncls, H, W = sc.class_maps.shape
    maps_cm = np.ascontiguousarray(np.transpose(sc.class_maps, (0, 2, 1)), np.float32)
    mask_cm = np.ascontiguousarray(sc.class_mask.T, np.uint8)
    vp = C.c_void_p
    m = vp()
    check(L.tdr_map_create(C.byref(m)))
    check(L.tdr_map_set(m, maps_cm.ctypes.data_as(vp), mask_cm.ctypes.data_as(vp), ncls, H, W, C.c_float(1.0), 3, 4))
    d = str(tmp_path).encode()
    check(L.tdr_map_save_cache(m, d, b"/maps/site.svg"))
    got_maps, got_mask = eig_io.load_cached_maps(str(tmp_path), ncls)
    assert np.array_equal(got_maps, sc.class_maps) and np.array_equal(got_mask, sc.class_mask)
    geo_o = oracle.geo_maps_from_class_maps(sc.class_maps, sc.class_mask, 1.0)
    for i in range(2):
        assert np.array_equal(eig_io.read_eig(str(tmp_path / f"geo_map{i}.eig"), np.float32).T, geo_o.maps_cm[i])
    m2, loaded = vp(), C.c_int(-1)
    check(L.tdr_map_create(C.byref(m2)))
    check(L.tdr_map_load_cache(m2, d, b"/maps/other.svg", ncls, C.c_float(1.0), 0, 0, C.byref(loaded)))
    assert loaded.value == 0
    check(L.tdr_map_load_cache(m2, d, b"/maps/site.svg", ncls + 1, C.c_float(1.0), 0, 0, C.byref(loaded)))
    assert loaded.value == 0
    check(L.tdr_map_load_cache(m2, d, b"/maps/site.svg", ncls, C.c_float(1.0), 3, 4, C.byref(loaded)))
    assert loaded.value == 1
    dist = np.zeros(ncls * 12 * 10, np.float32)
    msk = np.zeros(12 * 10, np.uint8)
    ref_d = np.zeros_like(dist)
    ref_m = np.zeros_like(msk)
    for handle, dd, mm in ((m, ref_d, ref_m), (m2, dist, msk)):
        check(L.tdr_map_local_map(handle, 0, C.c_float(40.0), C.c_float(35.0), C.c_float(0.4), C.c_float(1.0), 12, 10,
                                  dd.ctypes.data_as(vp), mm.ctypes.data_as(vp)))
    assert np.array_equal(dist, ref_d) and np.array_equal(msk, ref_m)
    L.tdr_map_destroy(m)
    L.tdr_map_destroy(m2)
