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
