"""np_oracle.py — TEST INFRASTRUCTURE ONLY (never imported by the product path).

Independent NumPy statement of the per-scan particle-filter update of KumarRobotics/top_down_renderer,
written from the maths recorded in SURVEY.md §8 / Appendix A (not from oracle.cpp), so that the two
derivations catch each other's indexing mistakes.  Vectorised, float32 where the reference is float32,
float64 accumulation where Eigen's reduction order is unspecified.  Meant for small cases only.

PARITY UNPINNED: the reference ships no fixtures for this path.

Reference lines followed (paths relative to the reference root):
  raster_polar      src/scan_renderer_polar.cpp:83-109
  raster_cart       src/scan_renderer.cpp:55-78
  polar_table       src/top_down_map.cpp:367-389, src/top_down_map_polar.cpp:7-19
  local_map_polar   src/top_down_map_polar.cpp:21-53
  cost_for_rot      src/state_particle.cpp:112-155
  compute_weights   src/state_particle.cpp:157-219
  update_weights    src/particle_filter.cpp:107-147
  resample          src/particle_filter.cpp:172-185
"""
import ctypes
import ctypes.util
import math

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m"))
for _n in ("cosf", "sinf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]


def _libm1(name, x):
    fn = getattr(_libm, name)
    return np.array([fn(float(v)) for v in np.asarray(x, f32).ravel()], f32).reshape(np.shape(x))


def _atan2f(a, b):
    a = np.asarray(a, f32).ravel()
    b = np.asarray(b, f32).ravel()
    return np.array([_libm.atan2f(float(u), float(v)) for u, v in zip(a, b)], f32)


def round_half_away(x):
    """C roundf on float32 data, evaluated exactly in float64."""
    x64 = np.asarray(x, np.float64)
    return (np.sign(x64) * np.floor(np.abs(x64) + 0.5)).astype(f32)


def raster_polar(pts_xyzc, res, ang_res, lut256, ncls, nb, nr):
    """pts_xyzc: (n,4) float32 x,y,z,class.  Returns (ncls, nb*nr) float32, index theta + nb*r."""
    pts = np.asarray(pts_xyzc, f32)
    x, y, cls = pts[:, 0], pts[:, 1], pts[:, 3]
    keep = ~((x == 0) & (y == 0))
    x, y, cls = x[keep], y[keep], cls[keep]
    theta = _atan2f(x, y)
    r = np.sqrt(x * x + y * y, dtype=f32)
    ti = (round_half_away(theta / f32(ang_res)) + f32(nb // 2)).astype(np.int64)  # trunc toward zero
    ri = round_half_away(r / f32(res)).astype(np.int64)
    c_raw = cls.astype(np.int64)  # float -> int truncation
    ok = (ti >= 0) & (ti < nb) & (ri >= 0) & (ri < nr) & (c_raw >= 0) & (c_raw <= 255)
    ti, ri, c_raw = ti[ok], ri[ok], c_raw[ok]
    c = np.asarray(lut256, np.int64)[c_raw]
    ok = (c >= 0) & (c < ncls)
    img = np.zeros((ncls, nb * nr), np.int64)
    np.add.at(img, (c[ok], ti[ok] + nb * ri[ok]), 1)
    return img.astype(f32)


def raster_cart(pts_xyzc, res, lut256, ncls, rows, cols):
    pts = np.asarray(pts_xyzc, f32)
    x, y, cls = pts[:, 0], pts[:, 1], pts[:, 3]
    keep = ~((x == 0) & (y == 0))
    x, y, cls = x[keep], y[keep], cls[keep]
    xi = (round_half_away(x / f32(res)) + f32(cols // 2)).astype(np.int64)
    yi = (round_half_away(y / f32(res)) + f32(rows // 2)).astype(np.int64)
    c_raw = cls.astype(np.int64)
    ok = (xi >= 0) & (xi < cols) & (yi >= 0) & (yi < rows) & (c_raw >= 0) & (c_raw <= 255)
    xi, yi, c_raw = xi[ok], yi[ok], c_raw[ok]
    c = np.asarray(lut256, np.int64)[c_raw]
    ok = (c >= 0) & (c < ncls)
    img = np.zeros((ncls, rows * cols), np.int64)
    np.add.at(img, (c[ok], yi[ok] + rows * xi[ok]), 1)
    return img.astype(f32)


def polar_table(nb, nr, ang_res, resolution):
    """Returns (2, nb*nr) float32: row 0 = cos(theta_i)*r_j, row 1 = sin(theta_i)*r_j, k = i + nb*j.
    theta_i = (i-(nb-1)/2)*ang_res ; r_j = j * float(1/resolution)."""
    i = np.arange(nb)
    # LinSpaced(nb, -(nb-1)/2, (nb-1)/2) has step exactly 1 -> exact half-integers in float32
    th = ((i - (nb - 1) / 2.0).astype(f32) * f32(ang_res)).astype(f32)
    rj = (np.arange(nr).astype(f32) * f32(1.0 / float(f32(resolution)))).astype(f32)
    c, s = _libm1("cosf", th), _libm1("sinf", th)
    t0 = (c[:, None] * rj[None, :]).astype(f32)  # (nb, nr)
    t1 = (s[:, None] * rj[None, :]).astype(f32)
    # k = i + nb*j  -> Fortran order flatten
    return np.stack([t0.ravel(order="F"), t1.ravel(order="F")]).astype(f32)


def local_map_polar(class_maps, class_mask, resolution, tab, cx, cy, scale, res):
    """class_maps: (ncls, H, W) float32 [row=y, col=x]; class_mask: (H, W) uint8 (1 = unknown).
    Returns dists (ncls, P) float32 and mask (P,) uint8."""
    ncls, H, W = class_maps.shape
    p0 = (tab[0] * f32(scale)).astype(f32) * f32(res)
    p1 = (tab[1] * f32(scale)).astype(f32) * f32(res)
    p0 = (p0 + f32(cy) / f32(resolution)).astype(f32)
    p1 = (p1 + f32(cx) / f32(resolution)).astype(f32)
    ri = round_half_away(p0).astype(np.int64)
    ci = round_half_away(p1).astype(np.int64)
    inb = (ri >= 0) & (ri < H) & (ci >= 0) & (ci < W)
    rs, cs = np.where(inb, ri, 0), np.where(inb, ci, 0)
    dists = np.where(inb[None, :], class_maps[:, rs, cs], f32(0)).astype(f32)
    mask = np.where(inb, class_mask[rs, cs], 1).astype(np.uint8)
    return dists, mask


def rot_shift(rot, nb):
    v = float(f32(f32(rot) * f32(nb)) / f32(2)) / math.pi
    s = int(math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))  # std::round(double)
    return s % nb


def cost_for_rot(scan, window, maskf, nb, nr, class_weights, rot):
    """scan, window: (ncls, nb*nr); maskf: (nb*nr,).  Scan row a pairs with window row (a - s) mod nb."""
    P = nb * nr
    if f32(f32(maskf.astype(np.float64).sum()) / f32(P)) < 0.5:
        return f32(np.nan)
    s = rot_shift(rot, nb)
    a = np.arange(nb)
    src = (a - s) % nb
    cost = f32(0)
    norm = f32(0)
    top = a < s
    for c in range(scan.shape[0]):
        sc = scan[c].reshape(nr, nb)  # [j, a]
        wn = window[c].reshape(nr, nb)[:, src]
        mk = maskf.reshape(nr, nb)[:, src]
        prod = (sc * wn).astype(f32).astype(np.float64)
        nprod = (sc * mk).astype(f32).astype(np.float64)
        for blk in (top, ~top):
            cost = f32(float(cost) + float(f32(prod[:, blk].sum())) * 0.01 * float(f32(class_weights[c])))
        for blk in (top, ~top):
            norm = f32(norm + f32(nprod[:, blk].sum()))
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(cost / norm)


def compute_weights(class_maps, class_mask, resolution, tab, nb, nr, scan, res, fp, states):
    """states: dict of float32 arrays init_x, init_y, dx, dy, theta, scale + uint8 have_init (mutated like the
    reference).  fp: dict with regularization, force_on_map, fixed_scale, scale_log_min/max, class_weights.
    Returns float32 weights."""
    n = len(states["theta"])
    ncls, H, W = class_maps.shape
    width, height = f32(W) * f32(resolution), f32(H) * f32(resolution)
    w = np.zeros(n, f32)
    for p in range(n):
        sc = f32(states["scale"][p])
        cx = f32(f32(states["dx"][p]) * sc + f32(states["init_x"][p]))
        cy = f32(f32(states["dy"][p]) * sc + f32(states["init_y"][p]))
        if fp["force_on_map"] and (cx < 0 or cy < 0 or cx > width or cy > height):
            continue
        if fp["fixed_scale"] < 0 and (
            float(sc) < 10.0 ** float(f32(fp["scale_log_min"])) or float(sc) > 10.0 ** float(f32(fp["scale_log_max"]))
        ):
            continue
        dists, mask = local_map_polar(class_maps, class_mask, resolution, tab, cx, cy, sc, res)
        maskf = (f32(1) - mask.astype(f32)).astype(f32)
        if not states["have_init"][p]:
            best, best_t = np.finfo(f32).max, f32(0)
            t = f32(0)
            while float(t) < 2 * math.pi:
                c = cost_for_rot(scan, dists, maskf, nb, nr, fp["class_weights"], t)
                if c < best:
                    best, best_t = c, t
                t = f32(float(t) + 2 * math.pi / 40)
            states["theta"][p] = best_t
            states["have_init"][p] = 1
            cost = f32(best)
        else:
            cost = cost_for_rot(scan, dists, maskf, nb, nr, fp["class_weights"], states["theta"][p])
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            w[p] = f32(1.0 / float(f32(cost + f32(fp["regularization"]))))
    return w


def update_weights(raw, last_dist):
    """NaN policy, normalise, motion regularisation, renormalise, argmax.  Returns (weights float32, argmax)."""
    raw = np.asarray(raw, f32)
    n = len(raw)
    valid = ~np.isnan(raw)
    s = f32(0)
    for v in raw[valid]:
        s = f32(s + v)  # serial float32
    nv = int(valid.sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = f32(s / f32(nv))
        under = valid & (raw < mean)
        bs = f32(0)
        for v in raw[under]:
            bs = f32(float(bs) + float(f32(v - mean)) ** 2)
        nu = int(under.sum())
        bs = f32(np.sqrt(f32(bs / f32(nu)))) if nu > 0 else f32(np.nan)
    w = raw.copy()
    if s == 0 or nu < 1:
        w[:] = 1
    else:
        w[~valid] = f32(mean - bs)
    w = (w / f32(w.astype(np.float64).sum())).astype(f32)
    d = np.minimum((np.asarray(last_dist, f32) * f32(5)).astype(f32), f32(1))
    w = ((d * w).astype(f32) + ((f32(1) - d).astype(f32) / f32(n)).astype(f32)).astype(f32)
    w = (w / f32(w.astype(np.float64).sum())).astype(f32)
    return w, int(np.argmax(w))


def resample(w, n_new, shift):
    """First j whose serial float32 prefix exceeds (i+shift)/n_new; last index as fallback."""
    w = np.asarray(w, f32)
    prefix = np.cumsum(w, dtype=f32)  # sequential float32 accumulation
    runmax = np.maximum.accumulate(np.where(np.isnan(prefix), -np.inf, prefix)).astype(f32)
    samples = ((np.arange(n_new).astype(f32) + f32(shift)).astype(f32) / f32(n_new)).astype(f32)
    idx = np.searchsorted(runmax, samples, side="right")
    return np.minimum(idx, len(w) - 1).astype(np.int32)


def load_compressed_raster_map(label_img, flatten_lut, ncls, resolution=1.0):
    """TopDownMap::loadCompressedRasterMap + computeDists (src/top_down_map.cpp:116-144, 289-326) for a class-index
    image laid out like a cv::Mat (row 0 = top).  Returns (class_maps (ncls, rows, cols) f32 indexed [cls, y, x],
    class_mask (rows, cols) u8, 1 = unknown).  cv::distanceTransform(DIST_L2, DIST_MASK_PRECISE) is the exact
    Euclidean distance transform: restated with scipy's exact EDT."""
    from scipy.ndimage import distance_transform_edt

    label_img = np.asarray(label_img, np.uint8)
    img_h, img_w = label_img.shape
    res = f32(resolution)
    rows, cols = int(f32(img_h) / res), int(f32(img_w) / res)                      # :121-122
    yi, xi = np.arange(rows), np.arange(cols)
    iy = np.maximum((f32(img_h) - yi.astype(f32) * res - f32(1)).astype(f32).astype(np.int64), 0)   # :137
    ix = np.minimum((xi.astype(f32) * res).astype(f32).astype(np.int64), img_w - 1)                 # :138
    lab = label_img[iy[:, None], ix[None, :]].astype(np.int64)
    lut = np.asarray(flatten_lut, np.int64)
    cls = np.where(lab < len(lut), lut[np.minimum(lab, len(lut) - 1)], -1)
    cls = np.where((cls >= 0) & (cls < ncls), cls, -1)                                              # :139
    unknown = cls < 0                                                                               # :294-299
    maps = np.empty((ncls, rows, cols), f32)
    for c in range(ncls):
        binary = cls != c                                                                           # 0 inside the class
        if binary.all():
            d = np.full((rows, cols), np.inf)
        else:
            d = distance_transform_edt(binary)
        d = np.sqrt(np.round(d * d).astype(np.float64)).astype(f32)    # exact integer d^2 -> correctly rounded float sqrt
        d = (d * res).astype(f32)                                                                   # :314
        d = np.minimum(d, f32(50))                                                                  # :315
        d[unknown] = 0                                                                              # :317
        maps[c] = d
    return maps, unknown.astype(np.uint8)


def write_eig(path, array):
    """write_binary (include/top_down_render/top_down_map.h:29-39): Index rows, Index cols (2 x int64), then the
    column-major raw scalars.  `array` is indexed [row, col]."""
    a = np.asarray(array)
    with open(path, "wb") as f:
        np.asarray([a.shape[0], a.shape[1]], np.int64).tofile(f)
        np.asfortranarray(a).ravel(order="F").tofile(f)


def read_eig(path, dtype):
    """read_binary (top_down_map.h:41-50)."""
    with open(path, "rb") as f:
        rows, cols = np.fromfile(f, np.int64, 2)
        data = np.fromfile(f, dtype, int(rows) * int(cols))
    return data.reshape(int(cols), int(rows)).T.copy()


# ---- N3: the mixture behind the adaptive particle count (src/particle_filter.cpp:151-157, 245-318) ------------------
# The reference fits cv::ml::EM (random k-means start; OpenCV is not available here: parity unpinned).  The product
# fits a DETERMINISTIC mixture (top_down_renderer_amd/csrc/tdr_gmm.cpp); this is the same algorithm stated in NumPy:
# seeding = sample nearest the overall mean + farthest-first traversal, 10 Lloyd iterations, EM with full covariances
# + 1e-6 I, at most `max_iter` iterations, stop when the mean log-likelihood moves by < 1e-6.
def gmm_fit(samples, k, max_iter=100):
    X = np.asarray(samples, np.float64)
    m, D = X.shape
    reg = 1e-6
    mean = X.sum(0) / m
    idx = [int(np.argmin(((X - mean) ** 2).sum(1)))]
    mind = np.full(m, np.inf)
    while len(idx) < k:
        mind = np.minimum(mind, ((X - X[idx[-1]]) ** 2).sum(1))
        idx.append(int(np.argmax(mind)))
    cen = X[idx].copy()
    label = np.zeros(m, np.int64)
    for _ in range(10):
        d2 = ((X[:, None, :] - cen[None, :, :]) ** 2).sum(2)
        label = np.argmin(d2, 1)
        for c in range(k):
            if (label == c).any():
                cen[c] = X[label == c].sum(0) / (label == c).sum()
    w = np.full(k, 1.0 / k)
    mu = cen.copy()
    cov = np.tile(np.eye(D), (k, 1, 1))
    resp = np.zeros((m, k))
    resp[np.arange(m), label] = 1.0

    def m_step():
        for c in range(k):
            nk = resp[:, c].sum()
            if not nk > 1e-10:
                continue
            mc = (resp[:, c, None] * X).sum(0) / nk
            dx = X - mc
            cc = (resp[:, c, None, None] * dx[:, :, None] * dx[:, None, :]).sum(0) / nk + reg * np.eye(D)
            try:
                np.linalg.cholesky(cc)
            except np.linalg.LinAlgError:
                continue
            w[c], mu[c], cov[c] = nk / m, mc, cc

    m_step()
    prev = ll = -np.inf
    for _ in range(max(1, max_iter)):
        lp = np.empty((m, k))
        for c in range(k):
            L = np.linalg.cholesky(cov[c])
            y = np.linalg.solve(L, (X - mu[c]).T)
            lp[:, c] = np.log(w[c]) - 0.5 * (D * np.log(2 * np.pi) + 2 * np.log(np.diag(L)).sum() + (y * y).sum(0))
        mx = lp.max(1)
        lse = mx + np.log(np.exp(lp - mx[:, None]).sum(1))
        resp = np.exp(lp - lse[:, None])
        ll = lse.sum() / m
        if abs(ll - prev) < 1e-6:
            break
        prev = ll
        m_step()
    return w, mu, cov, ll


def gmm_select(samples, num_particles, num_gaussians, max_k=32):
    """computeGMM's cluster-count search (:259, 276-297) and cluster conversion (:303-312)."""
    X = np.asarray(samples, np.float64)
    m = len(X)
    k = max(1, min(num_particles // 20 + 1, num_gaussians))
    k = min(k, max_k, m)
    ll = gmm_fit(X, k)[3]
    direction = 0
    if k * 50 < num_particles and k + 1 <= min(max_k, m):
        if ll + 0.3 < gmm_fit(X, k + 1)[3]:
            direction = 1
    if k > 1:
        if ll - 0.3 < gmm_fit(X, k - 1)[3]:
            direction = -1
    k += direction
    w, mu, cov, ll = gmm_fit(X, k)
    means = np.stack([mu[:, 0], mu[:, 1], np.arctan2(mu[:, 3], mu[:, 2])], 1).astype(np.float32)
    covs = np.zeros((k, 3, 3), np.float32)
    covs[:, :2, :2] = cov[:, :2, :2]
    covs[:, 2, 2] = 1
    return k, means, covs
