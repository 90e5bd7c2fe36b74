"""the Gaussian mixture behind the adaptive particle count (src/particle_filter.cpp:151-157, 245-318).
The reference uses cv::ml::EM (random start, OpenCV absent here): PARITY UNPINNED.  What is tested: the product's
deterministic fit (csrc/tdr_gmm.cpp, host code: runs without a GPU) against its NumPy restatement in the oracle, that it
recovers the parameters of synthetic mixtures, the cluster-count search, and the particle count of :151-157."""
import ctypes as C

import numpy as np
import pytest

from oracle import np_oracle as no
from top_down_renderer_amd import _lib


def _mixture(rng, centres, sig, n_each, sig_theta=0.1):
    xs = []
    for (cx, cy, th), s in zip(centres, sig):
        xy = rng.normal([cx, cy], s, (n_each, 2))
        t = rng.normal(th, sig_theta, n_each)
        xs.append(np.column_stack([xy, 50 * np.cos(t), 50 * np.sin(t)]))
    x = np.concatenate(xs)
    return x[rng.permutation(len(x))]


def _fit(lib, x, k, max_iter=100):
    x = np.ascontiguousarray(x, np.float64)
    w, mu, cov, ll = np.zeros(k), np.zeros((k, 4)), np.zeros((k, 4, 4)), C.c_double(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert lib.tdr_gmm_fit_host(p(x), len(x), k, max_iter, p(w), p(mu), p(cov), C.byref(ll)) == 0
    return w, mu, cov, ll.value


def test_fit_matches_the_numpy_restatement_and_recovers_the_mixture():
    lib = _lib.load()
    rng = np.random.default_rng(5)
    centres = [(100, 200, 0.3), (400, 250, -2.0), (250, 600, 1.5)]
    x = _mixture(rng, centres, [8, 15, 5], 300)
    w, mu, cov, ll = _fit(lib, x, 3)
    w2, mu2, cov2, ll2 = no.gmm_fit(x, 3)
    assert np.allclose(w, w2, rtol=1e-8) and np.allclose(mu, mu2, rtol=1e-8, atol=1e-8)
    assert np.allclose(cov, cov2, rtol=1e-6, atol=1e-8) and abs(ll - ll2) < 1e-9
    # parameters of the generating mixture, any order
    order = [int(np.argmin(((mu[:, :2] - np.asarray(c[:2])) ** 2).sum(1))) for c in centres]
    assert sorted(order) == [0, 1, 2]
    for c, s, g in zip(centres, [8, 15, 5], order):
        assert np.allclose(mu[g, :2], c[:2], atol=3 * s / np.sqrt(300))
        assert abs(np.arctan2(mu[g, 3], mu[g, 2]) - c[2]) < 0.05
        assert np.allclose(np.sqrt(np.diag(cov[g])[:2]), s, rtol=0.2)
        assert abs(w[g] - 1 / 3) < 0.02
    # determinism: same input, same bits
    w3, mu3, cov3, ll3 = _fit(lib, x, 3)
    assert np.array_equal(mu, mu3) and np.array_equal(cov, cov3) and ll == ll3
    # more clusters never fit worse on well separated data; one cluster fits much worse
    assert _fit(lib, x, 1)[3] < ll - 1.0
    with pytest.raises(Exception):
        assert _fit(lib, x[:2], 3)   # k > m is refused


def test_cluster_count_search_and_adaptive_count():
    lib = _lib.load()
    rng = np.random.default_rng(6)
    # a heading spread of 1 mrad keeps each cluster Gaussian in (x, y, 50 cos, 50 sin): a wider arc is curved, and the
    # model (the reference's too) then gains likelihood from splitting it
    x = _mixture(rng, [(100, 200, 0.3), (400, 250, -2.0)], [10, 10], 400, sig_theta=0.001)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def select(k0, num_particles):
        k = C.c_int(k0)
        means, covs = np.zeros((32, 3), np.float32), np.zeros((32, 9), np.float32)
        assert lib.tdr_gmm_select_host(p(np.ascontiguousarray(x)), len(x), num_particles, C.byref(k), 32, p(means),
                                       p(covs)) == 0
        return k.value, means[: k.value], covs[: k.value].reshape(-1, 3, 3)

    k, means, covs = select(1, 20000)          # one cluster is clearly too few: the search moves up (:280-286)
    assert k == 2
    k2, means2, covs2 = no.gmm_select(x, 20000, 1)
    assert k2 == 2 and np.allclose(means, means2, atol=1e-4) and np.allclose(covs, covs2, rtol=1e-4, atol=1e-4)
    assert np.allclose(covs[:, 2, 2], 1) and np.allclose(covs[:, 2, :2], 0)
    k, _, _ = select(2, 20000)                 # already right: a third cluster gains < 0.3, a single one loses > 0.3
    assert k == 2 == no.gmm_select(x, 20000, 2)[0]
    k, _, _ = select(4, 20000)                 # too many: moves down one per call (:288-294)
    assert k == 3 == no.gmm_select(x, 20000, 4)[0]
    xa = _mixture(rng, [(100, 200, 0.3), (400, 250, -2.0)], [10, 10], 400, sig_theta=0.15)   # curved arcs
    ka = C.c_int(2)
    ma, ca = np.zeros((32, 3), np.float32), np.zeros((32, 9), np.float32)
    assert lib.tdr_gmm_select_host(p(np.ascontiguousarray(xa)), len(xa), 20000, C.byref(ka), 32, p(ma), p(ca)) == 0
    assert ka.value == no.gmm_select(xa, 20000, 2)[0]
    k, _, _ = select(1, 40)                    # k*50 >= num_particles: never tries more (:280)
    assert k == 1
    k, _, _ = select(5, 30)                    # min(n/20 + 1, k) (:259)
    assert k <= 2
    # :151-157: sum of sqrt(l0)*sqrt(l1) over the clusters, bounded below by 3/4 of the last count + 10 and above by max
    cov = np.zeros((2, 9), np.float32)
    cov[0, [0, 4]] = [100.0, 400.0]            # 10 * 20
    cov[1, [0, 1, 3, 4]] = [50.0, 30.0, 30.0, 50.0]   # eigenvalues 80, 20 -> int(8.94 * 4.47) = 39 or 40
    n = lib.tdr_adaptive_count_host(p(cov), 2, 100, 100000)
    assert n in (239, 240)
    assert lib.tdr_adaptive_count_host(p(cov), 2, 1000, 100000) == 760
    assert lib.tdr_adaptive_count_host(p(cov), 2, 100, 150) == 150
    from oracle import c_oracle
    assert n == c_oracle.adaptive_count(cov.reshape(2, 3, 3)[:, :2, :2], 100, 100000)


@pytest.mark.gpu
def test_filter_compute_gmm_on_device_particles(oracle):
    """computeGMM through the Python host: samples gathered on the device (:262-272), fit on the host; against the
    oracle's restatement on the same particle set.  Then the count feeds update() like :151-157."""
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd import synth
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    sc = synth.make_scene("c1", n_particles=6000)
    cfg = sc.cfg
    st = sc.states.copy()
    half = len(st) // 2
    st["init_x_px"][:half] += 200       # two clusters
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False)
    f.set_states(st)
    f.computeGMM()
    means, covs = f.getGMM()
    n = len(st)
    idx = np.minimum(n - 1, np.arange(1000) * n // 1000)
    s = st[idx]
    x0 = (s["dx_m"] * s["scale"] + s["init_x_px"]).astype(np.float32)
    y0 = (s["dy_m"] * s["scale"] + s["init_y_px"]).astype(np.float32)
    x = np.column_stack([x0, y0, np.float32(50) * np.cos(s["theta"]), np.float32(50) * np.sin(s["theta"])]).astype(np.float64)
    k2, means2, covs2 = no.gmm_select(x, n, 1)
    assert len(means) == k2 and np.allclose(means, means2, atol=1e-3) and np.allclose(covs, covs2, rtol=1e-3, atol=1e-3)
    assert f.num_gaussians_ == k2
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    f.update(scan, None, cfg.res, covs=covs)
    want = oracle.adaptive_count(covs[:, :2, :2], n, n)
    assert f.numParticles() == want
