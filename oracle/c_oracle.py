"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/oracle.cpp header).  `build()` compiles the restatement with g++ (no GPU needed).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

STATE_DTYPE = np.dtype(
    [("init_x_px", "<f4"), ("init_y_px", "<f4"), ("dx_m", "<f4"), ("dy_m", "<f4"), ("theta", "<f4"),
     ("scale", "<f4"), ("have_init", "u1"), ("pad", "u1", (3,))]
)
assert STATE_DTYPE.itemsize == 28


class FilterParams(C.Structure):
    _fields_ = [
        ("pos_cov", C.c_float), ("theta_cov", C.c_float), ("regularization", C.c_float),
        ("init_pos_px_x", C.c_float), ("init_pos_px_y", C.c_float), ("init_pos_px_cov", C.c_float),
        ("init_pos_m_x", C.c_float), ("init_pos_m_y", C.c_float),
        ("init_pos_deg_theta", C.c_float), ("init_pos_deg_cov", C.c_float),
        ("force_on_map", C.c_int32),
        ("fixed_scale", C.c_float), ("scale_log_min", C.c_float), ("scale_log_max", C.c_float),
        ("num_classes", C.c_int32),
        ("class_weights", C.c_float * 16),
    ]


def make_params(ncls, pos_cov=0.3, theta_cov=np.pi / 100, regularization=0.15, force_on_map=False, fixed_scale=1.0,
                scale_log_min=-0.1, scale_log_max=1.0, class_weights=None, init_pos_px_x=-1.0, init_pos_px_y=-1.0,
                init_pos_px_cov=-1.0, init_pos_deg_theta=float("inf"), init_pos_deg_cov=10.0):
    fp = FilterParams()
    fp.pos_cov, fp.theta_cov, fp.regularization = pos_cov, theta_cov, regularization
    fp.init_pos_px_x, fp.init_pos_px_y, fp.init_pos_px_cov = init_pos_px_x, init_pos_px_y, init_pos_px_cov
    fp.init_pos_m_x = fp.init_pos_m_y = float("inf")
    fp.init_pos_deg_theta, fp.init_pos_deg_cov = init_pos_deg_theta, init_pos_deg_cov
    fp.force_on_map = int(force_on_map)
    fp.fixed_scale, fp.scale_log_min, fp.scale_log_max = fixed_scale, scale_log_min, scale_log_max
    fp.num_classes = ncls
    cw = [1.0] * ncls if class_weights is None else list(class_weights)
    for i in range(16):
        fp.class_weights[i] = cw[i] if i < ncls else 0.0
    return fp


class Map(C.Structure):
    _fields_ = [("class_maps", C.c_void_p), ("class_mask", C.c_void_p), ("ncls", C.c_int32), ("rows", C.c_int32),
                ("cols", C.c_int32), ("resolution", C.c_float)]


def build(force=False):
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_rng_create.restype = C.c_void_p
        L.orc_rng_create.argtypes = [C.c_uint32]
        L.orc_rng_destroy.argtypes = [C.c_void_p]
        L.orc_rng_uniform.restype = C.c_float
        L.orc_rng_uniform.argtypes = [C.c_void_p]
        L.orc_cost_for_rot.restype = C.c_float
        L.orc_update_weights.restype = C.c_long
        L.orc_freeze_scale.restype = C.c_float
        L.orc_classes_at_point.restype = C.c_uint32
        L.orc_initialize_particles.restype = C.c_long
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleMap:
    """class_maps: (ncls, H, W) float32 indexed [cls, row(y), col(x)]; class_mask: (H, W) uint8 (1 = unknown).
    Stored col-major per class exactly like the reference's Eigen arrays."""

    def __init__(self, class_maps, class_mask, resolution=1.0):
        ncls, H, W = class_maps.shape
        self.ncls, self.rows, self.cols, self.resolution = ncls, H, W, float(resolution)
        # element (r,c) at r + H*c  ==  array[c, r] C-contiguous
        self.maps_cm = np.ascontiguousarray(np.transpose(class_maps, (0, 2, 1)), np.float32)
        self.mask_cm = np.ascontiguousarray(class_mask.T, np.uint8)
        self.c = Map(_p(self.maps_cm), _p(self.mask_cm), ncls, H, W, resolution)


def raster_polar(pts, res, ang_res, lut256, ncls, nb, nr, stride=None, ioff=None):
    pts = np.ascontiguousarray(pts, np.float32)
    stride = pts.shape[1] if stride is None else stride
    ioff = stride - 1 if ioff is None else ioff
    out = np.empty((ncls, nb * nr), np.float32)
    lut = np.ascontiguousarray(lut256, np.int32)
    lib().orc_raster_polar(_p(pts), C.c_int(stride), C.c_int(ioff), C.c_long(pts.shape[0]), C.c_float(res),
                           C.c_float(ang_res), _p(lut), C.c_int(ncls), C.c_int(nb), C.c_int(nr), _p(out))
    return out


def raster_cart(pts, res, lut256, ncls, rows, cols):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.empty((ncls, rows * cols), np.float32)
    lut = np.ascontiguousarray(lut256, np.int32)
    lib().orc_raster_cart(_p(pts), C.c_int(pts.shape[1]), C.c_int(pts.shape[1] - 1), C.c_long(pts.shape[0]),
                          C.c_float(res), _p(lut), C.c_int(ncls), C.c_int(rows), C.c_int(cols), _p(out))
    return out


def geo_maps_from_class_maps(class_maps, class_mask, resolution=1.0):
    """geo_maps_ as the static-map constructor builds them (src/top_down_map.cpp:48-58): getGeoRasterMap (:410-427) on the
    class presence, then computeDists.  Returns an OracleMap with two "classes": [0] distance to the nearest cell WITHOUT
    a geometric class (flattened class >= 3), [1] to the nearest cell WITH one; nothing is masked."""
    from scipy.ndimage import distance_transform_edt
    class_maps = np.asarray(class_maps, np.float32)
    known = np.asarray(class_mask) == 0
    geo = np.zeros(class_maps.shape[1:], bool)
    for c in range(3, class_maps.shape[0]):
        geo |= (class_maps[c] == 0) & known                        # :417-419
    out = np.empty((2,) + geo.shape, np.float32)
    for i, zero_where in enumerate((~geo, geo)):                   # layer i is 0 exactly on `zero_where`
        if not zero_where.any():
            d = np.full(geo.shape, 3.0e38, np.float32)
        else:
            d = distance_transform_edt(~zero_where).astype(np.float32)
        out[i] = np.minimum(d * np.float32(resolution), np.float32(50))
    return OracleMap(out, np.zeros(geo.shape, np.uint8), resolution)


def compute_weights_geo(om, geo_om, tab, nb, nr, scan, geo_scan, res, fp, states, nthreads=0):
    """computeWeight with getCostForRot's geometric block (state_particle.cpp:145-152) switched on."""
    scan = np.ascontiguousarray(scan, np.float32)
    geo_scan = np.ascontiguousarray(geo_scan, np.float32)
    tab = np.ascontiguousarray(tab, np.float32)
    w = np.empty(len(states), np.float32)
    lib().orc_compute_weights_geo(C.byref(om.c), C.byref(geo_om.c), _p(tab), C.c_int(nb), C.c_int(nr), _p(scan),
                                  _p(geo_scan), C.c_float(res), C.byref(fp), _p(states), C.c_long(len(states)), _p(w),
                                  C.c_int(nthreads if nthreads > 0 else lib().orc_max_threads()))
    return w


def raster_geo_polar(pts, width, height, res, ang_res, nb, nr):
    """renderGeometricTopDown (scan_renderer_polar.cpp:6-81): (2, nb*nr) ground / obstacle images.  pts is the organised
    cloud in PCL order (element idy*width + idx)."""
    pts = np.ascontiguousarray(pts, np.float32)
    assert pts.shape[0] == width * height
    out = np.empty((2, nb * nr), np.float32)
    lib().orc_raster_geo_polar(_p(pts), C.c_int(pts.shape[1]), C.c_long(width), C.c_long(height), C.c_float(res),
                               C.c_float(ang_res), C.c_int(nb), C.c_int(nr), _p(out))
    return out


def raster_geo_cart(pts, width, height, res, rows, cols):
    """renderGeometricTopDown (scan_renderer.cpp:7-53): (2, rows*cols) ground / obstacle images."""
    pts = np.ascontiguousarray(pts, np.float32)
    assert pts.shape[0] == width * height
    out = np.empty((2, rows * cols), np.float32)
    lib().orc_raster_geo_cart(_p(pts), C.c_int(pts.shape[1]), C.c_long(width), C.c_long(height), C.c_float(res),
                              C.c_int(rows), C.c_int(cols), _p(out))
    return out


def polar_table(nb, nr, ang_res, resolution=1.0):
    """Returns (P, 2) float32 (interleaved like Eigen::Array2Xf)."""
    tab = np.empty((nb * nr, 2), np.float32)
    lib().orc_polar_table(C.c_int(nb), C.c_int(nr), C.c_float(ang_res), C.c_float(resolution), _p(tab))
    return tab


def sample_pts(cx, cy, rot, cols, rows, res):
    pts = np.empty((rows * cols, 2), np.float32)
    lib().orc_sample_pts(C.c_float(cx), C.c_float(cy), C.c_float(rot), _p(pts), C.c_int(cols), C.c_int(rows),
                         C.c_float(res))
    return pts


def local_map_polar(m, tab, cx, cy, scale, res):
    P = tab.shape[0]
    d = np.empty((m.ncls, P), np.float32)
    k = np.empty(P, np.uint8)
    lib().orc_local_map_polar(C.byref(m.c), _p(tab), C.c_long(P), C.c_float(cx), C.c_float(cy), C.c_float(scale),
                              C.c_float(res), _p(d), _p(k))
    return d, k


def rot_shift(rot, num_bins):
    """src/state_particle.cpp:123-128."""
    return int(lib().orc_rot_shift(C.c_float(rot), C.c_int(num_bins)))


def active_best_rel_pos(m, tab, nb, nr, preds):
    """ActiveLocalizer::getBestRelPos (src/active_localizer.cpp:44-82): ((dist, theta), best_diff, diffs[4][17])."""
    preds = np.ascontiguousarray(preds, np.float32).reshape(-1, 3)
    out = np.zeros(2, np.float32)
    best = C.c_float(0)
    diffs = np.zeros((4, 17), np.float32)
    lib().orc_active_best_rel_pos(C.byref(m.c), _p(tab), C.c_int(nb), C.c_int(nr), _p(preds), C.c_int(len(preds)), _p(out),
                                  C.byref(best), _p(diffs))
    return out, float(best.value), diffs


def local_map_cart(m, cx, cy, rot, res, rows, cols):
    P = rows * cols
    d = np.empty((m.ncls, P), np.float32)
    k = np.empty(P, np.uint8)
    lib().orc_local_map_cart(C.byref(m.c), C.c_float(cx), C.c_float(cy), C.c_float(rot), C.c_float(res),
                             C.c_int(rows), C.c_int(cols), _p(d), _p(k))
    return d, k


def cost_for_rot(scan, window, maskf, nb, nr, class_weights, rot):
    cw = np.ascontiguousarray(class_weights, np.float32)
    return float(lib().orc_cost_for_rot(_p(scan), _p(window), _p(maskf), C.c_int(scan.shape[0]), C.c_int(nb),
                                        C.c_int(nr), _p(cw), C.c_float(rot)))


def compute_weights(m, tab, nb, nr, scan, res, fp, states, nthreads=0):
    """states: structured array (STATE_DTYPE), mutated in place like the reference.  Returns float32 weights."""
    assert states.dtype == STATE_DTYPE and states.flags.c_contiguous
    scan = np.ascontiguousarray(scan, np.float32)
    w = np.zeros(len(states), np.float32)
    if nthreads <= 0:
        nthreads = lib().orc_max_threads()
    lib().orc_compute_weights(C.byref(m.c), _p(tab), C.c_int(nb), C.c_int(nr), _p(scan), C.c_float(res),
                              C.byref(fp), _p(states), C.c_long(len(states)), _p(w), C.c_int(nthreads))
    return w


def compute_weights_cart(m, rows, cols, scan, res, fp, states, nthreads=0):
    scan = np.ascontiguousarray(scan, np.float32)
    w = np.zeros(len(states), np.float32)
    if nthreads <= 0:
        nthreads = lib().orc_max_threads()
    lib().orc_compute_weights_cart(C.byref(m.c), C.c_int(rows), C.c_int(cols), _p(scan), C.c_float(res), C.byref(fp),
                                   _p(states), C.c_long(len(states)), _p(w), C.c_int(nthreads))
    return w


class Rng:
    def __init__(self, seed):
        self.h = C.c_void_p(lib().orc_rng_create(C.c_uint32(seed)))

    def uniform(self):
        return float(lib().orc_rng_uniform(self.h))

    def __del__(self):
        try:
            lib().orc_rng_destroy(self.h)
        except Exception:
            pass


def propagate(states, tx, ty, omega, scale_freeze, fp, rng):
    last = np.zeros(len(states), np.float32)
    lib().orc_propagate(_p(states), _p(last), C.c_long(len(states)), C.c_float(tx), C.c_float(ty), C.c_float(omega),
                        C.c_int(int(scale_freeze)), C.byref(fp), rng.h)
    return last


def propagate_normals(n, scale_freeze, rng):
    z = np.zeros((n, 4), np.float32)
    lib().orc_propagate_normals(C.c_long(n), C.c_int(int(scale_freeze)), _p(z), rng.h)
    return z


def update_weights(raw, last_dist):
    raw = np.ascontiguousarray(raw, np.float32)
    last_dist = np.ascontiguousarray(last_dist, np.float32)
    w = np.empty_like(raw)
    stats = np.zeros(4, np.float32)
    best = lib().orc_update_weights(_p(raw), _p(last_dist), C.c_long(len(raw)), _p(w), _p(stats))
    return w, int(best), stats


def resample_literal(w, n_new, shift):
    w = np.ascontiguousarray(w, np.float32)
    idx = np.empty(n_new, np.int32)
    lib().orc_resample_literal(_p(w), C.c_long(len(w)), C.c_long(n_new), C.c_float(shift), _p(idx))
    return idx


def resample_prefix(w, n_new, shift):
    w = np.ascontiguousarray(w, np.float32)
    idx = np.empty(n_new, np.int32)
    lib().orc_resample_prefix(_p(w), C.c_long(len(w)), C.c_long(n_new), C.c_float(shift), _p(idx))
    return idx


def gather_states(states, idx):
    out = np.empty(len(idx), STATE_DTYPE)
    idx = np.ascontiguousarray(idx, np.int32)
    lib().orc_gather_states(_p(states), _p(idx), C.c_long(len(idx)), _p(out))
    return out


def mean_cov(states):
    mean = np.zeros(4, np.float32)
    cov = np.zeros(16, np.float32)
    lib().orc_mean_cov(_p(states), C.c_long(len(states)), _p(mean), _p(cov))
    return mean, cov.reshape(4, 4)


def cov_about(states, ref):
    ref = np.ascontiguousarray(ref, np.float32)
    cov = np.zeros(16, np.float32)
    lib().orc_cov_about(_p(states), C.c_long(len(states)), _p(ref), _p(cov))
    return cov.reshape(4, 4)


def freeze_scale(states):
    return float(lib().orc_freeze_scale(_p(states), C.c_long(len(states))))


def shift_init(states, dx, dy):
    lib().orc_shift_init(_p(states), C.c_long(len(states)), C.c_int(dx), C.c_int(dy))


def classes_at_point(m, px, py):
    return int(lib().orc_classes_at_point(C.byref(m.c), C.c_int(px), C.c_int(py)))


def init_particle(m, fp, rng):
    """StateParticle's constructor draw (src/state_particle.cpp:3-49): one state from the shared generator."""
    out = np.zeros(1, STATE_DTYPE)
    lib().orc_init_particle(C.byref(m.c), C.byref(fp), rng.h, _p(out))
    return out


def initialize_particles(m, fp, max_num, rng):
    out = np.zeros(max_num + 16, STATE_DTYPE)
    n = lib().orc_initialize_particles(C.byref(m.c), C.byref(fp), C.c_int(max_num), rng.h, _p(out))
    return out[:n].copy()


def adaptive_count(covs2x2, last_num, max_num):
    c = np.ascontiguousarray(covs2x2, np.float32).reshape(-1, 4)
    return int(lib().orc_adaptive_count(_p(c), C.c_int(len(c)), C.c_int(last_num), C.c_int(max_num)))


def max_threads():
    return int(lib().orc_max_threads())


# ---- the unpinned overload choices as switches (oracle.cpp, "orc_set_overload_mode"); CPU sensitivity study only ----
def set_overload_mode(mode):
    """bit 0: polar raster through the double atan2 / sqrt; bit 1: meanLikelihood through the double cos / sin / atan2."""
    lib().orc_set_overload_mode(C.c_int(int(mode)))


def get_overload_mode():
    return int(lib().orc_get_overload_mode())


# ---- N2: TopDownRender::publishPoseEst (src/top_down_render.cpp:331-365) + the takeStep loop around it --------------
class NodeState(C.Structure):
    _fields_ = [("current_range_scale", C.c_float), ("range_scale_min", C.c_float), ("range_scale_max", C.c_float),
                ("target_uncertainty_m", C.c_float), ("is_converged", C.c_int32)]


def filter_scale(fp, scale_frozen, states):
    lib().orc_filter_scale.restype = C.c_float
    return float(lib().orc_filter_scale(C.byref(fp), C.c_int(int(scale_frozen)), _p(states), C.c_long(len(states))))


class TakeStepLoop:
    """The node's per-scan sequence on the CPU: takeStep (src/top_down_render.cpp:505-560) = render at
    current_range_scale_, propagate, update (weights, statistics, resample), then publishPoseEst (:331-365) = range-scale
    stepping, freeze trigger, convergence gate.  Holds what the node holds; one `step()` per scan."""

    def __init__(self, om, tab_fn, cfg_nb, cfg_nr, ang_res, lut, ncls, fp, states, seed, range_scale_min=0.5,
                 range_scale_max=4.0, target_uncertainty_m=2.5):
        self.om, self.nb, self.nr, self.ang_res, self.lut, self.ncls, self.fp = om, cfg_nb, cfg_nr, ang_res, lut, ncls, fp
        self.tab = tab_fn
        self.states = states.copy()
        self.rng = Rng(seed)
        self.node = NodeState(range_scale_max, range_scale_min, range_scale_max, target_uncertainty_m, 0)   # :47
        self.scale_frozen = False

    def step(self, pts, tx, ty, omega, force_idx=None):
        """Returns a dict of the step's intermediate results (raw weights before the resample, ...).  force_idx: resample
        with these indices instead of the loop's own (a test that keeps two implementations on one particle set; the
        loop's own indices are still reported)."""
        res = float(self.node.current_range_scale)
        scan = raster_polar(pts, res, self.ang_res, self.lut, self.ncls, self.nb, self.nr)              # :539
        last = propagate(self.states, tx, ty, omega, self.scale_frozen, self.fp, self.rng)              # :423
        pre = self.states.copy()
        raw = compute_weights(self.om, self.tab, self.nb, self.nr, scan, res, self.fp, self.states)     # :425
        w, best, stats = update_weights(raw, last)
        shift = self.rng.uniform()
        idx = resample_prefix(w, len(self.states), shift)
        scored = self.states.copy()    # after computeWeight: theta / have_init of un-initialised particles are set
        self.states = gather_states(self.states, idx if force_idx is None else np.ascontiguousarray(force_idx, np.int32))
        out = dict(res=res, scan=scan, pre=pre, last=last, scored=scored, raw=raw, w=w, best=best, shift=shift, idx=idx)
        out.update(self.publish_pose_est())
        return out

    def publish_pose_est(self):
        mean, cov = mean_cov(self.states)                                                               # :333
        sc = filter_scale(self.fp, self.scale_frozen, self.states)                                      # :335
        covf = np.ascontiguousarray(cov.reshape(16), np.float32)
        froze = False
        freeze = lib().orc_publish_pose_est(C.byref(self.node), _p(covf), C.c_float(sc), C.c_int(len(self.states)),
                                            C.c_float(float(mean[3])), C.c_int(int(self.scale_frozen)))
        if len(self.states) >= 1:
            if freeze:
                freeze_scale(self.states)                                                               # :359
                self.scale_frozen = True
                froze = True
            sc_now = filter_scale(self.fp, self.scale_frozen, self.states)
            lib().orc_publish_pose_est_gate(C.byref(self.node), _p(covf), C.c_float(np.float32(sc) * np.float32(sc)),
                                            C.c_float(sc_now))
        return dict(mean=mean, cov=cov, range_scale=float(self.node.current_range_scale), froze=froze,
                    scale_frozen=self.scale_frozen, converged=bool(self.node.is_converged))
