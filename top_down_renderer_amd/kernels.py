"""HipKernels: the one product implementation of the kernel interface the host classes use — thin wrappers that
hand torch-owned device memory and the current HIP stream to the stateless launchers of libtdr_hip.so
(include/tdr.h).  PyTorch is plumbing here (device memory, streams); every computation is a hand-written HIP kernel.

There is deliberately no CPU implementation in the product: constructing HipKernels without the built extension or
without a GPU raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import FilterParamsC, MapDescC, check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class ScoreCtx:
    """Owner of a tdr_score_ctx (include/tdr.h): one per filter."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def span(self):
        return float(self.lib.tdr_score_ctx_span(self.handle))

    def trial_calls(self):
        return int(self.lib.tdr_score_ctx_trial_calls(self.handle))

    def __del__(self):
        try:
            if self.handle:
                self.lib.tdr_score_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DevPtr:
    """A raw device pointer the library handed out, passed on like a tensor."""

    def __init__(self, addr):
        self.addr = addr

    def data_ptr(self):
        return self.addr


class RngPipe:
    """Owner of a tdr_rng_pipe (include/tdr.h)."""

    def __init__(self, kernels, handle):
        self.k, self.lib, self.handle = kernels, kernels.lib, handle

    def on_device(self):
        return bool(self.lib.tdr_rng_pipe_on_device(self.handle))

    def from_host(self, rng):
        check(self.lib.tdr_rng_pipe_from_host(self.handle, rng, self.k.stream()))

    def to_host(self, rng):
        check(self.lib.tdr_rng_pipe_to_host(self.handle, rng, self.k.stream()))

    def normals(self, n, lo, hi, scale_freeze):
        z = C.c_void_p(0)
        check(self.lib.tdr_rng_pipe_normals(self.handle, n, lo, hi, int(scale_freeze), C.byref(z), self.k.stream()))
        return DevPtr(z.value)

    def uniform(self):
        u = C.c_void_p(0)
        check(self.lib.tdr_rng_pipe_uniform(self.handle, C.byref(u), self.k.stream()))
        return DevPtr(u.value)

    def __del__(self):
        try:
            if self.handle:
                self.lib.tdr_rng_pipe_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceMap:
    """Device-resident interleaved map (tdr_map_desc) + the polar sampling table."""

    def __init__(self, rec, ncls, rows, cols, resolution):
        self.rec, self.ncls, self.rows, self.cols, self.resolution = rec, ncls, rows, cols, float(resolution)
        self.rec_floats = _lib.load().tdr_rec_floats(ncls)
        self.desc = MapDescC(rec.data_ptr(), ncls, rows, cols, self.rec_floats, self.resolution)
        self.tab = None       # device (P,2) f32
        self.tab_host = None  # numpy (P,2) f32
        self.fac = None       # device (2 nb + nr,) f32: the table's factors (tdr_polar_factors_host)
        self.nb = self.nr = 0
        self.ang_res = 0.0
        self.crec = self.dict = None   # compact form of the records (tdr_k_compact_map), when the map has one
        self.rec16 = None              # scratch of the 40-rotation search (tdr_map_desc.rec16), allocated on first use
        self.use_rec16 = True          # False: the search splits the f32 records on the fly (A/B, tests)

    def init_scratch(self, kernels):
        """Gives the descriptor the scratch the matrix-core init search writes its pre-split f16 records to."""
        if not self.use_rec16:
            self.desc.rec16 = None
            return
        if self.rec16 is None:
            nbytes = int(kernels.lib.tdr_map_rec16_bytes(self.ncls, self.rows, self.cols))
            if nbytes == 0:
                return
            self.rec16 = kernels.empty((nbytes,), torch.uint8)
        self.desc.rec16 = self.rec16.data_ptr()

    def geo_map(self, kernels, constant_one=False):
        """geo_maps_[0..1] as a 2-class DeviceMap (tdr_k_geo_map_from_map): distance to the nearest cell without / with
        a geometric class, derived from the class records; constant_one: what the reference's updateMap path leaves."""
        key = bool(constant_one)
        if getattr(self, "_geo", None) is None or self._geo[0] != key:
            lib = kernels.lib
            rec = kernels.empty((int(lib.tdr_map_rec_floats_total(2, self.rows, self.cols)),))
            ws = kernels.empty((int(lib.tdr_map_ingest_workspace_bytes(2, self.rows, self.cols)),), torch.uint8)
            check(lib.tdr_k_geo_map_from_map(C.byref(self.desc), int(key), _ptr(rec), _ptr(ws), kernels.stream()))
            kernels.synchronize()
            g = DeviceMap(rec, 2, self.rows, self.cols, self.resolution)
            g.tab, g.tab_host, g.nb, g.nr, g.ang_res = self.tab, self.tab_host, self.nb, self.nr, self.ang_res
            self._geo = (key, g)
        g = self._geo[1]
        g.tab, g.tab_host, g.nb, g.nr, g.ang_res = self.tab, self.tab_host, self.nb, self.nr, self.ang_res
        return g

    def compact(self, kernels):
        """Builds the compact records (csrc/tdr_cmap.hip): 10-bit dictionary indices instead of floats, exact by
        construction; read by scoring waves whose particles are spread over the map.  No-op for maps without one."""
        lib = kernels.lib
        nw = int(lib.tdr_cmap_words_total(self.ncls, self.rows, self.cols))
        if nw == 0:
            return False
        crec = kernels.empty((nw,), torch.int32)
        dic = kernels.empty((4096,))                    # TDR_CMAP_WIDE_MAX_DICT
        ws = kernels.empty((16384 * 4 + 16384 * 2 + 256,), torch.uint8)   # TDR_CMAP_WORKSPACE_BYTES
        check(lib.tdr_k_compact_map(C.byref(self.desc), _ptr(crec), _ptr(dic), _ptr(ws), kernels.stream()))
        if not self.desc.cwords and self.desc.dict_n < 0:
            # more than 1024 distinct distance values (a fine map resolution): the wide form, 16-bit fields
            crec = kernels.empty((int(lib.tdr_cmap_wide_words_total(self.ncls, self.rows, self.cols)),), torch.int32)
            check(lib.tdr_k_compact_map_wide(C.byref(self.desc), _ptr(crec), _ptr(dic), _ptr(ws), kernels.stream()))
        if self.desc.cwords:
            self.crec, self.dict = crec, dic            # keep the device memory alive with the descriptor
        else:
            self.desc.dict_n = 0
        return bool(self.desc.cwords)


class HipKernels:
    name = "hip"

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available() or self.lib.tdr_device_count() < 1:
            raise _lib.TdrError("no HIP device visible: the MI355X path cannot run (there is no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._ws = None
        self._pws = None
        self._rws = None

    # ---- plumbing -------------------------------------------------------------------------------------------
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def to_device(self, array):
        return torch.from_numpy(np.ascontiguousarray(array)).to(self.device)

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def read_device_floats(self, dev, n):
        """n floats behind a device pointer the library handed out (synchronises)."""
        if getattr(self, "_hip", None) is None:
            self._hip = C.CDLL("libamdhip64.so")
        self.synchronize()
        out = np.zeros(n, np.float32)
        rc = self._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(dev.data_ptr()), C.c_size_t(4 * n), 2)
        if rc != 0:
            raise _lib.TdrError(f"hipMemcpy device -> host failed ({rc})")
        return out

    # ---- map ------------------------------------------------------------------------------------------------
    def make_map(self, class_maps, class_mask, resolution):
        """class_maps (ncls,H,W) f32 indexed [cls,row,col]; class_mask (H,W) u8.  Uploaded in the reference's
        column-major per-class layout and interleaved on the device by tdr_k_pack_map."""
        ncls, H, W = class_maps.shape
        maps_cm = self.to_device(np.ascontiguousarray(np.transpose(class_maps, (0, 2, 1)), np.float32))
        mask_cm = self.to_device(np.ascontiguousarray(class_mask.T, np.uint8))
        rec = self.empty((int(self.lib.tdr_map_rec_floats_total(ncls, H, W)),))
        check(self.lib.tdr_k_pack_map(_ptr(maps_cm), _ptr(mask_cm), ncls, H, W, _ptr(rec), self.stream()))
        self.synchronize()
        del maps_cm, mask_cm
        m = DeviceMap(rec, ncls, H, W, resolution)
        m.compact(self)
        return m

    def make_map_from_labels(self, label_img, flatten_lut, ncls, resolution):
        """label_img: (img_h, img_w) uint8 class-index image (cv::Mat layout, row 0 = top).  Runs
        loadCompressedRasterMap + computeDists on the device (tdr_k_map_from_labels)."""
        label_img = np.ascontiguousarray(label_img, np.uint8)
        img_h, img_w = label_img.shape
        rows, cols = C.c_int(0), C.c_int(0)
        check(self.lib.tdr_map_ingest_shape(img_h, img_w, C.c_float(resolution), C.byref(rows), C.byref(cols)))
        rows, cols = rows.value, cols.value
        lut = np.ascontiguousarray(flatten_lut, np.int32).ravel()
        img_d, lut_d = self.to_device(label_img), self.to_device(lut)
        rec = self.empty((int(self.lib.tdr_map_rec_floats_total(ncls, rows, cols)),))
        ws = self.empty((int(self.lib.tdr_map_ingest_workspace_bytes(ncls, rows, cols)),), torch.uint8)
        check(self.lib.tdr_k_map_from_labels(_ptr(img_d), img_h, img_w, _ptr(lut_d), len(lut), ncls,
                                             C.c_float(resolution), _ptr(rec), _ptr(ws), self.stream()))
        self.synchronize()
        m = DeviceMap(rec, ncls, rows, cols, resolution)
        m.compact(self)
        return m

    def make_map_from_rasters(self, planes, resolution):
        """planes: (ncls, rows, cols) uint8, the class<i>.png images of a raster cache as stored (row 0 = top).  Runs
        loadRasterizedMaps' flip + computeDists on the device (tdr_k_map_from_rasters)."""
        planes = np.ascontiguousarray(planes, np.uint8)
        ncls, rows, cols = planes.shape
        pl_d = self.to_device(planes.reshape(-1))
        rec = self.empty((int(self.lib.tdr_map_rec_floats_total(ncls, rows, cols)),))
        ws = self.empty((int(self.lib.tdr_map_ingest_workspace_bytes(ncls, rows, cols)),), torch.uint8)
        check(self.lib.tdr_k_map_from_rasters(_ptr(pl_d), ncls, rows, cols, C.c_float(resolution), _ptr(rec), _ptr(ws),
                                              self.stream()))
        self.synchronize()
        m = DeviceMap(rec, ncls, rows, cols, resolution)
        m.compact(self)
        return m

    def png_read_gray8(self, path):
        w, h = C.c_int(0), C.c_int(0)
        rc = self.lib.tdr_png_read_gray8_host(str(path).encode(), None, 0, C.byref(w), C.byref(h))   # size query
        if w.value < 1 or h.value < 1:
            check(rc)
        out = np.empty((h.value, w.value), np.uint8)
        check(self.lib.tdr_png_read_gray8_host(str(path).encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(w), C.byref(h)))
        return out

    def png_write_gray8(self, path, img):
        img = np.ascontiguousarray(img, np.uint8)
        check(self.lib.tdr_png_write_gray8_host(str(path).encode(), img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0]))

    def unpack_map(self, m):
        """Device map -> the reference's host layout: class maps (ncls, cols, rows) i.e. column-major, mask (cols, rows)."""
        maps = self.empty((m.ncls, m.cols, m.rows))
        mask = self.empty((m.cols, m.rows), torch.uint8)
        check(self.lib.tdr_k_unpack_map(_ptr(m.rec), m.ncls, m.rows, m.cols, _ptr(maps), _ptr(mask), self.stream()))
        return maps.cpu().numpy(), mask.cpu().numpy()

    def set_polar_table(self, m, nb, nr, ang_res):
        tab = np.empty((nb * nr, 2), np.float32)
        check(self.lib.tdr_polar_table_host(nb, nr, C.c_float(ang_res), C.c_float(m.resolution),
                                            tab.ctypes.data_as(C.c_void_p)))
        m.tab_host, m.tab, m.nb, m.nr, m.ang_res = tab, self.to_device(tab), nb, nr, float(ang_res)
        fac = np.empty(2 * nb + nr, np.float32)
        check(self.lib.tdr_polar_factors_host(nb, nr, C.c_float(ang_res), C.c_float(m.resolution),
                                              fac.ctypes.data_as(C.c_void_p)))
        m.fac = self.to_device(fac)

    # ---- raster ---------------------------------------------------------------------------------------------
    def _raster_workspace(self, n):
        need = int(self.lib.tdr_raster_workspace_bytes(n))
        if self._rws is None or self._rws.numel() < need:
            self._rws = self.empty((max(need, 4),), torch.uint8)
        return self._rws

    def raster_polar(self, pts, n, stride, ioff, res, ang_res, lut, ncls, nb, nr, want_img=True):
        img = self.empty((ncls, nb * nr)) if want_img else None
        pk = self.empty((nr * nb * self.lib.tdr_rec_floats(ncls),))
        check(self.lib.tdr_k_raster_polar(_ptr(pts), stride, ioff, n, C.c_float(res), C.c_float(ang_res), _ptr(lut),
                                          ncls, nb, nr, _ptr(img), _ptr(pk), _ptr(self._raster_workspace(n)),
                                          self.stream()))
        return img, pk

    def raster_cart(self, pts, n, stride, ioff, res, lut, ncls, rows, cols, want_img=True):
        img = self.empty((ncls, rows * cols)) if want_img else None
        pk = self.empty((rows * cols * self.lib.tdr_rec_floats(ncls),))
        check(self.lib.tdr_k_raster_cart(_ptr(pts), stride, ioff, n, C.c_float(res), _ptr(lut), ncls, rows, cols,
                                         _ptr(img), _ptr(pk), _ptr(self._raster_workspace(n)), self.stream()))
        return img, pk

    def raster_geo(self, pts, stride, width, height, res, ang_res, rows, cols, polar):
        """renderGeometricTopDown on a device cloud (organised: element idy*width + idx): (2, rows*cols) device images."""
        img = self.empty((2, rows * cols))
        if polar:
            ws = self.empty((int(self.lib.tdr_raster_geo_workspace_bytes(max(1, width * height))),), torch.uint8)
            check(self.lib.tdr_k_raster_geo_polar(_ptr(pts), stride, width, height, C.c_float(res), C.c_float(ang_res),
                                                  rows, cols, _ptr(img), _ptr(ws), self.stream()))
        else:
            check(self.lib.tdr_k_raster_geo_cart(_ptr(pts), stride, width, height, C.c_float(res), rows, cols, _ptr(img),
                                                 self.stream()))
        return img

    # ---- getLocalMap materialised ----------------------------------------------------------------------------
    def local_map(self, m, polar, cx, cy, scale_or_rot, res, rows=None, cols=None):
        """(dists (ncls, rows*cols) float32, mask (rows*cols,) uint8) device tensors: the window of one pose
        (top_down_map_polar.cpp:21-53 / top_down_map.cpp:429-459)."""
        if polar:
            rows, cols = m.nb, m.nr
        d = self.empty((m.ncls, rows * cols))
        k = self.empty((rows * cols,), torch.uint8)
        if polar:
            check(self.lib.tdr_k_local_map_polar(C.byref(m.desc), _ptr(m.tab), rows, cols, C.c_float(cx), C.c_float(cy),
                                                 C.c_float(scale_or_rot), C.c_float(res), _ptr(d), _ptr(k), self.stream()))
        else:
            check(self.lib.tdr_k_local_map_cart(C.byref(m.desc), rows, cols, C.c_float(cx), C.c_float(cy),
                                                C.c_float(scale_or_rot), C.c_float(res), _ptr(d), _ptr(k), self.stream()))
        return d, k

    def pack_scan(self, img, ncls, nb, nr):
        pk = self.empty((nr * nb * self.lib.tdr_rec_floats(ncls),))
        check(self.lib.tdr_k_pack_scan(_ptr(img), ncls, nb, nr, _ptr(pk), self.stream()))
        return pk

    # ---- filter ---------------------------------------------------------------------------------------------
    def _workspace(self, ncls, nb, nr, n, n_total=0):
        need = int(self.lib.tdr_score_workspace_floats(ncls, nb, nr, n, n_total))
        if self._ws is None or self._ws.numel() < need:
            self._ws = self.empty((need,))
        return self._ws

    def score_ctx_create(self):
        """A tdr_score_ctx: the span tuner of ONE caller's scoring launches and the table's factors (include/tdr.h)."""
        h = C.c_void_p(0)
        check(self.lib.tdr_score_ctx_create(C.byref(h)))
        return ScoreCtx(self.lib, h)

    def score(self, m, scan_pk, res, fp, st, n, raw_w, perm=None, init_search=False, uniform_scale=0.0, n_total=0, ctx=None):
        """n_total: particle count of the whole (possibly sharded) filter, see tdr_k_score_polar; 0 = n.
        ctx: the caller's ScoreCtx (None: the kernels of a launch one after the other on the caller's stream)."""
        ws = self._workspace(m.ncls, m.nb, m.nr, n, n_total)
        cap = st.shape[1]
        if init_search and max(n, n_total) >= int(self.lib.tdr_config_rec16_min_particles(-1)):
            m.init_scratch(self)
        if ctx is not None:   # the table as its factors (include/tdr.h): the call checks them against m.tab itself
            check(self.lib.tdr_score_ctx_set_polar_factors(ctx.handle, _ptr(getattr(m, "fac", None)), m.nb, m.nr))
        check(self.lib.tdr_k_score_polar_ctx(C.byref(m.desc), _ptr(m.tab), _ptr(scan_pk), m.nb, m.nr, C.c_float(res),
                                             C.byref(fp), _ptr(st), cap, n, n_total, _ptr(perm), C.c_float(uniform_scale),
                                             int(bool(init_search)), _ptr(raw_w), _ptr(ws),
                                             ctx.handle if ctx is not None else C.c_void_p(0), self.stream()))

    def score_geo(self, m, gm, scan_pk, geo_pk, geo_sums, res, fp, st, n, raw_w, perm=None, init_search=False,
                  uniform_scale=0.0):
        """Scoring with the geometric term (tdr_k_score_polar_geo): gm = m.geo_map(...), geo_pk = pack_scan of the two
        geometric images, geo_sums = their sums."""
        need = int(self.lib.tdr_score_geo_workspace_floats(m.ncls, m.nb, m.nr, n, 0))
        if self._ws is None or self._ws.numel() < need:
            self._ws = self.empty((need,))
        check(self.lib.tdr_k_score_polar_geo(C.byref(m.desc), C.byref(gm.desc), _ptr(m.tab), _ptr(scan_pk), _ptr(geo_pk),
                                             C.c_float(geo_sums[0]), C.c_float(geo_sums[1]), m.nb, m.nr, C.c_float(res),
                                             C.byref(fp), _ptr(st), st.shape[1], n, 0, _ptr(perm),
                                             C.c_float(uniform_scale), int(bool(init_search)), _ptr(raw_w),
                                             _ptr(self._ws), self.stream()))

    def score_cart(self, m, scan_pk, rows, cols, res, fp, st, n, raw_w, perm=None, n_total=0):
        """n_total: particle count of the whole (possibly sharded) filter, see tdr_k_score_cart; 0 = n."""
        need = int(self.lib.tdr_score_cart_workspace_floats(m.ncls, rows, cols, n, n_total))
        if self._ws is None or self._ws.numel() < need:
            self._ws = self.empty((need,))
        check(self.lib.tdr_k_score_cart(C.byref(m.desc), _ptr(scan_pk), rows, cols, C.c_float(res), C.byref(fp),
                                        _ptr(st), st.shape[1], n, n_total, _ptr(perm), _ptr(raw_w), _ptr(self._ws),
                                        self.stream()))

    def propagate(self, st, n, last_dist, tx, ty, omega, scale_freeze, pos_cov, theta_cov, z4=None, seed=0, step=0,
                  index_base=0):
        check(self.lib.tdr_k_propagate(_ptr(st), st.shape[1], n, _ptr(last_dist), C.c_float(tx), C.c_float(ty),
                                       C.c_float(omega), int(scale_freeze), C.c_float(pos_cov), C.c_float(theta_cov),
                                       _ptr(z4), seed, step, index_base, self.stream()))

    def update_weights(self, raw_w, last_dist, n, w_out, info):
        check(self.lib.tdr_k_update_weights(_ptr(raw_w), _ptr(last_dist), n, _ptr(w_out), _ptr(info), self.stream()))

    def prefix_workspace(self, n):
        need = int(self.lib.tdr_prefix_workspace_bytes(n))
        if self._pws is None or self._pws.numel() < need:
            self._pws = self.empty((need,), torch.uint8)
        return self._pws

    def prefix(self, w, n, runmax):
        check(self.lib.tdr_k_prefix(_ptr(w), n, _ptr(runmax), _ptr(self.prefix_workspace(n)), self.stream()))

    def resample(self, runmax, n, n_new, shift, i_begin, i_end, idx):
        check(self.lib.tdr_k_resample(_ptr(runmax), n, n_new, C.c_float(shift), i_begin, i_end, _ptr(idx),
                                      self.stream()))

    def gather_states(self, src, idx, n_new, dst, src_shard=0):
        src_cap = src.shape[1] if src.dim() == 2 else 0
        check(self.lib.tdr_k_gather_states(_ptr(src), src_cap, src_shard, _ptr(idx), n_new, _ptr(dst), dst.shape[1],
                                           self.stream()))

    def save_ml_state(self, info, st, n, out12, src_shard=0):
        """out12[:7] = the SoA fields of particle argmax (info[0]), out12[8:12] its mlState; st: the [7][cap] planes, or
        with src_shard > 0 the all-gathered [rank][7][src_shard] buffer (particle_filter.cpp:145-147)."""
        cap = st.shape[1] if (st.dim() == 2 and src_shard == 0) else 0
        check(self.lib.tdr_k_save_ml_state(_ptr(info), _ptr(st), cap, src_shard, n, _ptr(out12), self.stream()))

    def mean_cov(self, st, n, about=None):
        """about: optional device tensor of 4 floats (computeCov about that mlState); None = about the mean."""
        out = self.empty((4800,))   # TDR_MEAN_COV_FLOATS: 24 results + reduction scratch
        check(self.lib.tdr_k_mean_cov(_ptr(st), st.shape[1], n, _ptr(about), _ptr(out), self.stream()))
        return out

    def sample_ml_states(self, st, n, num):
        """(num, 3) device tensor: mlState().head<3>() of particle min(n-1, i*n/num) (particle_filter.cpp:262-266)."""
        out = self.empty((num, 3))
        check(self.lib.tdr_k_sample_ml_states(_ptr(st), st.shape[1], n, num, _ptr(out), self.stream()))
        return out

    def gmm_select(self, samples, num_particles, num_gaussians, max_k=32):
        """Host: the deterministic mixture fit + cluster-count search of tdr_gmm.cpp.  samples: (m, 4) float64.
        Returns (k, means (k, 3) float32, covs (k, 3, 3) float32)."""
        x = np.ascontiguousarray(samples, np.float64)
        k = C.c_int(int(num_gaussians))
        means = np.zeros((max_k, 3), np.float32)
        covs = np.zeros((max_k, 9), np.float32)
        check(self.lib.tdr_gmm_select_host(x.ctypes.data_as(C.c_void_p), len(x), int(num_particles), C.byref(k), max_k,
                                           means.ctypes.data_as(C.c_void_p), covs.ctypes.data_as(C.c_void_p)))
        return k.value, means[: k.value].copy(), covs[: k.value].reshape(-1, 3, 3).copy()

    def set_scale(self, st, n, scale_dev):
        check(self.lib.tdr_k_set_scale(_ptr(st), st.shape[1], n, _ptr(scale_dev), self.stream()))

    def shift_init(self, st, n, dx, dy):
        check(self.lib.tdr_k_shift_init(_ptr(st), st.shape[1], n, C.c_float(dx), C.c_float(dy), self.stream()))

    def locality_order(self, st, n, rows, cols, perm):
        need = int(self.lib.tdr_locality_tmp_ints(n, rows, cols))
        tmp = self.empty((need,), torch.int32)
        check(self.lib.tdr_k_locality_order(_ptr(st), st.shape[1], n, rows, cols, _ptr(perm), _ptr(tmp),
                                            self.stream()))

    def locality_order_pose(self, st, n, rows, cols, perm, theta_radius):
        """Order for windows that rotate with the particle (Cartesian scoring): Morton code of (x, y, theta)."""
        need = int(self.lib.tdr_locality_pose_tmp_ints(n))
        tmp = self.empty((need + 2,), torch.int32)
        check(self.lib.tdr_k_locality_order_pose(_ptr(st), st.shape[1], n, rows, cols, C.c_float(theta_radius),
                                                 _ptr(perm), _ptr(tmp), self.stream()))

    def states_to_device(self, states_aos, st, n):
        """states_aos: numpy structured array with the reference's 28-byte State layout."""
        raw = self.to_device(np.ascontiguousarray(states_aos).view(np.uint8).reshape(-1))
        check(self.lib.tdr_k_states_aos_to_soa(_ptr(raw), n, _ptr(st), st.shape[1], self.stream()))
        self.synchronize()

    def states_to_host(self, st, n, dtype):
        raw = self.empty((n * 28,), torch.uint8)
        check(self.lib.tdr_k_states_soa_to_aos(_ptr(st), st.shape[1], n, _ptr(raw), self.stream()))
        return raw.cpu().numpy().view(dtype).reshape(-1).copy()

    # ---- host RNG (the reference's shared std::mt19937) -------------------------------------------------------
    def rng_create(self, seed):
        return C.c_void_p(self.lib.tdr_rng_create(C.c_uint32(seed & 0xFFFFFFFF)))

    # the same generator continued on the device (csrc/tdr_rng.hip): state = 640 uint32 words (include/tdr.h)
    device_rng = True

    def rng_state_to_device(self, rng):
        words = np.zeros(640, np.uint32)
        check(self.lib.tdr_rng_get_state_host(rng, words.ctypes.data_as(C.c_void_p)))
        return self.to_device(words.view(np.int32))

    def rng_state_to_host(self, rng, state_dev):
        words = np.ascontiguousarray(state_dev.cpu().numpy()).view(np.uint32)
        check(self.lib.tdr_rng_set_state_host(rng, words.ctypes.data_as(C.c_void_p)))

    def rng_propagate_normals_dev(self, state_dev, n, lo, hi, scale_freeze, z4_out, n_max):
        need = int(self.lib.tdr_rng_dev_workspace_bytes(max(n, n_max)))
        if getattr(self, "_rws", None) is None or self._rws.numel() < need:
            self._rws = self.empty((need,), torch.uint8)
        check(self.lib.tdr_k_rng_propagate_normals(_ptr(state_dev), n, lo, hi, int(scale_freeze), _ptr(z4_out),
                                                   _ptr(self._rws), self.stream()))

    def rng_uniform_dev(self, state_dev, out_dev):
        check(self.lib.tdr_k_rng_uniform(_ptr(state_dev), _ptr(out_dev), self.stream()))

    def resample_dev(self, runmax, n, n_new, shift_dev, i_begin, i_end, idx):
        check(self.lib.tdr_k_resample_dev(_ptr(runmax), n, n_new, _ptr(shift_dev), i_begin, i_end, _ptr(idx), self.stream()))

    def rng_pipe_create(self, n_max):
        """A tdr_rng_pipe (include/tdr.h): one filter's generator on the device, drawing ahead of the step."""
        h = C.c_void_p(0)
        check(self.lib.tdr_rng_pipe_create(n_max, C.byref(h)))
        return RngPipe(self, h)

    def rng_uniform(self, rng):
        return float(self.lib.tdr_rng_uniform_host(rng))

    def init_particles(self, rng, maps_cm_host, ncls, rows, cols, resolution, fp, max_num, dtype):
        out = np.zeros(max_num + 16, dtype)
        n = C.c_int64(0)
        check(self.lib.tdr_init_particles_host(rng, maps_cm_host.ctypes.data_as(C.c_void_p), ncls, rows, cols,
                                               C.c_float(resolution), C.byref(fp), max_num,
                                               out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out[: min(n.value, max_num + 16)].copy()

    def propagate_normals(self, rng, n, scale_freeze):
        z = np.empty((n, 4), np.float32)
        check(self.lib.tdr_propagate_normals_host(rng, n, int(scale_freeze), z.ctypes.data_as(C.c_void_p)))
        return z
