"""ParticleFilter — host-side mirror of the reference class (include/top_down_render/particle_filter.h:24-41,
src/particle_filter.cpp) driving the HIP kernels.  Same method names and argument meaning as the reference:

    propagate(trans, omega)                    particle_filter.cpp:86-92
    update(top_down_scan, top_down_geo, res)   particle_filter.cpp:94-189
    computeMeanCov / meanLikelihood / maxLikelihood / computeCov   :191-236
    freezeScale / isScaleFrozen / scale / numParticles / updateMap  :320-371

Particles live on the device as a structure of arrays and never visit the host inside a step.  With a
torch.distributed process group the particle set is sharded contiguously by rank: every rank scores its
own shard, ONE all-gather exchanges the raw weights (+ last_dist) so that every rank computes bit-identical weight
statistics and the order-exact prefix over the global index order, each rank resamples its own slice of the outputs
and fetches the source states from one all-gather of the state planes.  The N-rank result equals the 1-rank result
bit for bit (same kernels on the same global arrays).

Deliberate, documented deviations from the reference:
  * explicit RNG seed instead of std::random_device (particle_filter.cpp:4-5);
  * the three `for (int i; ...)` loops (:110,120,138) start at 0;
  * the adaptive particle count (:151-157) takes the GMM covariances / target count as an explicit input instead of
    racing an OpenCV EM thread; default keeps N;
  * top_down_geo is accepted and ignored like in the reference, whose geometric score term is commented out
    (state_particle.cpp:145-152); use_geometric_cost=True switches that term on.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from ._lib import FilterParamsC
from .synth import STATE_DTYPE


@dataclass
class FilterParams:
    """Mirror of the reference's FilterParams (include/top_down_render/state_particle.h:19-38), defaults from
    src/top_down_render.cpp:197-237."""
    pos_cov: float = 0.3
    theta_cov: float = math.pi / 100
    regularization: float = 0.15
    init_pos_px_x: float = -1.0
    init_pos_px_y: float = -1.0
    init_pos_px_cov: float = -1.0
    init_pos_m_x: float = float("inf")
    init_pos_m_y: float = float("inf")
    init_pos_deg_theta: float = float("inf")
    init_pos_deg_cov: float = 10.0
    force_on_map: bool = False
    fixed_scale: float = -1.0
    scale_log_min: float = -0.1
    scale_log_max: float = 1.0
    class_weights: list = field(default_factory=list)

    def to_c(self, ncls):
        c = FilterParamsC()
        for name in ("pos_cov", "theta_cov", "regularization", "init_pos_px_x", "init_pos_px_y", "init_pos_px_cov",
                     "init_pos_m_x", "init_pos_m_y", "init_pos_deg_theta", "init_pos_deg_cov", "fixed_scale",
                     "scale_log_min", "scale_log_max"):
            setattr(c, name, float(getattr(self, name)))
        c.force_on_map = int(bool(self.force_on_map))
        c.num_classes = ncls
        cw = list(self.class_weights) if len(self.class_weights) else [1.0] * ncls
        if len(cw) != ncls:
            raise ValueError(f"class_weights has {len(cw)} entries, map has {ncls} classes")
        for i in range(16):
            c.class_weights[i] = float(cw[i]) if i < ncls else 0.0
        return c


class _Comm:
    """The collective steps of the sharded filter on top of torch.distributed (nccl == RCCL on ROCm, gloo on CPU)."""

    def __init__(self, group, force_collectives=False):
        self.group = group
        if group is None:
            self.world, self.rank = 1, 0
        else:
            import torch.distributed as dist
            self.dist = dist
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # `active`: the filter takes the sharded code path (collectives, global arrays).  A one-rank group normally
        # does not — nothing to exchange — unless force_collectives asks for it (the RCCL test on a one-GPU box).
        self.active = group is not None and (self.world > 1 or force_collectives)

    def all_gather(self, out, inp, async_op=False):
        """out: [world * len(inp)] contiguous, rank-major.  async_op: returns a handle whose wait() orders the current
        stream behind the collective (RCCL runs it on its own stream, so kernels launched meanwhile overlap it)."""
        if not self.active:
            out.copy_(inp)
            return None
        return self.dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)

    def broadcast(self, t, src=0):
        if self.active:
            self.dist.broadcast(t, src=src, group=self.group)


class ParticleFilter:
    def __init__(self, N, map, params, seed=0, group=None, kernels=None, parity_rng=True, locality_every=0,
                 init_particles=True, force_collectives=False, use_geometric_cost=False):
        """N: maximum (global) particle count; map: TopDownMapPolar; params: FilterParams.
        group: torch.distributed process group (None = single process); particles are sharded over its ranks.
        parity_rng: propagate consumes host-generated std::mt19937 normals in the reference's order (bit-parity with
        the CPU path); False = counter-based RNG on the device (throughput mode).
        locality_every: recompute the cache-locality processing order every k updates (0 = never).
        init_particles: run initializeParticles() like the reference's constructor (False: call set_states()).
        use_geometric_cost: let top_down_geo enter the score (getCostForRot's geometric block, state_particle.cpp:145-152,
        commented out in the reference; single-rank filters only).
        force_collectives: take the sharded code path (scan broadcast, both all-gathers) even when the group has one
        rank — lets a one-GPU box exercise RCCL end to end."""
        self.map_ = map
        self.k = kernels if kernels is not None else map.k
        self.params_ = params
        self.comm = _Comm(group, force_collectives)
        self.use_geometric_cost = bool(use_geometric_cost)
        if self.use_geometric_cost and self.comm.active:
            raise ValueError("the geometric cost term is not available on a sharded filter")
        self.max_num_particles_ = int(N)
        if self.max_num_particles_ % self.comm.world:
            raise ValueError("N must be a multiple of the number of ranks")
        self.cap_local = self.max_num_particles_ // self.comm.world
        self.num_particles_ = 0          # global count
        self.scale_frozen_ = False
        self.parity_rng = parity_rng
        self.locality_every = int(locality_every)
        self.seed = int(seed)
        self.score_ctx = self.k.score_ctx_create()   # this filter's span tuner and table factors (include/tdr.h)
        self.gen_ = self.k.rng_create(seed)
        # parity mode: the generator's stream continues on the device between the host's own draws, drawn ahead of the
        # step (a tdr_rng_pipe, csrc/tdr_rng.hip)
        self._pipe = None
        self._shift_dev = None
        self._last_shift = None
        self.step_ = 0
        self.prop_calls_ = 0      # device RNG counter: every propagate call draws fresh noise
        self._maybe_uninit = True
        self._uniform_scale = 0.0
        self._ml_fields = None
        self._ml_buf = None
        self.num_gaussians_ = 1   # :7
        self.gmm_means_ = np.zeros((0, 3), np.float32)
        self.gmm_covs_ = np.zeros((0, 3, 3), np.float32)
        self._alloc()
        if map.haveMap():
            self.fp_c = params.to_c(map.numClasses())
            if init_particles:
                self.initializeParticles()

    # ------------------------------------------------------------------------------------------------------------
    def _alloc(self):
        k, cap, N = self.k, self.cap_local, self.max_num_particles_
        self.st = k.zeros((7, cap))
        self.st_new = k.zeros((7, cap))
        self.last_dist = k.zeros((cap,))
        self.raw_w = k.zeros((cap,))
        self.idx = k.zeros((cap,), torch.int32)
        self.perm = None
        self.info = k.zeros((65536,))   # TDR_UW_INFO_FLOATS
        self.weights_ = k.zeros((N,))
        self.runmax = k.zeros((N,))
        if self.comm.active:
            self.xchg_in = k.zeros((2, cap))
            self.xchg_out = k.zeros((self.comm.world, 2, cap))
            self.raw_glob = k.zeros((N,))
            self.ld_glob = k.zeros((N,))
            self.st_all = k.zeros((self.comm.world * 7 * cap,))

    @property
    def n_local(self):
        return self.num_particles_ // self.comm.world

    # ---- particle set I/O (host <-> device; not on the per-step path) ---------------------------------------------
    def set_states(self, states):
        """states: structured array of reference `State`s (global, rank-major order)."""
        n = len(states)
        if n > self.max_num_particles_ or n % self.comm.world:
            raise ValueError("bad particle count for this filter")
        self.num_particles_ = n
        nl = self.n_local
        mine = np.ascontiguousarray(states[self.comm.rank * nl:(self.comm.rank + 1) * nl])
        self.k.states_to_device(mine, self.st, nl)
        self._maybe_uninit = bool((states["have_init"] == 0).any())
        self.scale_frozen_ = self.scale_frozen_ or self.params_.fixed_scale > 0
        # all scales equal AND frozen (propagate leaves them alone): the scoring kernel may hoist (tab*scale)*res
        sc = np.unique(states["scale"]) if n else np.zeros(0, np.float32)
        self._uniform_scale = float(sc[0]) if (self.scale_frozen_ and len(sc) == 1 and sc[0] > 0) else 0.0
        self.perm = None

    def get_states(self):
        """This rank's shard as a structured array of `State`s."""
        return self.k.states_to_host(self.st, self.n_local, STATE_DTYPE)

    def weights(self):
        return self.weights_[: self._n_weights].cpu().numpy().copy()

    def raw_weights(self):
        return self.raw_w[: self._n_raw].cpu().numpy().copy()

    # ---- particle_filter.cpp:19-84 ----------------------------------------------------------------------------------
    def initializeParticles(self):
        p = self.params_
        if p.fixed_scale >= 0:
            self.scale_frozen_ = True
        m = self.map_
        if self.scale_frozen_ and p.init_pos_m_x != float("inf"):
            cx, cy = m.mapCenter()
            p.init_pos_px_x = p.init_pos_m_x * p.fixed_scale + cx   # :29-30
            p.init_pos_px_y = p.init_pos_m_y * p.fixed_scale + cy
            W, H = m.size()
            if p.init_pos_px_x < 0 or p.init_pos_px_x >= W or p.init_pos_px_y < 0 or p.init_pos_px_y >= H:
                return  # "No map received for input loc" :32-36
            good = any(1 in m.getClassesAtPoint((int(p.init_pos_px_x + dx), int(p.init_pos_px_y + dy)))
                       for dx in range(-4, 5) for dy in range(-4, 5))
            if not good:
                return  # "No road in map at init location" :49-52
        self.fp_c = p.to_c(m.numClasses())
        self._rng_to_host()
        states = self.k.init_particles(self.gen_, m.maps_cm_host, m.numClasses(), m.rows, m.cols, m.resolution(),
                                       self.fp_c, self.max_num_particles_, STATE_DTYPE)
        n = (len(states) // self.comm.world) * self.comm.world
        n = min(n, self.max_num_particles_)
        self.set_states(states[:n])

    @property
    def last_shift_(self):
        """The uniform draw of the last update's resample (src/particle_filter.cpp:172-173); read back from the device when it
        was drawn there."""
        if self._last_shift is None and self._shift_dev is not None:
            return float(self.k.read_device_floats(self._shift_dev, 1)[0])
        return self._last_shift

    @property
    def _rng_on_device(self):
        return self._pipe is not None and self._pipe.on_device()

    def _rng_to_device(self):
        if self._pipe is None:
            self._pipe = self.k.rng_pipe_create(self.max_num_particles_)
        if not self._pipe.on_device():
            self._pipe.from_host(self.gen_)

    def _rng_to_host(self):
        """The host engine takes the stream back (synchronises)."""
        if self._rng_on_device:
            self._pipe.to_host(self.gen_)

    # ---- particle_filter.cpp:86-92 ------------------------------------------------------------------------------------
    def propagate(self, trans, omega):
        n, nl = self.num_particles_, self.n_local
        if n == 0:
            return
        p = self.params_
        z_dev = None
        if self.parity_rng and getattr(self.k, "device_rng", False):
            # the reference draws serially in global particle order from one shared generator: every rank continues that
            # stream on its device and keeps the normals of its own particles
            self._rng_to_device()
            z_dev = self._pipe.normals(n, self.comm.rank * nl, (self.comm.rank + 1) * nl, self.scale_frozen_)
        elif self.parity_rng:
            self._rng_to_host()
            z = self.k.propagate_normals(self.gen_, n, self.scale_frozen_)
            z_dev = self.k.to_device(z[self.comm.rank * nl:(self.comm.rank + 1) * nl])
        self.k.propagate(self.st, nl, self.last_dist, float(trans[0]), float(trans[1]), float(omega),
                         self.scale_frozen_, p.pos_cov, p.theta_cov, z4=z_dev, seed=self.seed, step=self.prop_calls_,
                         index_base=self.comm.rank * nl)
        self.prop_calls_ += 1

    # ---- particle_filter.cpp:94-189 -----------------------------------------------------------------------------------
    def update(self, top_down_scan, top_down_geo=None, res=1.0, n_target=None, covs=None, shift=None):
        """top_down_scan: list of per-class (nb x nr) column-major images (the reference's std::vector<ArrayXXf>),
        an (ncls, nb*nr) array, or a device scan handle from ScanRendererPolar (.last_scan()).
        n_target / covs: explicit input of the adaptive particle count (:151-157); default keeps the count.  A filter
        sharded over W ranks keeps the same number of particles on every rank: the new count is rounded DOWN to a
        multiple of W (at least W) — a W-rank run asked for 70 particles resamples 64 at W = 8 and equals, bit for bit,
        the one-rank run asked for 64 (tests/test_distributed.py).
        shift: override of the systematic-resampling offset (default: one draw from the shared generator, :172-173)."""
        if self.num_particles_ == 0:
            return  # :96-99
        k, m, comm = self.k, self.map_, self.comm
        n, nl = self.num_particles_, self.n_local
        scan_pk = m.scan_handle(top_down_scan)
        comm.broadcast(scan_pk, 0)  # the rasterised scan is produced once (rank 0) and broadcast (north star)

        if self.locality_every and (self.perm is None or self.step_ % self.locality_every == 0):
            if self.perm is None:
                self.perm = k.zeros((self.cap_local,), torch.int32)
            if getattr(m, "polar", True):
                k.locality_order(self.st, nl, m.rows, m.cols, self.perm)
            else:   # Cartesian windows rotate with the particle: heading belongs in the key
                wr, wc = m.window_shape()
                k.locality_order_pose(self.st, nl, m.rows, m.cols, self.perm, theta_radius=(wr + wc) / 16.0)
        if getattr(m, "polar", True) and self.use_geometric_cost and top_down_geo is not None:
            geo = np.stack([np.asarray(g, np.float32).reshape(m.nb, m.nr, order="A").ravel(order="F")
                            for g in top_down_geo[:2]])
            geo_pk = k.pack_scan(k.to_device(geo), 2, m.nb, m.nr)
            sums = (float(np.float32(geo[0].astype(np.float64).sum())), float(np.float32(geo[1].astype(np.float64).sum())))
            k.score_geo(m.dev, m.geo_dev(), scan_pk, geo_pk, sums, float(res), self.fp_c, self.st, nl, self.raw_w,
                        perm=self.perm if self.locality_every else None, init_search=self._maybe_uninit,
                        uniform_scale=self._uniform_scale)
        elif getattr(m, "polar", True):
            k.score(m.dev, scan_pk, float(res), self.fp_c, self.st, nl, self.raw_w,
                    perm=self.perm if self.locality_every else None, init_search=self._maybe_uninit,
                    uniform_scale=self._uniform_scale, n_total=n, ctx=self.score_ctx)
        else:   # Cartesian window (BASELINE config 4; definition in include/tdr.h:tdr_k_score_cart)
            rows, cols = m.window_shape()
            k.score_cart(m.dev, scan_pk, rows, cols, float(res), self.fp_c, self.st, nl, self.raw_w,
                         perm=self.perm if self.locality_every else None, n_total=n)
        # the init search initialises every un-gated particle; only gated ones (state_particle.cpp:163-176) can stay
        # un-initialised, and gates exist only with force_on_map or an unknown scale
        if not (self.params_.force_on_map or self.params_.fixed_scale < 0):
            self._maybe_uninit = False
        self._n_raw = nl

        if comm.active:
            self.xchg_in[0, :nl].copy_(self.raw_w[:nl])
            self.xchg_in[1, :nl].copy_(self.last_dist[:nl])
            xin = self.xchg_in[:, :nl].contiguous()
            xout = self.xchg_out.view(-1)[: comm.world * 2 * nl]
            comm.all_gather(xout, xin.view(-1))
            xo = xout.view(comm.world, 2, nl)
            self.raw_glob[:n].view(comm.world, nl).copy_(xo[:, 0, :])
            self.ld_glob[:n].view(comm.world, nl).copy_(xo[:, 1, :])
            raw_glob, ld_glob = self.raw_glob, self.ld_glob
        else:
            raw_glob, ld_glob = self.raw_w, self.last_dist
        # the pre-resample states are final now: their all-gather (28 B x N) runs behind the statistics and the running sum
        st_work, sa = None, None
        if comm.active:
            sa = self.st_all[: comm.world * 7 * nl]
            self._st_send = self.st[:, :nl].contiguous().view(-1)
            st_work = comm.all_gather(sa, self._st_send, async_op=True)
        k.update_weights(raw_glob, ld_glob, n, self.weights_, self.info)
        self._n_weights = n

        # adaptive particle count (:151-157), explicit inputs only
        n_new = n
        if covs is not None:
            acc = 0
            for c in covs:
                ev = np.linalg.eigvals(np.asarray(c, np.float32)[:2, :2]).real.astype(np.float32)
                acc += int(np.float32(np.sqrt(ev[0])) * np.float32(np.sqrt(ev[1])))
            n_new = min(max(acc, 3 * n // 4 + 10), self.max_num_particles_)
        elif n_target is not None:
            n_new = min(int(n_target), self.max_num_particles_)
        n_new = max(comm.world, (n_new // comm.world) * comm.world)

        k.prefix(self.weights_, n, self.runmax)
        nl_new = n_new // comm.world
        i0 = comm.rank * nl_new
        if shift is None and self._rng_on_device:
            # :172-173 (every rank owns an identically seeded generator): the stream is on the device, so is the draw
            self._shift_dev = self._pipe.uniform()
            k.resample_dev(self.runmax, n, n_new, self._shift_dev, i0, i0 + nl_new, self.idx)
        else:
            if shift is None:
                shift = k.rng_uniform(self.gen_)
            k.resample(self.runmax, n, n_new, float(shift), i0, i0 + nl_new, self.idx)
        if comm.active:
            if st_work is not None:
                st_work.wait()
            k.gather_states(sa, self.idx, nl_new, self.st_new, src_shard=nl)
            self._save_ml_state(sa, nl)
        else:
            k.gather_states(self.st, self.idx, nl_new, self.st_new)
            self._save_ml_state(None, nl)
        self.st, self.st_new = self.st_new, self.st   # :187
        self.num_particles_ = n_new
        self.perm = None if n_new != n else self.perm
        self.step_ += 1
        self._last_shift = None if shift is None else float(shift)

    def _save_ml_state(self, st_all, nl):
        """max_likelihood_particle_ (:145-147) points at the PRE-resample particle: copy its 7 fields into a small
        tensor of the filter's own (device-side gather, no host round trip).  A view into st_all / self.st would be
        overwritten by the next all-gather of the resampled states (meanLikelihood, computeMeanCov, computeCov,
        computeGMM and freezeScale all gather) or by the next update."""
        if self._ml_buf is None:
            self._ml_buf = self.k.zeros((12,))
        if st_all is not None:
            self.k.save_ml_state(self.info, st_all, self.num_particles_, self._ml_buf, src_shard=nl)
        else:
            self.k.save_ml_state(self.info, self.st, self.num_particles_, self._ml_buf)
        self._ml_fields = self._ml_buf[:7]   # (one small kernel; the buffer is rewritten by the next update only)

    def resample_indices(self):
        """Global source index of every particle of this rank's shard after the last update (for parity tests)."""
        return self.idx[: self.n_local].cpu().numpy().copy()

    def _argmax(self):
        return int(self.info[:1].cpu().view(torch.int32).item())

    # ---- particle_filter.cpp:191-236 ----------------------------------------------------------------------------------
    def _global_states(self):
        nl = self.n_local
        if not self.comm.active:
            return self.st, nl, 0
        sa = self.st_all[: self.comm.world * 7 * nl]
        self.comm.all_gather(sa, self.st[:, :nl].contiguous().view(-1))
        full = sa.view(self.comm.world, 7, nl).permute(1, 0, 2).contiguous().view(7, -1)
        return full, self.num_particles_, 0

    def meanLikelihood(self):
        st, n, _ = self._global_states()
        return self.k.mean_cov(st, n)[:4].cpu().numpy().copy()

    # ---- particle_filter.cpp:238-318: the mixture behind the adaptive particle count --------------------------------
    def computeGMM(self):
        """computeGMM (:252-318) on the current particles, synchronously (the reference runs it in a detached thread
        once per second).  cv::ml::EM is replaced by the deterministic fit of csrc/tdr_gmm.cpp (parity unpinned)."""
        st, n, _ = self._global_states()
        if n < 1:
            return
        num = min(1000, n)   # :262
        h = self.k.sample_ml_states(st, n, num).cpu().numpy()
        x = np.empty((num, 4), np.float64)
        x[:, 0], x[:, 1] = h[:, 0], h[:, 1]
        x[:, 2] = np.float32(50) * np.cos(h[:, 2])   # :269-270
        x[:, 3] = np.float32(50) * np.sin(h[:, 2])
        self.num_gaussians_, self.gmm_means_, self.gmm_covs_ = self.k.gmm_select(x, n, self.num_gaussians_)

    def getGMM(self):
        """(means [k][3] = x, y, theta; covs [k][3][3]) of the last computeGMM (:238-243)."""
        return self.gmm_means_.copy(), self.gmm_covs_.copy()

    def computeMeanCov(self):
        if self.num_particles_ < 1:
            return np.zeros((4, 4), np.float32)
        st, n, _ = self._global_states()
        return self.k.mean_cov(st, n)[4:20].cpu().numpy().reshape(4, 4).copy()

    def maxLikelihood(self):
        if self._ml_fields is None:
            raise RuntimeError("maxLikelihood: no update yet, there is no max-likelihood particle")
        s = self._ml_fields.cpu().numpy()
        sc = np.float32(s[5])
        return np.asarray([np.float32(s[2] * sc + s[0]), np.float32(s[3] * sc + s[1]), s[4], sc], np.float32)

    def computeCov(self):
        """particle_filter.cpp:226-236: covariance about the max-likelihood particle."""
        st, n, _ = self._global_states()
        ref = self.k.to_device(self.maxLikelihood())
        return self.k.mean_cov(st, n, about=ref)[4:20].cpu().numpy().reshape(4, 4).copy()

    def freezeScale(self):
        if not self.scale_frozen_:
            st, n, _ = self._global_states()
            out = self.k.mean_cov(st, n)
            self.k.set_scale(self.st, self.n_local, out[20:21])
            self.scale_frozen_ = True
            self._uniform_scale = float(out[20].item())

    def isScaleFrozen(self):
        return self.scale_frozen_

    def scale(self):
        if self.params_.fixed_scale > 0:
            return self.params_.fixed_scale
        if self.scale_frozen_:
            return float(self.st[5, 0].item())
        return -1.0

    def numParticles(self):
        return self.num_particles_

    def updateMap(self, class_maps, class_mask=None, map_center=(0, 0)):
        """particle_filter.cpp:320-341.  updateMap(label_img, map_center) like the reference (class-index image, the
        distance transform runs on the GPU), or updateMap(class_maps, class_mask, map_center) with ready distance maps."""
        old = self.map_.mapCenter()
        if class_mask is None or np.ndim(class_maps) == 2:
            if class_mask is not None:
                map_center = class_mask
            self.map_.updateMap(class_maps, None, map_center)
        else:
            self.map_.updateMap(class_maps, class_mask, map_center)
        self.fp_c = self.params_.to_c(self.map_.numClasses())
        if self.num_particles_ > 0:
            self.k.shift_init(self.st, self.n_local, float(map_center[0] - old[0]), float(map_center[1] - old[1]))
        else:
            self.initializeParticles()
