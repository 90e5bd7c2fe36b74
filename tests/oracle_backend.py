"""Test double for the kernel interface (top_down_renderer_amd/kernels.py:HipKernels) backed by the CPU oracle.

TEST INFRASTRUCTURE ONLY.  It lets the host-side logic of the sharded ParticleFilter (particle partition, all-gather of
weights and states, per-rank resample slices) run on CPU tensors over the gloo backend, where no GPU exists.  The
product never constructs it: without a GPU, HipKernels() raises.
"""
import ctypes as C

import numpy as np
import torch

from oracle import c_oracle as oracle

F = ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta", "scale")


def soa_to_aos(st, n):
    a = np.zeros(n, oracle.STATE_DTYPE)
    for i, name in enumerate(F):
        a[name] = st[i, :n].numpy()
    a["have_init"] = (st[6, :n].numpy() != 0).astype(np.uint8)
    return a


def aos_to_soa(a, st):
    n = len(a)
    for i, name in enumerate(F):
        st[i, :n] = torch.from_numpy(np.ascontiguousarray(a[name]))
    st[6, :n] = torch.from_numpy(a["have_init"].astype(np.float32))


class OracleDeviceMap:
    def __init__(self, class_maps, class_mask, resolution):
        self.om = oracle.OracleMap(class_maps, class_mask, resolution)
        self.ncls, self.rows, self.cols = self.om.ncls, self.om.rows, self.om.cols
        self.resolution = float(resolution)
        self.nb = self.nr = 0
        self.tab = None


class OracleMapStub:
    """Quacks like top_down_renderer_amd.TopDownMapPolar for ParticleFilter."""

    def __init__(self, kernels, class_maps, class_mask, resolution=1.0):
        self.k = kernels
        self.dev = OracleDeviceMap(class_maps, class_mask, resolution)
        self.rows, self.cols = self.dev.rows, self.dev.cols
        self.maps_cm_host = self.dev.om.maps_cm
        self.center = (0, 0)

    def samplePtsPolar(self, shape, ang_res):
        self.dev.nb, self.dev.nr = int(shape[0]), int(shape[1])
        self.dev.tab = oracle.polar_table(self.dev.nb, self.dev.nr, ang_res, self.dev.resolution)

    def scan_handle(self, scan):
        if isinstance(scan, tuple):
            return scan[1]
        return torch.from_numpy(np.ascontiguousarray(scan, np.float32)).clone()

    def numClasses(self):
        return self.dev.ncls

    def size(self):
        return (self.cols, self.rows)

    def mapCenter(self):
        return self.center

    def resolution(self):
        return self.dev.resolution

    def haveMap(self):
        return True

    polar = True

    def window_shape(self):
        return (self.dev.nb, self.dev.nr)


class OracleKernels:
    name = "oracle-test-double"

    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = []

    def zeros(self, shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype)

    def empty(self, shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype)

    def to_device(self, array):
        return torch.from_numpy(np.ascontiguousarray(array)).clone()

    def synchronize(self):
        pass

    def states_to_device(self, states_aos, st, n):
        aos_to_soa(states_aos[:n], st)

    def states_to_host(self, st, n, dtype):
        return soa_to_aos(st, n)

    # host RNG: the oracle's std::mt19937
    def rng_create(self, seed):
        return oracle.Rng(seed & 0xFFFFFFFF)

    def rng_uniform(self, rng):
        return rng.uniform()

    def propagate_normals(self, rng, n, scale_freeze):
        return oracle.propagate_normals(n, bool(scale_freeze), rng)

    def score_ctx_create(self):
        return None   # (the device kernels' span tuner: nothing to keep here)

    def score(self, m, scan, res, fp, st, n, raw_w, perm=None, init_search=False, uniform_scale=0.0, n_total=0, ctx=None):
        self.calls.append(("score", n))
        a = soa_to_aos(st, n)
        fpo = oracle.FilterParams.from_buffer_copy(bytes(fp))
        if not init_search:
            a["have_init"] = 1
        w = oracle.compute_weights(m.om, m.tab, m.nb, m.nr, scan.numpy(), res, fpo, a, nthreads=2)
        if init_search:
            aos_to_soa(a, st)
        raw_w[:n] = torch.from_numpy(w)

    def propagate(self, st, n, last_dist, tx, ty, omega, scale_freeze, pos_cov, theta_cov, z4=None, seed=0, step=0,
                  index_base=0):
        assert z4 is not None, "the CPU test double only supports parity-mode normals"
        f32 = np.float32
        z = z4.numpy()
        th = st[4, :n].numpy().copy()
        c, s = np.cos(th.astype(np.float64)).astype(f32), np.sin(th.astype(np.float64)).astype(f32)
        gx = (c * f32(tx) + (-s) * f32(ty)).astype(f32)
        gy = (s * f32(tx) + c * f32(ty)).astype(f32)
        lx, ly = st[2, :n].numpy().copy(), st[3, :n].numpy().copy()
        dx, dy = (lx + gx).astype(f32), (ly + gy).astype(f32)
        dist = np.sqrt((gx * gx + gy * gy).astype(f32)).astype(f32)
        th = (th + ((z[:, 0] * (f32(theta_cov) * dist)).astype(f32) + f32(omega)).astype(f32)).astype(f32)
        dx = (dx + (z[:, 1] * (f32(pos_cov) * dist)).astype(f32)).astype(f32)
        dy = (dy + (z[:, 2] * (f32(pos_cov) * dist)).astype(f32)).astype(f32)
        if not scale_freeze:
            with np.errstate(divide="ignore"):
                sd = np.minimum(2.0 / dist.astype(np.float64), 0.02).astype(f32)
            st[5, :n] = torch.from_numpy((st[5, :n].numpy() * ((z[:, 3] * sd).astype(f32) + f32(1))).astype(f32))
        st[4, :n], st[2, :n], st[3, :n] = torch.from_numpy(th), torch.from_numpy(dx), torch.from_numpy(dy)
        mx, my = (lx - dx).astype(f32), (ly - dy).astype(f32)
        last_dist[:n] = torch.from_numpy(np.sqrt((mx * mx + my * my).astype(f32)).astype(f32))

    def update_weights(self, raw_w, last_dist, n, w_out, info):
        self.calls.append(("update_weights", n))
        w, best, stats = oracle.update_weights(raw_w[:n].numpy(), last_dist[:n].numpy())
        w_out[:n] = torch.from_numpy(w)
        info[0] = float(np.asarray([best], np.int32).view(np.float32)[0])
        info[1:5] = torch.from_numpy(stats)

    def prefix(self, w, n, runmax):
        self.calls.append(("prefix", n))
        pre = np.cumsum(w[:n].numpy(), dtype=np.float32)
        runmax[:n] = torch.from_numpy(np.maximum.accumulate(np.where(np.isnan(pre), -np.inf, pre)).astype(np.float32))

    def resample(self, runmax, n, n_new, shift, i_begin, i_end, idx):
        self.calls.append(("resample", i_begin, i_end))
        samples = ((np.arange(i_begin, i_end).astype(np.float32) + np.float32(shift)).astype(np.float32)
                   / np.float32(n_new)).astype(np.float32)
        j = np.searchsorted(runmax[:n].numpy(), samples, side="right")
        idx[: i_end - i_begin] = torch.from_numpy(np.minimum(j, n - 1).astype(np.int32))

    def gather_states(self, src, idx, n_new, dst, src_shard=0):
        j = idx[:n_new].long()
        if src_shard:
            full = src.view(-1, 7, src_shard).permute(1, 0, 2).reshape(7, -1)
            dst[:, :n_new] = full[:, j]
        else:
            dst[:, :n_new] = src[:, j]

    def save_ml_state(self, info, st, n, out12, src_shard=0):
        j = int(info[:1].view(torch.int32).item())
        if j < 0 or j >= n:
            j = 0
        if src_shard:
            full = st.view(-1, 7, src_shard).permute(1, 0, 2).reshape(7, -1)
            f = full[:, j]
        else:
            f = st[:, j]
        out12[:7] = f
        out12[7] = 0.0
        out12[8] = f[2] * f[5] + f[0]
        out12[9] = f[3] * f[5] + f[1]
        out12[10] = f[4]
        out12[11] = f[5]

    def mean_cov(self, st, n, about=None):
        a = soa_to_aos(st, n)
        mean, cov = oracle.mean_cov(a)
        if about is not None:
            cov = oracle.cov_about(a, about.numpy())
        out = torch.zeros(24)
        out[:4] = torch.from_numpy(mean)
        out[4:20] = torch.from_numpy(cov.reshape(-1))
        out[20] = float(np.exp(np.log(a["scale"].astype(np.float64)).mean()))
        return out

    def set_scale(self, st, n, scale_dev):
        st[5, :n] = scale_dev[0]

    def shift_init(self, st, n, dx, dy):
        st[0, :n] += dx
        st[1, :n] += dy

    def locality_order(self, st, n, rows, cols, perm):
        perm[:n] = torch.arange(n, dtype=torch.int32)

    def init_particles(self, rng, maps_cm_host, ncls, rows, cols, resolution, fp, max_num, dtype):
        raise NotImplementedError
