"""Start / end time of every workgroup of one scoring launch (config 2): occupancy over time, dense vs scattered
workgroups, the drain at the end.  Needs a diagnostic build:
   python -c "from top_down_renderer_amd import build as b; b.OUT='/root/repo/top_down_renderer_amd/libtdr_hip_tl.so'; b.build(force=True, extra_flags=['-DTDR_SCORE_TIMELINE'])"
   TDR_LIB_PATH=.../libtdr_hip_tl.so PYTHONPATH=. python tools/diag_score_timeline.py      (GPU box)
Finding (round 1): ~1220 workgroups resident throughout (4.8 per CU), ~17 % of them scattered ones at any time; the
drain takes the last ~8 % of the launch (one scattered workgroup lasts 1.2 ms, a dense one 0.76 ms)."""
import ctypes as C, sys
import numpy as np, torch
import top_down_renderer_amd as pkg
from top_down_renderer_amd import synth
from top_down_renderer_amd.kernels import HipKernels
k = HipKernels()
cfg = synth.CONFIGS["c2"]; sc = synth.make_scene(cfg); n = cfg.n_particles
m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
r = pkg.ScanRendererPolar(sc.lut, kernels=k); r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res); scan = r.last_scan()[1]
fp = pkg.FilterParams(fixed_scale=1.0).to_c(cfg.ncls)
st, raw, perm = k.zeros((7, n)), k.zeros((n,)), k.zeros((n,), torch.int32)
k.states_to_device(sc.states, st, n); k.locality_order(st, n, m.rows, m.cols, perm)
for _ in range(3):
    k.score(m.dev, scan, cfg.res, fp, st, n, raw, perm=perm, uniform_scale=1.0)
k.synchronize()
gx = (n + 255) // 256; nch = 84
nb = gx * nch
buf = np.zeros(2 * nb, np.uint64)
k.lib.tdr_debug_read_timeline.argtypes = [C.c_void_p, C.c_int]
assert k.lib.tdr_debug_read_timeline(buf.ctypes.data_as(C.c_void_p), nb) == 0
t = buf.reshape(-1, 2).astype(np.int64)
ok = t[:, 0] > 0
nch = int(ok.sum()) // gx
print("chunks", nch)
t = t[: gx * nch]
t0 = t[:, 0].min(); s = (t[:, 0] - t0) / 100.0; e = (t[:, 1] - t0) / 100.0   # microseconds
dur = e - s
print("blocks", nb, "kernel span us", e.max())
# which groups are scattered: per group of 256 sorted particles, bbox
pp = perm.cpu().numpy(); cx = sc.states["init_x_px"][pp]; cy = sc.states["init_y_px"][pp]
grp_spread = np.array([max(np.ptp(cx[g * 256:(g + 1) * 256]), np.ptp(cy[g * 256:(g + 1) * 256])) for g in range(gx)])
scat = grp_spread > 64
print("scattered groups", int(scat.sum()), "of", gx)
blk_scat = np.tile(scat, nch)
print("mean dur us: dense %.1f scattered %.1f" % (dur[~blk_scat].mean(), dur[blk_scat].mean()))
print("last dense end %.0f us, last scattered end %.0f us" % (e[~blk_scat].max(), e[blk_scat].max()))
edges = np.linspace(0, e.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    act = (s <= mid) & (e > mid)
    started = (s >= a) & (s < b)
    ch = np.repeat(np.arange(nch), gx)
    print("t=%6.0f us active WGs %5d (scattered %5d dense %5d)  chunks starting: %s" % (mid, act.sum(), (act & blk_scat).sum(), (act & ~blk_scat).sum(), (ch[started].min(), ch[started].max()) if started.any() else "-"))
