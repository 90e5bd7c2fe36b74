"""What the span tuner of the mixed polar launch (tdr_su_span_begin, csrc/tdr_score_su.hip) settles on, against every fixed
span, for several particle distributions on config 2's map and scan.  usage: python tools/check_span_tuner.py (GPU box)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import top_down_renderer_amd as pkg
from top_down_renderer_amd import synth
from top_down_renderer_amd.kernels import HipKernels
k = HipKernels()
sc = synth.make_scene("c2")
cfg = sc.cfg
m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
r = pkg.ScanRendererPolar(sc.lut, kernels=k)
r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
rng = np.random.default_rng(5)
n = 100_000
dists = {"bench mix": sc.states,
         "8 clusters 40 px, one heading": synth.make_cluster_particles(cfg, sc.lab, rng, per_cluster=n // 8),
         "Gaussian 30 px, one heading": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0, sigma_deg=0.0),
         "Gaussian 5 px": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=0.0, sigma_px=5.0),
         "uniform": synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, uniform_frac=1.0)}
for name, st in dists.items():
    st = st.copy(); st["have_init"] = 1
    f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False, locality_every=1)
    f.set_states(st)
    perm = k.zeros((f.cap_local,), torch.int32)
    k.locality_order(f.st, len(st), m.rows, m.cols, perm)
    k.lib.tdr_config_shift_uniform_span(-2.0)
    # a different particle count per distribution restarts the tuner (shape key): vary n_total slightly instead
    res = {}
    for fixed in (None, 8.0, 16.0, 24.0, 40.0):
        if fixed is None:
            k.lib.tdr_config_shift_uniform_span(-2.0)
            # force a restart of the trial: one call with another shape
            k.score(m.dev, m.scan_handle(r.last_scan()), float(cfg.res), f.fp_c, f.st, len(st) - 64, f.raw_w, perm=None, uniform_scale=f._uniform_scale, n_total=len(st))
            for _ in range(9):
                k.score(m.dev, m.scan_handle(r.last_scan()), float(cfg.res), f.fp_c, f.st, len(st), f.raw_w, perm=perm, uniform_scale=f._uniform_scale, n_total=len(st))
        else:
            k.lib.tdr_config_shift_uniform_span(fixed)
            for _ in range(2):
                k.score(m.dev, m.scan_handle(r.last_scan()), float(cfg.res), f.fp_c, f.st, len(st), f.raw_w, perm=perm, uniform_scale=f._uniform_scale, n_total=len(st))
        k.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            k.score(m.dev, m.scan_handle(r.last_scan()), float(cfg.res), f.fp_c, f.st, len(st), f.raw_w, perm=perm, uniform_scale=f._uniform_scale, n_total=len(st))
        e1.record(); k.synchronize()
        res[fixed] = (e0.elapsed_time(e1) / 5, float(k.lib.tdr_config_shift_uniform_span(-1.0)))
    print(f"{name:32s} tuned: {res[None][0]:6.2f} ms (span {res[None][1]:4.0f}) | fixed 8: {res[8.0][0]:6.2f}  16: {res[16.0][0]:6.2f}  24: {res[24.0][0]:6.2f}  40: {res[40.0][0]:6.2f}")
k.lib.tdr_config_shift_uniform_span(-2.0)
