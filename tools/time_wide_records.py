"""Times the scoring kernel on a config-2 map with ~3000 distinct distance values: dense records against the wide compact
form (csrc/tdr_cmap.hip).   PYTHONPATH=. python3 tools/time_wide_records.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import top_down_renderer_amd as pkg
from top_down_renderer_amd import synth
from top_down_renderer_amd.kernels import HipKernels
k = HipKernels()
cfg = synth.CONFIGS['c2']
sc = synth.make_scene(cfg, n_particles=100000)
yy, xx = np.mgrid[0:cfg.map_size, 0:cfg.map_size]
pert = (((xx + yy) % 4) / 256.0).astype(np.float32)
maps = sc.class_maps.copy()
maps += pert[None] * (maps > 0)
print("distinct", len(np.unique(maps)))
m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), maps, sc.class_mask, kernels=k)
print("cwords", m.dev.desc.cwords, "dict_n", m.dev.desc.dict_n)
m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
r = pkg.ScanRendererPolar(sc.lut, kernels=k); r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
r.renderSemanticTopDown(sc.pts, cfg.res, cfg.ang_res)
scan = r.last_scan()[1]
fp = pkg.FilterParams(fixed_scale=1.0).to_c(cfg.ncls)
n = 100000
st = k.zeros((7, n)); k.states_to_device(sc.states, st, n)
perm = k.zeros((n,), torch.int32)
k.locality_order(st, n, m.dev.rows, m.dev.cols, perm)
raw = k.zeros((n,))
res = {}
for name, on in (("dense", 0), ("wide", 1)):
    k.lib.tdr_config_compact(on)
    for _ in range(2): k.score(m.dev, scan, cfg.res, fp, st, n, raw, perm=perm, uniform_scale=1.0)
    k.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): k.score(m.dev, scan, cfg.res, fp, st, n, raw, perm=perm, uniform_scale=1.0)
    e1.record(); k.synchronize()
    res[name] = raw.cpu().numpy().copy()
    print(name, e0.elapsed_time(e1) / 5, "ms")
print("identical", np.array_equal(res["dense"], res["wide"], equal_nan=True))
