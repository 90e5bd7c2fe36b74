#!/usr/bin/env python3
"""Times the run-time map replacement of the reference node (TopDownRender::aerialMapCallback -> ParticleFilter::updateMap,
src/top_down_render.cpp:574-600, src/particle_filter.cpp:320-341) through the C ABI: tdr_filter_update_map_labels on a label
image of the given size (default 4000 x 4000, 6 classes) — ingest on the device (label -> classes -> exact distance
transforms), compact form + known mask, geometric layers, particle shift — and the raster-cache load of the same map.
    python3 tools/time_map_update.py [size]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from top_down_renderer_amd import synth  # noqa: E402
from top_down_renderer_amd._lib import FilterParamsC, check  # noqa: E402
from top_down_renderer_amd.kernels import HipKernels  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    ncls = 6
    k = HipKernels()
    L, vp = k.lib, C.c_void_p
    rng = np.random.default_rng(3)
    lab = synth.make_label_image(size, ncls, rng)
    img = np.where(lab < 0, 255, lab).astype(np.uint8)[::-1].copy()
    lut = np.full(256, -1, np.int32)
    lut[:ncls] = np.arange(ncls)
    m = vp()
    check(L.tdr_map_create(C.byref(m)))
    check(L.tdr_map_sample_pts_polar(m, 100, 25, C.c_float(2 * np.pi / 100)))
    fp = FilterParamsC()
    fp.pos_cov, fp.theta_cov, fp.regularization = 0.3, 0.03, 0.15
    fp.init_pos_px_x = fp.init_pos_px_y = fp.init_pos_px_cov = -1.0
    fp.init_pos_m_x = fp.init_pos_m_y = fp.init_pos_deg_theta = float("inf")
    fp.init_pos_deg_cov, fp.fixed_scale, fp.scale_log_min, fp.scale_log_max, fp.num_classes = 10.0, 1.0, -0.1, 1.0, ncls
    for c in range(ncls):
        fp.class_weights[c] = 1.0
    f = vp()
    check(L.tdr_map_set_labels(m, img.ctypes.data_as(vp), size, size, lut.ctypes.data_as(vp), 256, ncls, C.c_float(1.0), 0, 0))
    check(L.tdr_filter_create(m, 20000, C.byref(fp), 1, C.byref(f)))
    check(L.tdr_filter_initialize_particles(f))
    for rep in range(3):
        t0 = time.perf_counter()
        check(L.tdr_filter_update_map_labels(f, img.ctypes.data_as(vp), size, size, lut.ctypes.data_as(vp), 256, ncls,
                                             C.c_float(1.0), 10 * rep, 0))
        k.synchronize()
        print(f"tdr_filter_update_map_labels {size} x {size}, {ncls} classes: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    d = tempfile.mkdtemp(prefix="tdr_rasters_")
    t0 = time.perf_counter()
    check(L.tdr_map_save_rasters(m, d.encode()))
    print(f"tdr_map_save_rasters: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    m2 = vp()
    check(L.tdr_map_create(C.byref(m2)))
    t0 = time.perf_counter()
    check(L.tdr_map_load_rasters(m2, d.encode(), ncls, C.c_float(1.0), 0, 0))
    k.synchronize()
    print(f"tdr_map_load_rasters (PNG decode + ingest): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
