"""score_cart_su_kernel (csrc/tdr_score_cart.hip): the dense share of the Cartesian integer form with its sample loop in
generated assembly (tools/gen_cart_asm.py) and the known mask staged in LDS per wave — against the plain kernel
(score_cart_skip_kernel with integer accumulators: tdr_config_tuning("cart_seg_rows", 0)), the ray-mapped kernel and the
oracle.  Integer sums are exact: every path must give the SAME BITS (array_equal), whatever the segment length, the loop
variant a wave's box selects (general / no clamp / every cell known), or the fallback a box that does not fit takes.
Reference: getLocalMap + getCostForRot, src/top_down_map.cpp:429-459, src/state_particle.cpp:112-155 (the Cartesian score
is a DEFINITION of this repository, include/tdr.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ALL_RAY = 1e-6


@pytest.fixture(scope="module")
def tdr():
    import torch
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return pkg, HipKernels()


def _run(pkg, k, m, st, scan, res, seg_rows, span, mode=2):
    before = k.lib.tdr_config_shift_uniform(-1)
    seg0 = k.lib.tdr_config_tuning(b"cart_seg_rows", -1)
    try:
        k.lib.tdr_config_shift_uniform(mode)
        k.lib.tdr_config_shift_uniform_span(span)
        k.lib.tdr_config_tuning(b"cart_seg_rows", seg_rows)
        f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False, locality_every=1)
        f.set_states(st)
        f.update(np.ascontiguousarray(scan, np.float32), None, res)
        return f.raw_weights()
    finally:
        k.lib.tdr_config_shift_uniform(before)
        k.lib.tdr_config_shift_uniform_span(-2.0)
        k.lib.tdr_config_tuning(b"cart_seg_rows", seg0)


CASES = [
    # ncls, rows, cols, kind, particles
    (6, 48, 64, "scan", "mixed"),          # several column groups and segments, borders, every variant somewhere
    (6, 64, 40, "multi", "mixed"),         # many bins with several classes: the chunk lists
    (5, 32, 24, "scan", "mixed"),          # a record with an unused class slot
    (4, 40, 16, "scan", "mixed"),
    (6, 48, 64, "scan", "known"),          # a map without unknown cells around the cloud: the loop without mask lookups
    (6, 48, 64, "scan", "large"),          # scales that make a wave's box overflow its LDS area: the plain steps
    (6, 128, 72, "scan", "mixed"),         # a partial last column group (72 = 9 x 8 -> chunks of 8: whole; cpc may be 16: partial)
    (6, 36, 20, "scan", "mixed"),          # a partial column group behind whole ones
]


@pytest.mark.parametrize("ncls,rows,cols,kind,particles", CASES)
def test_generated_loop_equals_plain_kernel_ray_kernel_and_oracle(tdr, oracle, ncls, rows, cols, kind, particles):
    from top_down_renderer_amd import synth
    pkg, k = tdr
    cfg = synth.Config("cartsu", 8000, ncls, rows, cols, 600, 1400, polar=False, seed=300 + ncls + rows + cols, res=0.75)
    sc = synth.make_scene(cfg)
    rng = np.random.default_rng(21)
    st = synth.make_particles(cfg, sc.lab, sc.pose, rng, n=1400, sigma_px=8.0, sigma_deg=6.0, uniform_frac=0.1)
    st["scale"] = rng.uniform(0.9, 1.1, len(st)).astype(np.float32)
    class_maps, class_mask = sc.class_maps, sc.class_mask
    if particles == "mixed":
        st["init_x_px"][:6] = np.asarray([-50, 5, 600, 595, 300, 0.5], np.float32)          # off / at the border
        st["init_y_px"][:6] = np.asarray([300, 300, 300, 300, -40, 0.5], np.float32)
        # a second cloud hugging the map's corner: waves whose boxes reach the guard ring (the clamping variant)
        st["init_x_px"][700:1100] = rng.normal(6.0, 3.0, 400).astype(np.float32)
        st["init_y_px"][700:1100] = rng.normal(8.0, 3.0, 400).astype(np.float32)
    elif particles == "known":
        # every cell labelled: distance maps of a label image without holes (the mask is all zero)
        lab = sc.lab.copy()
        lab[lab < 0] = 0
        class_maps, class_mask = synth.label_to_maps(lab, ncls, cfg.map_resolution)
        st["init_x_px"] = rng.normal(300, 6.0, len(st)).astype(np.float32)
        st["init_y_px"] = rng.normal(300, 6.0, len(st)).astype(np.float32)
    elif particles == "large":
        st["scale"] = rng.uniform(6.0, 9.0, len(st)).astype(np.float32)
    scan = oracle.raster_cart(sc.pts, cfg.res, sc.lut, ncls, rows, cols)
    if kind == "multi":
        scan = (rng.random(scan.shape) < 0.3).astype(np.float32) * rng.integers(1, 4, scan.shape).astype(np.float32)
    om = oracle.OracleMap(class_maps, class_mask, 1.0)
    with np.errstate(all="ignore"):
        ref = oracle.compute_weights_cart(om, rows, cols, scan, cfg.res, oracle.make_params(ncls), st.copy())
    m = pkg.TopDownMap(pkg.Params(resolution=1.0), class_maps, class_mask, kernels=k)
    m.setWindow(rows, cols)
    plain = _run(pkg, k, m, st, scan, cfg.res, 0, 0.0)             # every particle dense, the plain kernel
    ray = _run(pkg, k, m, st, scan, cfg.res, 32, ALL_RAY)          # every particle through the ray-mapped kernel
    for seg in (32, 8, 4, 64):
        got = _run(pkg, k, m, st, scan, cfg.res, seg, 0.0)         # every particle dense, the generated loop
        assert np.array_equal(got, plain, equal_nan=True), f"segment of {seg} rows: {int((got != plain).sum())} weights differ"
    mixed = _run(pkg, k, m, st, scan, cfg.res, 32, 6.0)
    assert np.array_equal(plain, ray, equal_nan=True)
    assert np.array_equal(plain, mixed, equal_nan=True)
    assert np.array_equal(np.isnan(plain), np.isnan(ref))
    ok = ~np.isnan(ref)
    err = np.abs(plain[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-30)
    assert err.max(initial=0.0) <= 1e-5, err.max()
    assert ok.sum() > len(st) // 2


def test_device_selftest_of_the_scoring_kernels(tdr):
    """tdr_selftest_score (include/tdr.h): the generated loops against the plain kernels on the library's own fixed problem."""
    pkg, k = tdr
    rc = k.lib.tdr_selftest_score()
    assert rc == 0, k.lib.tdr_last_error().decode()
    # the process-wide switches it sets are back where they were
    assert k.lib.tdr_config_tuning(b"cart_seg_rows", -1) == 32
