"""N2, second half: the node's per-scan control loop — TopDownRender::takeStep + publishPoseEst
(src/top_down_render.cpp:505-560, 331-365) — through the drop-in classes (include/top_down_render/top_down_render_core.h,
tests/cpp/facade_loop.cpp) against the oracle's restatement of the same loop (oracle.cpp: orc_publish_pose_est,
c_oracle.TakeStepLoop).

The point of the test: the node moves `current_range_scale_` on EVERY step (+0.05 while the position covariance is large,
-0.02 otherwise, :337-345), so one filter is rendered and scored with a different `res` scan after scan — every cached
product keyed on `res` (sample offsets, ray tables, the span tuner's state) must follow.  16 steps; the scenario passes
through the oscillation around range_scale_max, the freezeScale trigger (:356-359), the convergence gate (:362-364) and the
shrinking phase.  PARITY UNPINNED like the rest of the oracle (the reference holds no test of this logic).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "top_down_renderer_amd")
STEPS = 16
MOTION = (0.3, 0.1, 0.005)
RS_MIN, RS_MAX, TARGET = 0.5, 4.0, 6.5
SEED = 7


def _scenario(n=3072):
    """A cloud of sigma 9 px about the true pose whose scale is unknown (fixed_scale < 0): most particles near scale 1, 4 %
    at scale 2.5 scattered around it — while those survive the scale variance keeps freezeScale away; they die out within a
    few resamples."""
    from top_down_renderer_amd import synth
    sc = synth.make_scene("ref", n_particles=16)
    cfg = sc.cfg
    rng = np.random.default_rng(9)
    st = synth.make_particles(cfg, sc.lab, sc.pose, rng, n=n, sigma_px=9.0, sigma_deg=4.0, uniform_frac=0.0)
    st["scale"] = rng.normal(1.0, 0.015, n).astype(np.float32)
    no = int(0.04 * n)
    sel = rng.permutation(n)[:no]
    st["scale"][sel] = 2.5
    st["init_x_px"][sel] += rng.normal(0, 60, no).astype(np.float32)
    st["init_y_px"][sel] += rng.normal(0, 60, no).astype(np.float32)
    return sc, cfg, st


def _oracle_loop(oracle, sc, cfg, st, constructor_draws=False):
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, cfg.map_resolution)
    tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, cfg.map_resolution)
    fp = oracle.make_params(cfg.ncls, fixed_scale=-1.0)
    loop = oracle.TakeStepLoop(om, tab, cfg.nb, cfg.nr, cfg.ang_res, sc.lut, cfg.ncls, fp, st, seed=SEED,
                               range_scale_min=RS_MIN, range_scale_max=RS_MAX, target_uncertainty_m=TARGET)
    if constructor_draws:
        # ParticleFilter's constructor initialises a particle set from the shared generator when the map is there
        # (src/particle_filter.cpp:14-16, 19-84) — facade_loop.cpp then replaces the set, but the generator has moved on.
        # FilterParams as that program leaves them (include/top_down_render/state_particle.h defaults + its own settings)
        fpi = oracle.make_params(cfg.ncls, fixed_scale=-1.0, init_pos_deg_theta=-1.0, init_pos_deg_cov=-1.0)
        fpi.init_pos_m_x = fpi.init_pos_m_y = 1e9
        oracle.initialize_particles(om, fpi, len(st), loop.rng)
    return loop


def test_publish_pose_est_logic_by_hand(oracle):
    """orc_publish_pose_est against values worked out from the text of src/top_down_render.cpp:331-365."""
    L = oracle.lib()
    f32 = np.float32

    def call(ns, cov, scale, n, ml_scale, frozen):
        c = np.ascontiguousarray(cov, np.float32).reshape(16)
        return L.orc_publish_pose_est(C.byref(ns), oracle._p(c), C.c_float(scale), C.c_int(n), C.c_float(ml_scale),
                                      C.c_int(frozen))

    big = np.zeros((4, 4), np.float32)
    big[0, 0], big[1, 1], big[3, 3] = 100, 50, 1.0
    ns = oracle.NodeState(4.0, 0.5, 4.0, 2.5, 0)
    # cov large but the range scale is AT its maximum: the first condition fails on `< range_scale_max_`, the else-if
    # shrinks (:342-345) — float member stepped by a double constant
    assert call(ns, big, 1.0, 10, 1.0, 0) == 0
    assert f32(ns.current_range_scale) == f32(np.float64(f32(4.0)) - 0.02)
    prev = f32(ns.current_range_scale)
    call(ns, big, 1.0, 10, 1.0, 0)              # now below the maximum and the covariance is large: widen (:341)
    assert f32(ns.current_range_scale) == f32(np.float64(prev) + 0.05)
    # an unknown scale (-1): scale_2 = 1 (:336); 6.2 < 2.5^2 = 6.25 -> shrink
    small = np.zeros((4, 4), np.float32)
    small[0, 0], small[1, 1], small[3, 3] = 6.2, 3.0, 1.0
    prev = f32(ns.current_range_scale)
    call(ns, small, -1.0, 10, 1.0, 0)
    assert f32(ns.current_range_scale) == f32(np.float64(prev) - 0.02)
    # the same covariance over scale^2 = 0.25 is 24.8 > 6.25 -> widen (below the maximum again after the shrink)
    ns.current_range_scale = 3.0
    call(ns, small, 0.5, 10, 1.0, 0)
    assert f32(ns.current_range_scale) == f32(np.float64(f32(3.0)) + 0.05)
    # at the minimum and converged: nothing moves
    ns.current_range_scale = 0.5
    call(ns, small * 0, 1.0, 10, 1.0, 0)
    assert f32(ns.current_range_scale) == f32(0.5)
    # freeze trigger (:356): cov(3,3) < 0.003 * ml_state[3], only while not frozen, only with particles (:347)
    tight = np.zeros((4, 4), np.float32)
    tight[3, 3] = 0.0029
    assert call(ns, tight, -1.0, 10, 1.0, 0) == 1
    assert call(ns, tight, -1.0, 10, 1.0, 1) == 0
    assert call(ns, tight, -1.0, 0, 1.0, 0) == 0
    tight[3, 3] = 0.0031
    assert call(ns, tight, -1.0, 10, 1.0, 0) == 0
    assert call(ns, tight, -1.0, 10, 1.1, 0) == 1          # 0.0031 < 0.0033
    # convergence gate (:363): all four conditions, the scale read after a possible freeze
    gate = np.zeros((4, 4), np.float32)
    gate[0, 0], gate[1, 1], gate[2, 2] = 39.0, 39.0, 0.4

    def gate_call(cov, s2, scale_now):
        n2 = oracle.NodeState(4.0, 0.5, 4.0, 2.5, 0)
        c = np.ascontiguousarray(cov, np.float32).reshape(16)
        L.orc_publish_pose_est_gate(C.byref(n2), oracle._p(c), C.c_float(s2), C.c_float(scale_now))
        return n2.is_converged

    assert gate_call(gate, 1.0, 1.0) == 1
    assert gate_call(gate, 1.0, -1.0) == 0                 # scale unknown and not frozen
    g2 = gate.copy(); g2[2, 2] = 0.5
    assert gate_call(g2, 1.0, 1.0) == 0                    # cov(2,2) < 0.5 is strict
    g3 = gate.copy(); g3[1, 1] = 41.0
    assert gate_call(g3, 1.0, 1.0) == 0
    assert gate_call(g3, 4.0, 2.0) == 1                    # 41 / 4 < 40


def test_oracle_loop_passes_through_every_phase(oracle):
    """The scenario the GPU test replays, on the oracle alone: `res` differs on every step, the scale freezes after the
    first steps, the convergence gate opens later, the range scale then shrinks."""
    sc, cfg, st = _scenario()
    loop = _oracle_loop(oracle, sc, cfg, st, constructor_draws=True)
    res, froze, conv = [], [], []
    for k in range(STEPS):
        r = loop.step(sc.pts, *MOTION)
        res.append(r["res"])
        froze.append(r["froze"])
        conv.append(r["converged"])
    assert all(a != b for a, b in zip(res, res[1:]))
    assert sum(froze) == 1 and 2 <= froze.index(True) < STEPS - 4
    assert conv[-1] and not conv[froze.index(True)]        # converges later than it freezes
    assert res[-1] < res[-2] < res[-3] < res[-4]           # the shrinking phase
    assert max(res) > RS_MAX                               # ... after the oscillation around the maximum (4.03, 4.04)


@pytest.fixture(scope="module")
def loop_exe():
    from top_down_renderer_amd import build
    build.build()
    exe = os.path.join(tempfile.mkdtemp(prefix="tdr_facade_"), "facade_loop")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_loop.cpp"), "-o", exe, "-L", PKG, "-ltdr_hip",
                    f"-Wl,-rpath,{PKG}"], check=True)
    return exe


def write_inputs(d, sc, cfg, st, steps, device_scan, fixed_scale=-1.0, clouds=None):
    clouds = [sc.pts] if clouds is None else clouds
    open(os.path.join(d, "meta.txt"), "w").write(
        f"{cfg.ncls} {cfg.map_size} {cfg.map_size} {cfg.nb} {cfg.nr} {len(clouds[0])} {len(clouds)} {len(st)} {SEED} {steps} "
        f"{RS_MIN} {RS_MAX} {TARGET} {fixed_scale} {cfg.map_resolution} {device_scan}\n")
    np.ascontiguousarray(np.transpose(sc.class_maps, (0, 2, 1)), np.float32).tofile(os.path.join(d, "maps.bin"))
    np.ascontiguousarray(sc.class_mask.T, np.uint8).tofile(os.path.join(d, "mask.bin"))
    pcl = np.zeros((len(clouds), len(clouds[0]), 8), np.float32)
    for k, p in enumerate(clouds):
        pcl[k, :, :3] = p[:, :3]
        pcl[k, :, 4] = p[:, 3]
    pcl.tofile(os.path.join(d, "pts.bin"))
    st.tofile(os.path.join(d, "states.bin"))
    np.tile(np.asarray(MOTION, np.float32), (steps, 1)).tofile(os.path.join(d, "motion.bin"))


def test_loop_program_compiles_and_fails_loudly_without_gpu(loop_exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from top_down_renderer_amd import synth
    sc = synth.make_scene("micro")
    d = tempfile.mkdtemp()
    write_inputs(d, sc, sc.cfg, sc.states, 2, 0)
    r = subprocess.run([loop_exe, d], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("device_scan", [0, 1])
def test_take_step_loop_with_a_moving_range_scale_matches_oracle(loop_exe, oracle, device_scan):
    sc, cfg, st = _scenario()
    n = len(st)
    d = tempfile.mkdtemp(prefix="tdr_loop_")
    write_inputs(d, sc, cfg, st, STEPS, device_scan)
    r = subprocess.run([loop_exe, d], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(os.path.join(d, "out_raw.bin"), np.float32).reshape(STEPS, n)
    idx = np.fromfile(os.path.join(d, "out_idx.bin"), np.int32).reshape(STEPS, n)
    states = np.fromfile(os.path.join(d, "out_states.bin"), oracle.STATE_DTYPE).reshape(STEPS, n)
    est = np.fromfile(os.path.join(d, "out_est.bin"), np.float32).reshape(STEPS, 26)

    loop = _oracle_loop(oracle, sc, cfg, st, constructor_draws=True)
    fields = ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta", "scale")
    res_seen, froze_at, conv_at, mism_total = [], None, None, 0
    for k in range(STEPS):
        # the oracle steps from the SAME particle set (its resample follows the device's indices), with its own generator,
        # its own range scale and its own flags: nothing of the device's control state is handed over
        o = loop.step(sc.pts, *MOTION, force_idx=idx[k])
        assert np.float32(o["res"]) == est[k, 0], f"step {k}: rendered at res {est[k, 0]}, the oracle at {o['res']}"
        res_seen.append(float(est[k, 0]))
        # the particle set after the step: the propagated, scored states gathered by the device's indices, bit for bit
        # (a freeze replaces the scale by the geometric mean: 1e-6)
        want = loop.states
        for name in fields:
            if name == "scale" and o["froze"]:
                assert np.allclose(states[k][name], want[name], rtol=2e-6), f"step {k}: frozen scale"
            else:
                bad = states[k][name] != want[name]
                assert not bad.any(), f"step {k}: {name} differs on {int(bad.sum())} particles (propagate / gather)"
        # per-step raw weights (the scoring at THIS step's res), NaN pattern, 1e-5
        assert np.array_equal(np.isnan(raw[k]), np.isnan(o["raw"])), f"step {k}: NaN pattern"
        ok = ~np.isnan(o["raw"])
        # (gated particles weigh exactly 0 on both sides: state_particle.cpp:163-176)
        same = raw[k][ok] == o["raw"][ok]
        rel = np.where(same, 0.0, np.abs(raw[k][ok] - o["raw"][ok]) / np.maximum(np.abs(o["raw"][ok]), 1e-30))
        assert rel.max() < 1e-5, (f"step {k}: raw weights off by {rel.max():.2e} at res {o['res']} "
                                  f"({int((rel > 1e-5).sum())} of {len(rel)} particles above 1e-5)")
        # resample: the oracle's statistics + prefix on the DEVICE's raw weights give the device's indices
        w_o, _, _ = oracle.update_weights(raw[k], o["last"])
        idx_o = oracle.resample_prefix(w_o, n, o["shift"])
        mism = int((idx_o != idx[k]).sum())
        mism_total += mism
        assert mism <= 2 + n // 200, f"step {k}: {mism} resample indices differ"
        # publishPoseEst: range-scale trajectory, freeze step, convergence step — identical
        assert np.float32(o["range_scale"]) == est[k, 1], f"step {k}: range scale {est[k, 1]} vs {o['range_scale']}"
        assert bool(est[k, 2]) == o["froze"], f"step {k}: freeze trigger (cov33 {o['cov'][3, 3]:.6f}, scale {o['mean'][3]:.4f})"
        assert bool(est[k, 3]) == o["converged"], f"step {k}: convergence gate"
        assert bool(est[k, 5]) == o["scale_frozen"]
        assert np.allclose(est[k, 6:22].reshape(4, 4), o["cov"], rtol=2e-4, atol=2e-4), f"step {k}: covariance"
        assert np.allclose(est[k, 22:26], o["mean"], rtol=2e-5, atol=2e-5), f"step {k}: mean state"
        if o["froze"]:
            froze_at = k
            # after the freeze every particle carries the geometric mean (:343-357): copy it over so that both sides keep
            # stepping from identical bits (the device's pow / product order differs in the last ulp)
            loop.states["scale"] = states[k]["scale"]
        if o["converged"] and conv_at is None:
            conv_at = k
    assert all(a != b for a, b in zip(res_seen, res_seen[1:])), "res must differ on every step"
    assert froze_at is not None and froze_at >= 2 and conv_at is not None and conv_at > froze_at
    assert res_seen[-1] < res_seen[-2] < res_seen[-3]
    print(f"loop ok: res {res_seen[0]:.2f} .. {min(res_seen):.2f}/{max(res_seen):.2f}, froze at {froze_at}, converged at {conv_at}, "
          f"{mism_total} resample indices differed over {STEPS} steps")



# ---- the Python form of the same class (top_down_renderer_amd/top_down_render_core.py) --------------------------------
@pytest.fixture(scope="module")
def tdr():
    import torch
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return pkg, HipKernels()


class _StubFilter:
    """What publishPoseEst asks of a filter, with the answers given."""

    def __init__(self, cov, scale, n, ml_scale, frozen):
        self.cov, self.scale_, self.n, self.ml, self.frozen, self.froze_called = np.asarray(cov, np.float32), scale, n, ml_scale, frozen, 0

    def computeMeanCov(self):
        return self.cov

    def scale(self):
        return 1.0 if self.froze_called and self.scale_ < 0 else self.scale_   # (a freeze makes the scale known)

    def numParticles(self):
        return self.n

    def meanLikelihood(self):
        return np.asarray([0, 0, 0, self.ml], np.float32)

    def isScaleFrozen(self):
        return bool(self.frozen)

    def freezeScale(self):
        self.froze_called += 1


def test_python_core_publish_pose_est_equals_the_oracles(oracle):
    """TopDownRenderCore.publishPoseEst (Python) against orc_publish_pose_est / _gate over a sweep of covariances, scales and
    range scales: the same range-scale trajectory bit for bit, the same freeze and convergence decisions."""
    from top_down_renderer_amd.top_down_render_core import CoreConfig, TopDownRenderCore
    L = oracle.lib()
    rng = np.random.default_rng(11)
    core = TopDownRenderCore(CoreConfig(range_scale_min=RS_MIN, range_scale_max=RS_MAX, target_uncertainty_m=2.5))
    ns = oracle.NodeState(RS_MAX, RS_MIN, RS_MAX, 2.5, 0)
    for k in range(400):
        cov = np.zeros((4, 4), np.float32)
        cov[0, 0], cov[1, 1] = rng.choice([0.0, 3.0, 6.2, 6.3, 39.0, 41.0, 100.0], 2)
        cov[2, 2] = rng.choice([0.1, 0.4, 0.5, 0.6])
        cov[3, 3] = rng.choice([0.0029, 0.0031, 0.5])
        scale = float(rng.choice([-1.0, 0.5, 1.0, 2.0]))
        n = int(rng.choice([0, 10]))
        ml = float(rng.choice([1.0, 1.1]))
        frozen = int(rng.integers(0, 2))
        if k % 50 == 0:   # restart both sides somewhere else
            start = float(np.float32(rng.uniform(RS_MIN, RS_MAX)))
            core.current_range_scale_ = np.float32(start)
            core.is_converged_ = False
            ns = oracle.NodeState(start, RS_MIN, RS_MAX, 2.5, 0)
        stub = _StubFilter(cov, scale, n, ml, frozen)
        core.filter_ = stub
        e = core.publishPoseEst()
        c16 = np.ascontiguousarray(cov.reshape(16))
        freeze = L.orc_publish_pose_est(C.byref(ns), oracle._p(c16), C.c_float(scale), C.c_int(n), C.c_float(ml), C.c_int(frozen))
        if n >= 1:
            sc_now = 1.0 if (freeze and scale < 0) else scale
            L.orc_publish_pose_est_gate(C.byref(ns), oracle._p(c16), C.c_float(np.float32(scale) * np.float32(scale)), C.c_float(sc_now))
        assert np.float32(e.range_scale) == np.float32(ns.current_range_scale), k
        assert e.froze_scale == bool(freeze) and stub.froze_called == int(bool(freeze)), k
        assert e.converged == bool(ns.is_converged), k
        assert (e.ml_state is None) == (n < 1)


@pytest.mark.gpu
def test_python_core_take_step_loop_matches_oracle(tdr, oracle):
    """The Python TopDownRenderCore over the HIP path, 12 steps with `res` moving on every step, against the oracle's loop
    stepping from the same particle set (teacher-forced resample indices, like the C++ test above)."""
    from top_down_renderer_amd.top_down_render_core import CoreConfig, TopDownRenderCore
    pkg, k = tdr
    sc, cfg, st = _scenario()
    n, steps = len(st), 12
    m = pkg.TopDownMapPolar(pkg.Params(resolution=cfg.map_resolution), sc.class_maps, sc.class_mask, kernels=k)
    core = TopDownRenderCore(CoreConfig(particle_count=n, range_scale_min=RS_MIN, range_scale_max=RS_MAX, target_uncertainty_m=TARGET,
                                        theta_bins=cfg.nb, range_bins=cfg.nr, seed=SEED), kernels=k)
    core.initialize(m, pkg.FilterParams(fixed_scale=-1.0), sc.lut, init_particles=False)
    core.filter_.set_states(st)
    loop = _oracle_loop(oracle, sc, cfg, st)
    fields = ("init_x_px", "init_y_px", "dx_m", "dy_m", "theta", "scale")
    res_seen, froze_at = [], None
    for s in range(steps):
        res_now = core.currentRangeScale()
        e = core.takeStep(sc.pts, MOTION[:2], MOTION[2])
        assert e is not None and core.lastRes() == res_now
        raw = core.filter_.raw_weights()[:n]
        idx = core.filter_.resample_indices()[:n]
        got = core.filter_.get_states()
        o = loop.step(sc.pts, *MOTION, force_idx=idx)
        assert np.float32(o["res"]) == np.float32(res_now), f"step {s}"
        res_seen.append(res_now)
        for name in fields:
            if name == "scale" and o["froze"]:
                assert np.allclose(got[name], loop.states[name], rtol=2e-6)
            else:
                assert np.array_equal(got[name], loop.states[name]), f"step {s}: {name}"
        assert np.array_equal(np.isnan(raw), np.isnan(o["raw"]))
        ok = ~np.isnan(o["raw"])
        same = raw[ok] == o["raw"][ok]
        rel = np.where(same, 0.0, np.abs(raw[ok] - o["raw"][ok]) / np.maximum(np.abs(o["raw"][ok]), 1e-30))
        assert rel.max() < 1e-5, f"step {s}: raw weights off by {rel.max():.2e}"
        assert np.float32(e.range_scale) == np.float32(o["range_scale"]), f"step {s}"
        assert e.froze_scale == o["froze"] and e.converged == o["converged"], f"step {s}"
        if o["froze"]:
            froze_at = s
            loop.states["scale"] = got["scale"]
    assert all(a != b for a, b in zip(res_seen, res_seen[1:]))
    assert froze_at is not None
