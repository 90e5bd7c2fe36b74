"""The reference's random stream on the device (csrc/tdr_rng.hip): std::mt19937 + libstdc++'s normal / uniform
distributions reproduced word for word — the normals of a propagate call, the words it consumes and the generator state
afterwards against the host's own engine (tdr_propagate_normals_host / tdr_rng_uniform_host, i.e. libstdc++ itself).
Run with `pytest -m gpu`; the state conversion is checked on the CPU."""
import ctypes as C

import numpy as np
import pytest


def _lib():
    from top_down_renderer_amd import _lib
    return _lib.load()


def _host_normals(L, rng, n, freeze):
    z = np.empty((n, 4), np.float32)
    assert L.tdr_propagate_normals_host(rng, n, int(freeze), z.ctypes.data_as(C.c_void_p)) == 0
    return z


def test_engine_state_round_trip_on_the_host():
    """get_state / set_state move a std::mt19937 through libstdc++'s own representation: an engine rebuilt from the words
    continues the original's stream, wherever in a block it stood."""
    L = _lib()
    for seed, burn in ((1, 0), (5489, 1), (77, 623), (77, 624), (123456, 1000), (9, 3 * 624 + 5)):
        a = C.c_void_p(L.tdr_rng_create(C.c_uint32(seed)))
        for _ in range(burn):
            L.tdr_rng_uniform_host(a)
        words = np.zeros(640, np.uint32)
        assert L.tdr_rng_get_state_host(a, words.ctypes.data_as(C.c_void_p)) == 0
        assert words[624] <= 624 and not words[625:].any()
        b = C.c_void_p(L.tdr_rng_create(C.c_uint32(424242)))
        assert L.tdr_rng_set_state_host(b, words.ctypes.data_as(C.c_void_p)) == 0
        za, zb = _host_normals(L, a, 300, False), _host_normals(L, b, 300, False)
        assert np.array_equal(za.view(np.uint32), zb.view(np.uint32))
        L.tdr_rng_destroy(a)
        L.tdr_rng_destroy(b)
    bad = np.zeros(640, np.uint32)
    bad[625] = 1     # the device's "ran out of attempts" flag: the host refuses such a state
    a = C.c_void_p(L.tdr_rng_create(C.c_uint32(1)))
    assert L.tdr_rng_set_state_host(a, bad.ctypes.data_as(C.c_void_p)) != 0
    L.tdr_rng_destroy(a)


@pytest.fixture(scope="module")
def k():
    import torch
    from top_down_renderer_amd.kernels import HipKernels
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return HipKernels()


@pytest.mark.gpu
@pytest.mark.parametrize("n,freeze,burn", [(1, False, 0), (1, True, 5), (2, False, 623), (7, False, 624), (64, True, 100),
                                           (1000, False, 0), (1000, True, 311), (20_000, False, 7), (100_003, False, 12345),
                                           (100_000, True, 1)])
def test_device_normals_are_the_hosts_bit_for_bit(k, n, freeze, burn):
    """One propagate call's normals for n particles: the device's [n][4] against libstdc++'s, bit for bit; the generator
    state afterwards: an engine rebuilt from it draws what the host engine draws next; then the uniform of the resample."""
    L = k.lib
    seed = 1000 + n + burn
    host = C.c_void_p(L.tdr_rng_create(C.c_uint32(seed)))
    for _ in range(burn):
        L.tdr_rng_uniform_host(host)
    state = k.rng_state_to_device(host)
    z_host = _host_normals(L, host, n, freeze)
    z = k.zeros((n, 4))
    k.rng_propagate_normals_dev(state, n, 0, n, freeze, z, n)
    got = z.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), z_host.view(np.uint32))
    # the uniform draw of the resample continues the stream on the device ...
    u = k.zeros((64,))
    k.rng_uniform_dev(state, u)
    assert float(u[0].item()) == float(L.tdr_rng_uniform_host(host))
    # ... a second call continues it further ...
    z2_host = _host_normals(L, host, min(n, 500), not freeze)
    z2 = k.zeros((min(n, 500), 4))
    k.rng_propagate_normals_dev(state, min(n, 500), 0, min(n, 500), not freeze, z2, n)
    assert np.array_equal(z2.cpu().numpy().view(np.uint32), z2_host.view(np.uint32))
    # ... and the host engine rebuilt from the device state is where the host's own engine is
    back = C.c_void_p(L.tdr_rng_create(C.c_uint32(0)))
    k.rng_state_to_host(back, state)
    a, b = _host_normals(L, host, 50, False), _host_normals(L, back, 50, False)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for h in (host, back):
        L.tdr_rng_destroy(h)


@pytest.mark.gpu
def test_a_rank_of_a_sharded_filter_keeps_its_slice_and_the_common_state(k):
    """Ranks pass their own [lo, hi): each gets exactly its particles' normals, all end in the same generator state."""
    L = k.lib
    n, world = 4096, 4
    host = C.c_void_p(L.tdr_rng_create(C.c_uint32(31)))
    z_host = _host_normals(L, host, n, False)
    states = []
    for r in range(world):
        g = C.c_void_p(L.tdr_rng_create(C.c_uint32(31)))
        st = k.rng_state_to_device(g)
        z = k.zeros((n // world, 4))
        k.rng_propagate_normals_dev(st, n, r * n // world, (r + 1) * n // world, False, z, n)
        assert np.array_equal(z.cpu().numpy().view(np.uint32), z_host[r * n // world:(r + 1) * n // world].view(np.uint32))
        states.append(st.cpu().numpy()[:626].copy())
        L.tdr_rng_destroy(g)
    for s in states[1:]:
        assert np.array_equal(s, states[0])
    L.tdr_rng_destroy(host)


@pytest.mark.gpu
def test_many_uniform_draws_cross_the_blocks(k):
    """1500 single-word draws on the device: over two twists of the state."""
    L = k.lib
    host = C.c_void_p(L.tdr_rng_create(C.c_uint32(2024)))
    state = k.rng_state_to_device(host)
    u = k.zeros((64,))
    for i in range(1500):
        k.rng_uniform_dev(state, u)
        if i % 97 == 0 or 620 <= i <= 630 or 1244 <= i <= 1252:
            assert float(u[0].item()) == float(L.tdr_rng_uniform_host(host)), i
        else:
            L.tdr_rng_uniform_host(host)
    L.tdr_rng_destroy(host)


@pytest.mark.gpu
def test_the_pipe_draws_ahead_and_never_changes_the_stream(k):
    """Call sequences through a tdr_rng_pipe against the host engine drawing the same things in the same order: the step's
    own pattern (normals, uniform, normals, ... — everything after the first call is served from what was drawn ahead),
    and every way of breaking it: two propagates in a row, two uniforms, a change of the particle count and of the freeze
    flag, a sharded range, the host taking the stream back in the middle and handing it over again."""
    L = k.lib
    host = C.c_void_p(L.tdr_rng_create(C.c_uint32(99)))
    for _ in range(17):
        L.tdr_rng_uniform_host(host)
    mirror = C.c_void_p(L.tdr_rng_create(C.c_uint32(99)))
    for _ in range(17):
        L.tdr_rng_uniform_host(mirror)
    pipe = k.rng_pipe_create(6000)
    pipe.from_host(host)
    seq = [("n", 5000, 0, 5000, 0), ("u",), ("n", 5000, 0, 5000, 0), ("u",), ("n", 5000, 0, 5000, 0),   # the step's pattern
           ("n", 5000, 0, 5000, 0),                      # two propagates in a row
           ("u",), ("u",),                               # two uniforms
           ("n", 3000, 0, 3000, 0), ("u",),              # another count
           ("n", 3000, 0, 3000, 1), ("u",),              # the freeze flag
           ("n", 3000, 0, 3000, 1), ("host",),           # the host takes the stream (after a draw-ahead started) ...
           ("u",), ("n", 6000, 1500, 3000, 0), ("u",),   # ... and gives it back; a rank's slice of a sharded call
           ("n", 6000, 1500, 3000, 0), ("u",), ("n", 6000, 1500, 3000, 0), ("host",)]
    for step in seq:
        if step[0] == "n":
            _, n, lo, hi, fr = step
            z = pipe.normals(n, lo, hi, fr)
            got = k.read_device_floats(z, 4 * (hi - lo)).reshape(-1, 4)
            ref = _host_normals(L, mirror, n, fr)[lo:hi]
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), step
        elif step[0] == "u":
            if not pipe.on_device():
                pipe.from_host(host)
            got = float(k.read_device_floats(pipe.uniform(), 1)[0])
            assert got == float(L.tdr_rng_uniform_host(mirror)), step
        else:
            pipe.to_host(host)
            a, b = _host_normals(L, host, 20, False), _host_normals(L, mirror, 20, False)   # the host engine continues
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for h in (host, mirror):
        L.tdr_rng_destroy(h)


# ---- jump-ahead: long calls fill stretches of the stream side by side (csrc/tdr_rng.hip: mt_jump_kernel) ---------------------
def test_committed_jump_polynomials_are_the_generators_output(tmp_path):
    """csrc/tdr_mt_jump.h is what tools/gen_mt_jump.py computes (characteristic polynomial by Berlekamp-Massey, t^J mod it
    by square-and-multiply), and the generator's own check — the host restatement of the device's procedure jumps exactly
    STRIDE and 2 STRIDE honest block steps — passes."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_mt_jump", os.path.join(root, "tools", "gen_mt_jump.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    g.OUT = str(tmp_path / "jump.h")
    g.main()                                        # (runs self_check)
    assert open(g.OUT).read() == open(os.path.join(root, "top_down_renderer_amd", "csrc", "tdr_mt_jump.h")).read()


@pytest.mark.gpu
@pytest.mark.parametrize("n,freeze,burn", [(16_000, False, 3), (20_000, True, 700), (100_003, False, 12345), (1_000_003, False, 99)])
def test_stretches_reached_by_jump_ahead_give_the_serial_stream(k, n, freeze, burn):
    """The same call with the raw stream walked by one wave and filled in stretches (their first blocks reached by jump-ahead
    polynomials): normals, uniform and a following call bit for bit — and both are the host engine's (the test above)."""
    import time
    import torch
    L = k.lib
    before = L.tdr_config_tuning(b"mt_stretches", -1)
    out, ms = {}, {}
    try:
        for mode in (0, 1):
            L.tdr_config_tuning(b"mt_stretches", mode)
            host = C.c_void_p(L.tdr_rng_create(C.c_uint32(4000 + n)))
            for _ in range(burn):
                L.tdr_rng_uniform_host(host)
            state = k.rng_state_to_device(host)
            z = k.zeros((n, 4))
            k.rng_propagate_normals_dev(state, n, 0, n, freeze, z, n)     # (first use: allocations)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            z2 = k.zeros((n, 4))
            k.rng_propagate_normals_dev(state, n, 0, n, freeze, z2, n)
            torch.cuda.synchronize()
            ms[mode] = (time.perf_counter() - t0) * 1e3
            u = k.zeros((64,))
            k.rng_uniform_dev(state, u)
            out[mode] = (z.cpu().numpy().view(np.uint32), z2.cpu().numpy().view(np.uint32), float(u[0].item()))
            if mode == 0 and n <= 100_003:
                zh = _host_normals(L, host, n, freeze)
                assert np.array_equal(out[0][0], zh.view(np.uint32))
            L.tdr_rng_destroy(host)
    finally:
        L.tdr_config_tuning(b"mt_stretches", before)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2]
    print(f"n = {n}: one wave {ms[0]:.3f} ms, stretches {ms[1]:.3f} ms (whole call, incl. attempts / scan / normals)")
