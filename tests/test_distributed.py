"""CPU tests of the multi-rank path (world_size 2, gloo): the sharded ParticleFilter must reproduce the single-rank
filter bit for bit — same raw weights, same normalised weights on every rank, same resample indices, same states —
because every rank runs the same statistics / prefix code on the same all-gathered arrays.

The kernels are stood in for by tests/oracle_backend.py (CPU oracle); what is under test is the product's host logic
in top_down_renderer_amd/particle_filter.py: particle partition, scan broadcast, weight/last_dist all-gather,
per-rank output slices, state all-gather + gather by global index, shared-seed host RNG.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N = 96
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from top_down_renderer_amd import synth
    cfg = synth.Config("dist", 3000, 3, 32, 24, 300, N, seed=4242)
    sc = synth.make_scene(cfg)
    st = sc.states.copy()
    st["have_init"][5:11] = 0          # exercise the init search on both shards
    st["have_init"][60:64] = 0
    st["init_x_px"][7] = -400.0        # un-initialised and all unknown
    st["init_x_px"][70] = -400.0       # NaN weight on the second shard
    return sc, cfg, st


def _run_filter(group, out_path, n_targets=(None, 64, None)):
    from oracle import c_oracle as oracle
    from oracle_backend import OracleKernels, OracleMapStub
    from top_down_renderer_amd.particle_filter import FilterParams, ParticleFilter

    sc, cfg, st = _scene()
    k = OracleKernels()
    m = OracleMapStub(k, sc.class_maps, sc.class_mask, 1.0)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    f = ParticleFilter(N, m, FilterParams(fixed_scale=1.0, regularization=0.3), seed=99, group=group, kernels=k,
                       init_particles=False)
    f.set_states(st)
    rank = f.comm.rank
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    log = {}
    # n_targets: shrink once — exercises N' != N and re-partitioning
    for step in range(STEPS):
        f.propagate((1.0, 0.2), 0.02)
        # only rank 0 holds the real scan; the other rank must receive it through the broadcast in update()
        f.update(scan if rank == 0 else np.zeros_like(scan), None, cfg.res, n_target=n_targets[step])
        log[f"raw{step}"] = f.raw_weights()
        log[f"w{step}"] = f.weights()
        log[f"idx{step}"] = f.resample_indices()
        log[f"st{step}"] = f.get_states().view(np.uint8).reshape(-1, 28)
        log[f"n{step}"] = np.int64(f.numParticles())
        # the reference node's order (top_down_render.cpp:333,354): computeMeanCov and meanLikelihood FIRST — both
        # all-gather the resampled states — then computeCov and maxLikelihood, which must still see the PRE-resample
        # max-likelihood particle (also across the change of N at step 1)
        log[f"cov{step}"] = f.computeMeanCov()
        log[f"mean{step}"] = f.meanLikelihood()
        log[f"covml{step}"] = f.computeCov()
        log[f"ml{step}"] = f.maxLikelihood()
    log["calls"] = np.asarray([c[1] for c in k.calls if c[0] == "score"], np.int64)
    np.savez(out_path, **log)


def _worker(rank, world, port, tmp, n_targets=(None, 64, None)):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _run_filter(dist.group.WORLD, os.path.join(tmp, f"rank{rank}.npz"), n_targets)
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def runs(oracle):
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp(prefix="tdr_dist_")
    _run_filter(None, os.path.join(tmp, "single.npz"))
    mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    load = lambda n: np.load(os.path.join(tmp, n), allow_pickle=False)
    return load("single.npz"), load("rank0.npz"), load("rank1.npz")


def test_two_ranks_equal_one_rank_bit_for_bit(runs):
    single, r0, r1 = runs
    for step in range(STEPS):
        assert int(r0[f"n{step}"]) == int(r1[f"n{step}"]) == int(single[f"n{step}"])
        # every rank holds the same global weights, identical to the single-rank ones
        assert np.array_equal(r0[f"w{step}"], single[f"w{step}"], equal_nan=True)
        assert np.array_equal(r1[f"w{step}"], single[f"w{step}"], equal_nan=True)
        # shards concatenate to the single-rank arrays
        for key in ("raw", "idx", "st"):
            both = np.concatenate([r0[f"{key}{step}"], r1[f"{key}{step}"]])
            assert np.array_equal(both, single[f"{key}{step}"], equal_nan=True), (key, step)
        assert np.array_equal(r0[f"ml{step}"], single[f"ml{step}"]) and np.array_equal(r1[f"ml{step}"], single[f"ml{step}"])
        assert np.allclose(r0[f"cov{step}"], single[f"cov{step}"], rtol=1e-5, atol=1e-5)
        for key in ("mean", "covml"):
            assert np.allclose(r0[f"{key}{step}"], single[f"{key}{step}"], rtol=1e-5, atol=1e-5), (key, step)
            assert np.array_equal(r0[f"{key}{step}"], r1[f"{key}{step}"]), (key, step)


def test_max_likelihood_is_the_pre_resample_argmax(runs, oracle):
    """maxLikelihood after the pose statistics (the reference node's call order) is mlState of the particle that won the
    update — recomputed here from the logged pre-resample quantities of the single-rank run."""
    single, r0, r1 = runs
    sc, cfg, st = _scene()
    fp = oracle.make_params(cfg.ncls, regularization=0.3)
    rng = oracle.Rng(99)
    oracle.propagate(st, 1.0, 0.2, 0.02, True, fp, rng)     # step 0's pre-resample set (propagate is deterministic)
    best = int(np.argmax(single["w0"]))
    s = st[best]
    # theta of an un-initialised winner is chosen by the search; position and scale are not touched by it
    exp = np.asarray([s["dx_m"] * s["scale"] + s["init_x_px"], s["dy_m"] * s["scale"] + s["init_y_px"]], np.float32)
    for run in (single, r0, r1):
        assert np.allclose(run["ml0"][:2], exp, rtol=1e-6, atol=1e-4)
        assert run["ml0"][3] == s["scale"]


def test_each_rank_scores_only_its_shard(runs):
    single, r0, r1 = runs
    assert list(single["calls"]) == [96, 96, 64]
    assert list(r0["calls"]) == [48, 48, 32] and list(r1["calls"]) == [48, 48, 32]


def test_sharded_run_matches_plain_oracle_sequence(runs, oracle):
    """The single-rank filter (product host logic + test double) follows the oracle's own step sequence."""
    single, _, _ = runs
    sc, cfg, st = _scene()
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, 1.0)
    tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, 1.0)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    fp = oracle.make_params(cfg.ncls, regularization=0.3)
    rng = oracle.Rng(99)
    last = oracle.propagate(st, 1.0, 0.2, 0.02, True, fp, rng)
    raw = oracle.compute_weights(om, tab, cfg.nb, cfg.nr, scan, cfg.res, fp, st)
    w, best, _ = oracle.update_weights(raw, last)
    idx = oracle.resample_prefix(w, N, rng.uniform())
    assert np.array_equal(np.isnan(raw), np.isnan(single["raw0"]))
    assert np.allclose(raw, single["raw0"], rtol=1e-5, atol=0, equal_nan=True)
    assert np.allclose(w, single["w0"], rtol=1e-5, atol=0)
    assert (idx != single["idx0"]).sum() <= 2


def test_eight_ranks_equal_one_rank_with_a_count_that_does_not_divide(oracle):
    """world_size 8 (the node the scaling run uses): 12 particles per rank, among them particles without a heading on
    several shards, one of those with an all-unknown window, and a NaN weight on a later shard (_scene).  The second update
    asks for 70 particles: a sharded filter rounds the count DOWN to a multiple of the world size (64: 8 per rank —
    ParticleFilter.update), so the one-rank run it must reproduce bit for bit is the one asked for 64."""
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp(prefix="tdr_dist8_")
    _run_filter(None, os.path.join(tmp, "single.npz"), (None, 64, None))
    mp.spawn(_worker, args=(8, _free_port(), tmp, (None, 70, None)), nprocs=8, join=True)
    load = lambda n: np.load(os.path.join(tmp, n), allow_pickle=False)
    single, ranks = load("single.npz"), [load(f"rank{r}.npz") for r in range(8)]
    assert [int(single[f"n{s}"]) for s in range(STEPS)] == [96, 64, 64]
    assert np.isnan(single["raw0"]).any()
    for step in range(STEPS):
        for r in ranks:
            assert int(r[f"n{step}"]) == int(single[f"n{step}"])
            assert np.array_equal(r[f"w{step}"], single[f"w{step}"], equal_nan=True)
            assert np.array_equal(r[f"ml{step}"], single[f"ml{step}"])
            assert np.array_equal(r[f"mean{step}"], ranks[0][f"mean{step}"])
        for key in ("raw", "idx", "st"):
            allr = np.concatenate([r[f"{key}{step}"] for r in ranks])
            assert np.array_equal(allr, single[f"{key}{step}"], equal_nan=True), (key, step)
    assert all(list(r["calls"]) == [12, 12, 8] for r in ranks)


def test_particle_count_must_divide_evenly():
    from oracle_backend import OracleKernels, OracleMapStub
    from top_down_renderer_amd.particle_filter import FilterParams, ParticleFilter

    class FakeComm:
        pass
    sc, cfg, st = _scene()
    k = OracleKernels()
    m = OracleMapStub(k, sc.class_maps, sc.class_mask, 1.0)
    m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    f = ParticleFilter(N, m, FilterParams(fixed_scale=1.0), kernels=k, init_particles=False)
    with pytest.raises(ValueError):
        f.set_states(st[: N + 0].repeat(2))   # more particles than the filter's maximum
