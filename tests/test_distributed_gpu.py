"""GPU test of the multi-rank path with the REAL kernels: two processes share the one GPU of the test box (gloo for the
collectives, libtdr_hip.so for everything else) and must reproduce the single-process filter bit for bit — raw weights,
normalised weights, resample indices, states — over several steps, with the device RNG (keyed by step and GLOBAL
particle index), the locality order on (each rank sorts its own shard) and a change of the particle count.
On an 8-GPU node the same code runs with backend "nccl" (RCCL over xGMI); the driver's scaling bench covers that."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N = 6144
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(group, out_path, six_classes, force_collectives=False):
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd import synth
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    cart = six_classes == "cart"
    # six classes: records of 8 floats, the init search runs on the matrix cores (16 particles per MFMA tile)
    if cart:     # BASELINE config 4's path: Cartesian render + Cartesian window score (definition in include/tdr.h)
        cfg = synth.Config("distc", 20000, 6, 40, 64, 700, N, polar=False, seed=79)
    else:
        cfg = synth.Config("dist6", 20000, 6, 64, 48, 700, N, seed=77) if six_classes else synth.CONFIGS["c1"]
    sc = synth.make_scene(cfg, n_particles=N)
    st = sc.states.copy()
    if not cart:
        st["have_init"][100:140] = 0            # init search on the first shard
        st["have_init"][4000:4040] = 0          # ... and on the second
    if cart:
        m = pkg.TopDownMap(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
        m.setWindow(cfg.nb, cfg.nr)
    else:
        m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
        m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    f = pkg.ParticleFilter(N, m, pkg.FilterParams(fixed_scale=1.0), seed=7, group=group, kernels=k,
                           parity_rng=False, locality_every=1, init_particles=False,
                           force_collectives=force_collectives)
    assert f.comm.active == (group is not None)
    f.set_states(st)
    r = (pkg.ScanRenderer if cart else pkg.ScanRendererPolar)(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    log = {}
    n_targets = [None, 4096, None]
    render = (lambda pts: r.renderSemanticTopDown(pts, cfg.res)) if cart else \
        (lambda pts: r.renderSemanticTopDown(pts, cfg.res, cfg.ang_res))
    for step in range(STEPS):
        f.propagate((1.0, 0.2), 0.02)
        if f.comm.rank == 0:
            render(sc.pts)
            scan = r.last_scan()
        else:   # receives the render through the broadcast inside update()
            render(sc.pts[:1])
            scan = r.last_scan()
        f.update(scan, None, cfg.res, n_target=n_targets[step])
        log[f"raw{step}"] = f.raw_weights()
        log[f"w{step}"] = f.weights()
        log[f"idx{step}"] = f.resample_indices()
        log[f"st{step}"] = f.get_states().view(np.uint8).reshape(-1, 28)
        # the reference node's order (top_down_render.cpp:333,354): the pose statistics first (they all-gather the
        # resampled states), then the max-likelihood particle, which is the PRE-resample one
        log[f"mean{step}"] = f.meanLikelihood()
        log[f"cov{step}"] = f.computeMeanCov()
        log[f"covml{step}"] = f.computeCov()
        log[f"ml{step}"] = f.maxLikelihood()
    np.savez(out_path, **log)


def _worker(rank, world, port, tmp, six_classes):
    import torch.distributed as dist
    if six_classes == "cart":
        # few window chunks, so that their number depends on the particle count: a shard must still split the window
        # like the whole filter does (tdr_k_score_cart's n_total).
        from top_down_renderer_amd import _lib
        _lib.load().tdr_config_tuning(b"score_waves", 256)
    if world == 1:
        _run(None, os.path.join(tmp, "single.npz"), six_classes)
        return
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _run(dist.group.WORLD, os.path.join(tmp, f"rank{rank}.npz"), six_classes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("six_classes", [False, True, "cart"])
def test_two_gpu_ranks_equal_one_rank_bit_for_bit(six_classes):
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp(prefix="tdr_dist_gpu_")
    mp.spawn(_worker, args=(1, _free_port(), tmp, six_classes), nprocs=1, join=True)   # the one-rank filter
    mp.spawn(_worker, args=(2, _free_port(), tmp, six_classes), nprocs=2, join=True)
    load = lambda n: np.load(os.path.join(tmp, n), allow_pickle=False)  # noqa: E731
    single, r0, r1 = load("single.npz"), load("rank0.npz"), load("rank1.npz")
    for step in range(STEPS):
        assert np.array_equal(r0[f"w{step}"], single[f"w{step}"]) and np.array_equal(r1[f"w{step}"], single[f"w{step}"])
        for key in ("raw", "idx", "st"):
            both = np.concatenate([r0[f"{key}{step}"], r1[f"{key}{step}"]])
            assert np.array_equal(both, single[f"{key}{step}"], equal_nan=True), (key, step)
        assert np.array_equal(r0[f"ml{step}"], single[f"ml{step}"]) and np.array_equal(r1[f"ml{step}"], single[f"ml{step}"])
        for key in ("mean", "cov", "covml"):
            assert np.array_equal(r0[f"{key}{step}"], single[f"{key}{step}"]), (key, step)
            assert np.array_equal(r1[f"{key}{step}"], single[f"{key}{step}"]), (key, step)
    assert len(single["st2"]) == 4096 and len(r0["st2"]) == 2048


def _rccl_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        _run(dist.group.WORLD, os.path.join(tmp, "rccl.npz"), True, force_collectives=True)
    finally:
        dist.destroy_process_group()


def test_rccl_backend_world_size_one_equals_groupless_run():
    """backend "nccl" IS RCCL on ROCm.  With the one GPU of this box the group has a single rank, but every collective
    of the sharded filter goes through RCCL for real: the scan broadcast, the blocking all-gather of {raw weights,
    last_dist}, the ASYNC all-gather of the state planes (its own stream, waited for at the gather) and the gathers
    behind the pose statistics.  Results must equal the group-less filter bit for bit.  (The 8-GPU run is the driver's.)"""
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp(prefix="tdr_rccl_")
    _run(None, os.path.join(tmp, "single.npz"), True)
    mp.spawn(_rccl_worker, args=(1, _free_port(), tmp), nprocs=1, join=True)
    load = lambda n: np.load(os.path.join(tmp, n), allow_pickle=False)  # noqa: E731
    single, rc = load("single.npz"), load("rccl.npz")
    assert set(single.files) == set(rc.files)
    for key in single.files:
        assert np.array_equal(single[key], rc[key], equal_nan=True), key
