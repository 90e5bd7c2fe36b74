"""The C++ handle layer's sharded filter (tdr_filter_create_sharded, csrc/tdr_host.cpp + csrc/tdr_comm.cpp): particles
partitioned over the ranks of a tdr_comm, scan broadcast from rank 0, one all-gather of {raw weight, last_dist}, one
all-gather of the state planes.  Must equal the unsharded handle bit for bit.

  * RCCL transport (ncclAllGather / ncclBroadcast called directly): the one GPU of the test box gives a one-rank
    communicator — every collective still goes through RCCL.
  * caller-supplied transport, two ranks sharing the GPU: the collectives are done by this test (device -> host -> gloo
    -> device), the C++ sharding logic is what is under test.
The 8-GPU run over xGMI is the driver's."""
import ctypes as C
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N = 4096
STEPS = 3
N_TARGETS = [-1, 3072, -1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from top_down_renderer_amd import synth
    cfg = synth.Config("shard", 20000, 6, 64, 48, 700, N, seed=78)
    sc = synth.make_scene(cfg)
    st = sc.states.copy()
    st["have_init"][50:90] = 0            # init search on the first shard ...
    st["have_init"][3000:3040] = 0        # ... and on the second
    return cfg, sc, st


def _run(comm_kind, rank, world, out_path, dist=None):
    """comm_kind: None (plain handle), "rccl" or "callbacks"."""
    import torch
    from top_down_renderer_amd import _lib
    from top_down_renderer_amd._lib import FilterParamsC, check
    L = _lib.load()
    torch.cuda.set_device(0)
    # the un-initialised particles' 40-rotation search through the handle's half-record scratch (by default only filters
    # of >= 8192 particles take it): allocated by the first update, on the map.  3000 lies between a shard (2048) and the
    # filter (4096): the filter's total decides, so the shards take the same kernel as the plain handle
    L.tdr_config_rec16_min_particles(3000)
    cfg, sc, st = _scene()
    ncls, H, W = sc.class_maps.shape
    vp = C.c_void_p

    def P(a):
        return a.ctypes.data_as(vp)

    m = vp()
    check(L.tdr_map_create(C.byref(m)))
    maps_cm = np.ascontiguousarray(np.transpose(sc.class_maps, (0, 2, 1)), np.float32)
    mask_cm = np.ascontiguousarray(sc.class_mask.T, np.uint8)
    check(L.tdr_map_set(m, P(maps_cm), P(mask_cm), ncls, H, W, C.c_float(1.0), 0, 0))
    check(L.tdr_map_sample_pts_polar(m, cfg.nb, cfg.nr, C.c_float(cfg.ang_res)))
    lut = np.ascontiguousarray(sc.lut, np.int32)
    r = vp()
    check(L.tdr_renderer_create(P(lut), C.byref(r)))
    fp = FilterParamsC()
    fp.pos_cov, fp.theta_cov, fp.regularization = 0.3, np.pi / 100, 0.15
    fp.init_pos_px_x = fp.init_pos_px_y = fp.init_pos_px_cov = -1
    fp.init_pos_m_x = fp.init_pos_m_y = float("inf")
    fp.init_pos_deg_theta, fp.init_pos_deg_cov = float("inf"), 10
    fp.fixed_scale, fp.scale_log_min, fp.scale_log_max, fp.num_classes = 1.0, -0.1, 1.0, ncls
    for i in range(ncls):
        fp.class_weights[i] = 1.0
    comm, keep = vp(), []
    if comm_kind == "rccl":
        uid = (C.c_char * 128)()
        check(L.tdr_comm_rccl_unique_id(uid))
        check(L.tdr_comm_create_rccl(world, rank, uid, C.byref(comm)))
    elif comm_kind == "callbacks":
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [vp]
        AG = C.CFUNCTYPE(C.c_int, vp, vp, vp, C.c_size_t, vp)
        BC = C.CFUNCTYPE(C.c_int, vp, vp, C.c_size_t, C.c_int, vp)

        def all_gather(ctx, send, recv, nbytes, stream):
            hip.hipStreamSynchronize(stream)
            mine = torch.empty(nbytes, dtype=torch.uint8)
            assert hip.hipMemcpy(mine.data_ptr(), send, nbytes, 2) == 0           # device -> host
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, mine)
            allb = torch.cat(parts)
            assert hip.hipMemcpy(recv, allb.data_ptr(), nbytes * world, 1) == 0    # host -> device
            return 0

        def broadcast(ctx, buf, nbytes, root, stream):
            hip.hipStreamSynchronize(stream)
            t = torch.empty(nbytes, dtype=torch.uint8)
            if rank == root:
                assert hip.hipMemcpy(t.data_ptr(), buf, nbytes, 2) == 0
            dist.broadcast(t, src=root)
            if rank != root:
                assert hip.hipMemcpy(buf, t.data_ptr(), nbytes, 1) == 0
            return 0

        class Ops(C.Structure):
            _fields_ = [("ctx", vp), ("all_gather", AG), ("broadcast", BC)]
        ops = Ops(None, AG(all_gather), BC(broadcast))
        keep.append(ops)
        check(L.tdr_comm_create(world, rank, C.byref(ops), C.byref(comm)))
    f = vp()
    if comm_kind:
        check(L.tdr_filter_create_sharded(m, N, C.byref(fp), 7, comm, C.byref(f)))
        assert L.tdr_comm_world(comm) == world and L.tdr_comm_rank(comm) == rank
    else:
        check(L.tdr_filter_create(m, N, C.byref(fp), 7, C.byref(f)))
    check(L.tdr_filter_configure(f, 1, 1))       # reference-ordered host RNG, locality order on
    check(L.tdr_filter_set_states(f, P(st), N))
    pcl = np.zeros((len(sc.pts), 8), np.float32)
    pcl[:, :3], pcl[:, 4] = sc.pts[:, :3], sc.pts[:, 3]
    log = {}
    for step in range(STEPS):
        check(L.tdr_filter_propagate(f, C.c_float(1.0), C.c_float(0.2), C.c_float(0.02)))
        if rank == 0:
            check(L.tdr_renderer_render(r, 1, P(pcl), 8, 4, len(pcl), C.c_float(cfg.res), C.c_float(cfg.ang_res), ncls,
                                        cfg.nb, cfg.nr, None))
            check(L.tdr_filter_update(f, None, r, C.c_float(cfg.res), N_TARGETS[step]))
        else:   # receives rank 0's packed scan through the broadcast
            check(L.tdr_filter_update(f, None, None, C.c_float(cfg.res), N_TARGETS[step]))
        n = L.tdr_filter_num_particles(f)
        nl = L.tdr_filter_num_local(f)
        assert nl * world == n
        w = np.empty(n, np.float32)
        check(L.tdr_filter_get_weights(f, P(w), n))
        idx = np.empty(nl, np.int32)
        check(L.tdr_filter_get_resample_indices(f, P(idx), nl))
        stl = np.zeros(nl, st.dtype)
        check(L.tdr_filter_get_states(f, P(stl), nl))
        mean, cov, ml, covml = (np.zeros(4, np.float32), np.zeros(16, np.float32), np.zeros(4, np.float32),
                                np.zeros(16, np.float32))
        check(L.tdr_filter_mean_cov(f, 0, P(mean), P(cov)))     # the node's order: pose statistics first ...
        check(L.tdr_filter_mean_cov(f, 1, P(ml), P(covml)))     # ... then the max-likelihood particle
        log.update({f"w{step}": w, f"idx{step}": idx, f"st{step}": stl.view(np.uint8).reshape(-1, 28),
                    f"mean{step}": mean, f"cov{step}": cov, f"ml{step}": ml, f"covml{step}": covml,
                    f"n{step}": np.int64(n)})
    check(L.tdr_filter_compute_gmm(f))
    log["count"] = np.int64(L.tdr_filter_adaptive_count(f))
    L.tdr_filter_destroy(f)
    if comm_kind:
        L.tdr_comm_destroy(comm)
    np.savez(out_path, **log)


def _worker(rank, world, port, tmp, kind):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if kind == "callbacks":
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _run(kind, rank, world, os.path.join(tmp, f"{kind}{rank}.npz"), dist if kind == "callbacks" else None)
    finally:
        if kind == "callbacks":
            dist.destroy_process_group()


@pytest.fixture(scope="module")
def runs():
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp(prefix="tdr_shard_")
    mp.spawn(_worker, args=(1, _free_port(), tmp, None), nprocs=1, join=True)          # the plain handle
    os.rename(os.path.join(tmp, "None0.npz"), os.path.join(tmp, "plain.npz"))
    mp.spawn(_worker, args=(1, _free_port(), tmp, "rccl"), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, _free_port(), tmp, "callbacks"), nprocs=2, join=True)
    load = lambda n: np.load(os.path.join(tmp, n), allow_pickle=False)  # noqa: E731
    return load("plain.npz"), load("rccl0.npz"), load("callbacks0.npz"), load("callbacks1.npz")


def test_rccl_one_rank_equals_plain_handle(runs):
    plain, rccl, _, _ = runs
    assert set(plain.files) == set(rccl.files)
    for key in plain.files:
        assert np.array_equal(plain[key], rccl[key], equal_nan=True), key


def test_two_ranks_equal_plain_handle_bit_for_bit(runs):
    plain, _, r0, r1 = runs
    for step in range(STEPS):
        assert int(r0[f"n{step}"]) == int(r1[f"n{step}"]) == int(plain[f"n{step}"])
        for key in ("w", "mean", "cov", "ml", "covml"):      # global quantities: identical on every rank
            assert np.array_equal(r0[f"{key}{step}"], plain[f"{key}{step}"]), (key, step)
            assert np.array_equal(r1[f"{key}{step}"], plain[f"{key}{step}"]), (key, step)
        for key in ("idx", "st"):                            # shards concatenate to the unsharded arrays
            both = np.concatenate([r0[f"{key}{step}"], r1[f"{key}{step}"]])
            assert np.array_equal(both, plain[f"{key}{step}"]), (key, step)
    assert int(plain["n1"]) == 3072 and len(r0["st1"]) == 1536
    assert int(r0["count"]) == int(r1["count"]) == int(plain["count"])
