#!/usr/bin/env python3
"""bench.py — particle-updates/s of the per-scan particle-filter update (render + propagate + score + weight
statistics + resample) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload: N = 1 -> BASELINE.json configs[1] ("c2", 100k particles).  N > 1 -> configs[2] ("c3": the same scan / map with
1 M particles over 8 GPUs, i.e. 125 000 per GPU — weak scaling at that per-GPU load for every N > 1).

A "step" is one full pass of the hot path over one synthetic scan: raster kernel over the packed scan points (resident in
HBM like every other input when the timed region starts; `pcie_inclusive` is the same step with the raster kernel reading the
points out of pinned host memory instead — the form the boundary hands them over in, the transfer inside the kernel),
propagate kernel (device counter-based RNG), scoring kernel over this rank's particles, weight statistics,
order-exact prefix, resample + state gather (and, for N > 1, the scan broadcast and the weight/state all-gathers over
RCCL).  Map, sampling table and particles are resident in HBM before the timed region.  Workload at N = 1 is
BASELINE.json configs[1] ("c2": 100k-pt scan, 6 classes, 256x256 polar render, 4000x4000 map, 100k particles);
for N > 1 every GPU holds the same number of particles (weak scaling; N = 8 with --particles-per-gpu 125000 is
configs[2]).

One JSON line on stdout (rank 0).  `roofline` prices the scoring launch (its average duration measured live with HIP
events on its launch stream; a polar launch runs two kernels, the events span both) against the 8 TB/s HBM peak:
  * `traffic` = read bytes per launch the L2s request from the fabric, measured with rocprofv3 --pmc
    TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum in a pass of its own over this same command (tools/traffic_from_pmc.py ->
    profiles/score_traffic.json) and REPLAYED here — `traffic_source` says so and carries the record's source hash; when
    the kernel sources or the launch shape differ from the record's, `traffic`, `achieved`, `frac` and `issue` are null.
    Hits in the Infinity Cache are among the requests: an upper bound on what HBM delivers.
  * `achieved` / `frac` = those COUNTER bytes per launch / the launch duration measured in this run (/ peak): the fraction of
    the HBM roofline the launch really uses.  (The scoring launches are not HBM-bound: `bound` names the unit that is.)
  * `bound_unit_frac` = how far the binding unit is from ITS floor, for launches bound by the L1 address path: vector-memory
    wave-instructions of the launch (SQ_INSTS_VMEM_RD, a fourth counter pass) x 16 cycles — what a 64-lane gather costs the
    address path at best (tools/ta_cost.hip) — over the CU-cycles of the launch (256 CUs x its cycles).  1.0 = every
    address-path cycle of every CU spent on gathers that touch four lines or fewer.
  * `issue` = {valu_busy, lds_busy, l1_addr_busy, insts_per_sample} from two more counter passes (valu_busy counts four
    cycles per vector instruction — an upper bound, see tools/valu_cost.hip —, l1_addr_busy = TA_BUSY_avr over the kernels'
    cycles: the texture addressers, what a gather's cache lines cost); `bound` = the busiest of {hbm: `frac`, valu,
    lds, l1_addr};
  * `shares` = the launch's two kernels timed on their own (one extra launch behind the timed region, on the particle set
    every timed step starts from, before propagate): shift-uniform kernel over the dense particles, ray-mapped kernel over the
    scattered ones, particles in each;
  * `dense_work_rate` = the dense byte model of SURVEY.md §8(d) (B_pu = P*(4*ncls+1) + 64 per particle-update) over the
    same duration.  The launch does not move those bytes (compact records, empty bins skipped, shared lines): a WORK rate,
    not a roofline figure — `x_hbm_peak` above 1 says exactly that;
  * `sparse_algorithmic` = a second work rate: the bytes a launch that shares nothing between particles could not do without,
    from THIS run's scan — per particle one mask BIT per window sample (the known-fraction gate reads every sample) + 2 bytes
    per non-empty scan bin (the class plane's cell); `traffic_over_sparse` = what the counters saw of it;
  * `tuner` = the span tuner's state: the span in use and how many of the timed launches ran at a trial span.
`cpu_baseline` times the CPU oracle (oracle/oracle.cpp, OpenMP over particles like the reference's parallel for_each)
on a bounded sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500, help="default: a timed region of >= 2 s at configs[1]")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="", help="default: c2 on one GPU, c3 (125 000 particles per GPU) on several")
    ap.add_argument("--particles-per-gpu", type=int, default=0, help="default: the config's particle count (c3, c5: / 8)")
    ap.add_argument("--locality-every", type=int, default=1, help="recompute the cache-locality order every k steps (0 = off)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="particles in the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--device-rng", action="store_true",
                    help="time the steps with the counter-based device noise instead of the reference's std::mt19937 stream "
                         "(default: the reference's stream, reproduced on the device; the other figure is reported beside it)")
    ap.add_argument("--tuning", default="", help="name=value[,name=value...] for tdr_config_tuning (A/B measurements)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate devices)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launch / rendezvous / collect path only: every rank joins the group, all-reduces a one and rank 0 "
                         "prints a line — no GPU is touched (CPU test of the self-launch; use with --backend gloo)")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: this process — which has not touched the GPU and never will — starts
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD (no exec: a process
    must not be replaced once anything initialised the GPU, and a child keeps that rule trivially true), relays rank 0's
    JSON line and exits with the child's code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in p.stdout:
        if out.lstrip().startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        rc = 1
    raise SystemExit(rc)


def plumbing_only(a, world, rank):
    """The N > 1 launch path up to the first collective, on the CPU."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    t = torch.ones(1)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "particle-updates/sec (render+score+resample)", "value": None, "plumbing_only": True,
                          "n_gpus": world, "ranks_in_allreduce": int(t.item()), "backend": a.backend}), flush=True)
    dist.destroy_process_group()


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))   # a 1-GPU box's CPU share is 16


def cpu_baseline(sc, cfg, n_sample, threads):
    """The oracle's full update on a strided sample of the particle set: raster, propagate, score,
    weight statistics, resample, state copy.  Config 1 runs the resample as the reference writes it — the O(N N')
    double loop of src/particle_filter.cpp:172-185; the larger configs its O(N) prefix + search form (same indices)."""
    import numpy as np
    from oracle import c_oracle as oracle

    oracle.build()
    om = oracle.OracleMap(sc.class_maps, sc.class_mask, cfg.map_resolution)
    fp = oracle.make_params(cfg.ncls)
    st = np.ascontiguousarray(sc.states[:: max(1, len(sc.states) // n_sample)][:n_sample]).copy()
    first = ""
    if not st["have_init"].all():
        # config 5: the timed steady-state steps start from particles that already have a heading (bench.py's GPU leg
        # does the same); the one-off 40-rotation search is timed beside it
        t0 = time.perf_counter()
        tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, cfg.map_resolution)
        scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
        oracle.compute_weights(om, tab, cfg.nb, cfg.nr, scan, cfg.res, fp, st, nthreads=threads)   # sets theta, have_init
        first = f"; first update with the 40-rotation search: {len(st) / (time.perf_counter() - t0):.0f} particle-updates/s"
    literal = cfg.name == "c1"
    t0 = time.perf_counter()
    if cfg.polar:
        tab = oracle.polar_table(cfg.nb, cfg.nr, cfg.ang_res, cfg.map_resolution)
        scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    else:
        scan = oracle.raster_cart(sc.pts, cfg.res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    last = oracle.propagate(st, 1.0, 0.0, 0.01, True, fp, oracle.Rng(1))
    if cfg.polar:
        raw = oracle.compute_weights(om, tab, cfg.nb, cfg.nr, scan, cfg.res, fp, st, nthreads=threads)
    else:
        raw = oracle.compute_weights_cart(om, cfg.nb, cfg.nr, scan, cfg.res, fp, st, nthreads=threads)
    w, _, _ = oracle.update_weights(raw, last)
    idx = oracle.resample_literal(w, len(st), 0.5) if literal else oracle.resample_prefix(w, len(st), 0.5)
    oracle.gather_states(st, idx)
    dt = time.perf_counter() - t0
    return {"value": len(st) / dt, "unit": "particle-updates/s", "cores": threads, "kind": "port",
            "sample": f"{len(st)} of the {len(sc.states)} particles of the same scene (strided), full step incl. the "
                      f"{'literal O(N N-prime) resample loop of the reference' if literal else 'O(N) prefix + search form of the resample'}"
                      f", {dt:.2f} s on {threads} OpenMP threads (scoring parallel over particles, the rest serial like "
                      f"the reference)" + first}


def variant_shares(v, polar):
    """Which loop variant the dense kernel's wave-sectors (polar) / wave-segments (Cartesian) ran during the timed steps
    (tdr_profile_variants): shares of the total."""
    if polar:
        names = ("wg_box_all_known", "wg_box_inside", "wg_box_general", "wave_box_all_known", "wave_box_inside",
                 "wave_box_general", "far_path")
        vals = v[:7]
    else:
        names = ("all_known", "inside", "general", "plain_steps")
        vals = v[8:12]
    tot = sum(vals)
    return {n: (x / tot if tot else None) for n, x in zip(names, vals)} | {"units": tot}


def kernel_source_hash():
    """Hash of the sources the scoring kernels are built from: a traffic record made with other sources is stale."""
    import hashlib
    h = hashlib.sha256()
    for f in ("tdr_score.hip", "tdr_score_su.hip", "tdr_score_ray.hip", "tdr_cmap.hip", "tdr_score_su.h", "tdr_score_su_asm.h", "tdr_score_cart.hip", "tdr_score_cart.h", "tdr_score_cart_asm.h", "tdr_score_dev.h", "tdr_common.h", "tdr_sincosf.h"):
        h.update(open(os.path.join(ROOT, "top_down_renderer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(cfg_name, kernel, n_local):
    """(bytes per launch from profiles/score_traffic.json, issue figures, note) — None unless the record was made from
    the kernel sources in this tree, for this kernel and this launch shape."""
    tpath = os.path.join(ROOT, "profiles", "score_traffic.json")
    try:
        rec = json.load(open(tpath))["entries"][cfg_name]
    except Exception:
        return None, None, "no record"
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, None, "record is from other kernel sources"
    if rec.get("particles_per_launch") != n_local or kernel not in rec.get("kernel", ""):
        return None, None, "record is for another launch shape"
    return float(rec["hbm_bytes_per_launch"]), rec.get("issue"), rec.get("source", "")


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)          # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started {world} rank(s)")
    if a.plumbing_only:
        return plumbing_only(a, world, rank)
    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback in the product path)")
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
        group = dist.group.WORLD
        # how many ranks the backend really joined: a one on every rank's device through one all-reduce
        ones = torch.ones(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(ones)
        ranks_joined = int(ones.item())

    import top_down_renderer_amd as pkg
    from top_down_renderer_amd import synth
    from top_down_renderer_amd.kernels import HipKernels

    k = HipKernels()
    for kv in filter(None, a.tuning.split(",")):
        name, _, val = kv.partition("=")
        if k.lib.tdr_config_tuning(name.encode(), int(val)) < 0:
            raise SystemExit(f"--tuning: unknown knob {name!r}")
    cfg = synth.CONFIGS[a.config or ("c2" if world == 1 else "c3")]
    # configs 3 and 5 name a total over 8 GPUs: every GPU holds an eighth of it, whatever N is (weak scaling)
    per_gpu = a.particles_per_gpu or (cfg.n_particles // 8 if cfg.name in ("c3", "c5") else cfg.n_particles)
    n_global = per_gpu * world
    sc = synth.make_scene(cfg, n_particles=n_global)   # same seed on every rank -> identical scene
    if cfg.polar:
        m = pkg.TopDownMapPolar(pkg.Params(resolution=cfg.map_resolution), sc.class_maps, sc.class_mask, kernels=k)
        m.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
        r = pkg.ScanRendererPolar(sc.lut, kernels=k)
    else:   # BASELINE configs[3]: Cartesian render + Cartesian window score
        m = pkg.TopDownMap(pkg.Params(resolution=cfg.map_resolution), sc.class_maps, sc.class_mask, kernels=k)
        m.setWindow(cfg.nb, cfg.nr)
        r = pkg.ScanRenderer(sc.lut, kernels=k)
    r.set_output_shape(cfg.ncls, cfg.nb, cfg.nr)
    # propagate draws the reference's own noise: std::mt19937 + libstdc++'s normal_distribution in the reference's order,
    # reproduced on the device (csrc/tdr_rng.hip) — the identical-results path is the one that is timed
    f = pkg.ParticleFilter(n_global, m, pkg.FilterParams(fixed_scale=1.0), seed=1, group=group, kernels=k,
                           parity_rng=not a.device_rng, locality_every=a.locality_every, init_particles=False)
    f.set_states(sc.states)
    nl = f.n_local
    pts_host = torch.from_numpy(sc.pts).pin_memory()
    pts_dev = pts_host.to(k.device)
    scan_src = [pts_dev]      # what step() rasterises: the resident points (timed), the pinned host points (pcie_inclusive)

    def render(pts):
        if cfg.polar:
            r.renderSemanticTopDown(pts, cfg.res, cfg.ang_res)
        else:
            r.renderSemanticTopDown(pts, cfg.res)

    init_step_ms = None
    if not cfg.have_init:
        # BASELINE configs[4]: particles start without a heading; the FIRST update runs the 40-rotation search of
        # src/state_particle.cpp:195-206.  It is a one-off: timed on its own, then the steady-state steps below start
        # from the initialised set.
        # Reported twice: the very first update of the process (it also pays every first-use allocation — score workspace,
        # sort buffers, the search's half-record scratch — and the loading of the kernels' code objects), and the same
        # update repeated on the same un-initialised particles with all of that in place.
        init_ms = []
        for _ in range(2):
            f.set_states(sc.states)
            render(pts_host)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f.update(r.last_scan(), None, cfg.res)
            torch.cuda.synchronize()
            init_ms.append((time.perf_counter() - t0) * 1e3)
        init_cold_ms, init_step_ms = init_ms
        f.st, f.st_new = f.st_new, f.st     # keep the scored (now initialised) set, drop the resampled one
        f.num_particles_ = n_global
    st0 = f.st[:, :nl].clone()

    def step():
        # every step scores the SAME particle distribution (90 % Gaussian about the true pose + 10 %
        # uniform): the resampled set of the previous step is replaced by the initial one (a 2.8 MB device copy),
        # otherwise the filter converges within a few steps and later steps measure an easier, cache-friendlier case
        f.st[:, :nl].copy_(st0)
        f.num_particles_ = n_global
        # only rank 0 "receives" the scan; the others get the rasterised scan by broadcast inside update()
        if rank == 0:
            # timed: the scan's points resident in HBM.  pcie_inclusive: in pinned host memory, read from there by the raster
            # kernel (the host-to-device transfer of the 1.6 MB happens inside the kernel: no runtime copy, nothing to wait for)
            render(scan_src[0])
            scan = r.last_scan()
        else:
            scan = ("pk", k.empty((cfg.nr * cfg.nb * k.lib.tdr_rec_floats(cfg.ncls),)))
        f.propagate((1.0, 0.0), 0.01)
        f.update(scan, None, cfg.res)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    import ctypes as C
    k.lib.tdr_profile_enable(1)
    trials0 = f.score_ctx.trial_calls() if cfg.polar else 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    trials_timed = (f.score_ctx.trial_calls() - trials0) if cfg.polar else 0
    tot_ms, launches = C.c_double(0), C.c_int64(0)
    k.lib.tdr_profile_score_ms(C.byref(tot_ms), C.byref(launches))
    k.lib.tdr_profile_enable(0)
    # the same steps with the other noise source, beside the headline (a fifth of the steps, at least 5).  The steps beside the
    # headline score through a context of their own: a context tunes its span from its 31st call on (ten trial calls), which in a
    # short run would fall exactly into these few steps; a fresh one stays at the configured span, like a short timed region
    other_steps = max(5, a.steps // 5)
    ctx_main = f.score_ctx
    f.score_ctx = k.score_ctx_create()
    f.parity_rng = not f.parity_rng
    for _ in range(2):
        step()
    barrier()
    t1 = time.perf_counter()
    for _ in range(other_steps):
        step()
    barrier()
    dt_other = (time.perf_counter() - t1) / other_steps
    f.parity_rng = not f.parity_rng
    # ... and with the scan's points arriving in pinned host memory (the PCIe-inclusive step; never `value`)
    scan_src[0] = pts_host
    for _ in range(2):
        step()
    barrier()
    t2 = time.perf_counter()
    for _ in range(other_steps):
        step()
    barrier()
    dt_pcie = (time.perf_counter() - t2) / other_steps
    scan_src[0] = pts_dev
    f.score_ctx = ctx_main
    if world > 1:
        t = torch.tensor([dt], device=k.device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # which loop variant the dense kernel's waves run: two more steps with the device counters on (they cost an atomic per
    # wave-sector, so never inside the timed region)
    vstats = (C.c_int64 * 16)()
    k.lib.tdr_profile_enable(2)
    for _ in range(2):
        step()
    barrier()
    k.lib.tdr_profile_variants(vstats)
    k.lib.tdr_profile_enable(0)

    shares = None
    if rank == 0 and cfg.polar:
        # the launch's two kernels on their own: one more scoring call on the particle set every timed step starts from
        try:
            f.st[:, :nl].copy_(st0)   # the particle set every timed step starts from
            if f.locality_every and f.perm is not None:
                k.locality_order(f.st, nl, m.rows, m.cols, f.perm)
            k.lib.tdr_profile_enable(1)
            k.score(m.dev, r.last_scan()[1], float(cfg.res), f.fp_c, f.st, nl, f.raw_w,
                    perm=f.perm if f.locality_every else None, uniform_scale=f._uniform_scale, n_total=n_global,
                    ctx=k.score_ctx_create())   # (a context like the filter's: the table's factors, the configured span)
            d_ms, s_ms, s_n = C.c_double(0), C.c_double(0), C.c_int64(0)
            if k.lib.tdr_profile_shares(C.byref(d_ms), C.byref(s_ms), C.byref(s_n)) == 0:
                shares = {"dense_ms": d_ms.value, "dense_particles": nl - int(s_n.value), "dense_kernel": "score_polar_su_kernel",
                          "scattered_ms": s_ms.value, "scattered_particles": int(s_n.value),
                          "scattered_kernel": "score_polar_ray_kernel",
                          "how": "HIP events around each kernel, one after the other on one stream, on the particle set "
                                 "every timed step starts from"}
        finally:
            k.lib.tdr_profile_enable(0)
    if rank == 0:
        P = cfg.nb * cfg.nr
        b_pu = P * (4 * cfg.ncls + 1) + 64
        kname = "score_cart"   # score_cart_su_kernel + score_cart_ray_kernel (tdr_score_cart.hip), or the float form score_cart_kernel
        if cfg.polar:
            kname = "score_polar"   # score_polar_su_kernel + score_polar_ray_kernel (the integer form), or score_polar_kernel (the float form)
        n_local = per_gpu
        avg_ms = tot_ms.value / max(1, launches.value)
        alg_gbps = (b_pu * n_local) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, issue_rec, traffic_note = measured_traffic(cfg.name, kname, n_local)
        # ONE roofline fraction: what the counters saw the launch request from the fabric, over the duration measured here
        achieved = traffic / (avg_ms * 1e-3) / 1e9 if traffic is not None and avg_ms > 0 else None
        frac = achieved / 8000.0 if achieved is not None else None
        # the unit the launch keeps busiest: memory system (fraction of the HBM peak), vector issue, the LDS arrays, the L1 address path
        util = {"hbm": frac or 0.0}
        issue, bound_unit_frac = None, None
        if issue_rec:
            issue = {"valu_busy": issue_rec.get("valu_busy"), "lds_busy": issue_rec.get("lds_busy"),
                     "l1_addr_busy": issue_rec.get("l1_addr_busy"),
                     "insts_per_sample": issue_rec.get("insts_per_sample"),
                     "vmem_rd_insts_per_launch": issue_rec.get("vmem_rd_insts"), "cycles_per_launch": issue_rec.get("cycles"),
                     "how": "rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE, --pmc "
                            "TA_BUSY_avr and --pmc SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE over this command "
                            "(tools/traffic_from_pmc.py): issue cycles of the vector "
                            "units at four per instruction (an upper bound: tools/valu_cost.hip measures 2.4 - 4.2) / cycles "
                            "of the LDS arrays / busy cycles of the texture addressers (the L1 address path) over the "
                            "kernels' cycles, vector instructions per wave per window sample"}
            util["valu"] = issue["valu_busy"] or 0.0
            util["lds"] = issue["lds_busy"] or 0.0
            util["l1_addr"] = issue["l1_addr_busy"] or 0.0
            bound_unit_frac = issue_rec.get("bound_unit_frac")
        bound = max(util, key=util.get)
        # a work rate beside the dense one: bytes a launch that shares nothing between particles could not do without
        nnz = int((r.last_images().sum(dim=0) > 0).sum().item())
        sparse_bytes = n_local * (P // 8 + 2 * nnz)
        sparse = {"bytes_per_launch": sparse_bytes, "nonempty_bins": nnz, "bins": P,
                  "GBps": sparse_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
                  "x_hbm_peak": sparse_bytes / (avg_ms * 1e-3) / 1e9 / 8000.0 if avg_ms > 0 else None,
                  "traffic_over_sparse": traffic / sparse_bytes if traffic is not None else None,
                  "what": "per particle one mask bit per window sample + 2 bytes (a class-plane cell) per non-empty scan "
                          "bin, nothing shared between particles: a work rate, not bytes moved"}
        out = {
            "metric": "particle-updates/sec (render+score+resample)",
            "value": n_global * a.steps / dt,
            "unit": "particle-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "toolchain": __import__("top_down_renderer_amd.build", fromlist=["toolchain"]).toolchain(),
            "rng": {"timed": ("device counter-based noise (Philox)" if a.device_rng else
                              "the reference's std::mt19937 + std::normal_distribution stream, reproduced on the device "
                              "(bit-identical propagate; csrc/tdr_rng.hip)"),
                    "other": "std::mt19937 stream" if a.device_rng else "device counter-based noise (Philox)",
                    "other_ms_per_step": dt_other * 1e3, "other_steps": other_steps},
            "pcie_inclusive": {"ms_per_step": dt_pcie * 1e3, "value": n_global / dt_pcie, "steps": other_steps,
                               "what": "the same step with the scan's points in pinned host memory, read over PCIe by the "
                                       "raster kernel itself (how the C++ boundary hands a scan over); `value` has every "
                                       "input resident in HBM"},
            "config": {"workload": f"{cfg.name}: {cfg.n_pts}-pt scan, {cfg.ncls} classes, {cfg.nb}x{cfg.nr} "
                                   f"{'polar' if cfg.polar else 'Cartesian'} render, {cfg.map_size}x{cfg.map_size} map",
                       "particles_per_gpu": per_gpu, "particles_total": n_global,
                       "particle_distribution": ("90% Gaussian (30 px, 10 deg) about the true pose + 10% uniform"
                                                 if cfg.have_init else "8 Gaussian clusters (40 px) on road cells"),
                       "locality_every": a.locality_every, "parallelism": f"particles sharded over {world} GPU(s)"},
            "roofline": {"bound": bound, "kernel": kname, "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": frac, "traffic": traffic,
                         "basis": "counter traffic: bytes the L2s request from the fabric per launch (TCC_EA0_RDREQ, a counter "
                                  "pass of its own over this command) over the launch duration measured in this run",
                         "bound_unit_frac": bound_unit_frac,
                         "bound_unit_frac_is": "vector-memory wave-instructions per launch x 16 address-path cycles (the floor "
                                               "of a 64-lane gather) / (256 CUs x the launch's cycles)",
                         "traffic_source": (f"replayed from profiles/score_traffic.json (kernel sources {kernel_source_hash()}): "
                                            + traffic_note) if traffic is not None else "none (" + traffic_note + ")",
                         "sparse_algorithmic": sparse, "shares": shares,
                         "traffic_is": "read requests the L2s send to the fabric (TCC_EA0_RDREQ), Infinity-Cache hits "
                                       "included: an upper bound on the bytes HBM itself delivers",
                         "issue": issue,
                         "note": "`achieved`, `peak`, `frac` price the launch's counter traffic against the HBM roofline "
                                 "whatever `bound` says; `bound` names the unit the launch keeps busiest (hbm = `frac`, "
                                 "valu / lds / l1_addr = `issue`).  "
                                 "A polar launch runs two kernels one after the other — score_polar_su_kernel for the dense "
                                 "particles, score_polar_ray_kernel for the scattered ones, both bound by the L1 address path "
                                 "(cache lines per gather) — and `avg_launch_ms` spans both (DESIGN.md 5.1)",
                         "variants": variant_shares(list(vstats), cfg.polar),
                         "avg_launch_ms": avg_ms, "launches": launches.value,
                         # polar configs: the dense / scattered split of the mixed launch this filter's tuner settled on
                         "tuner": ({"span_cells": f.score_ctx.span(), "trials_in_timed_region": trials_timed,
                                    "timed_launches": launches.value} if cfg.polar else None),
                         "dense_work_rate": {"bytes_per_launch": b_pu * n_local, "GBps": alg_gbps,
                                             "x_hbm_peak": alg_gbps / 8000.0,
                                             "what": "SURVEY 8(d) dense byte model over the launch duration: a work rate, "
                                                     "not bytes moved"}},
        }
        if world > 1:
            out["rccl_ranks" if a.backend == "nccl" else a.backend + "_ranks"] = ranks_joined
        if init_step_ms is not None:
            out["config"]["init_search_first_step_ms"] = init_step_ms
            out["config"]["init_search_first_step_cold_ms"] = init_cold_ms
            out["config"]["init_search_particle_updates_per_s"] = n_global / (init_step_ms * 1e-3)
        if not a.no_cpu and a.cpu_sample != 0 and world == 1:
            # ~12 s of CPU work: the oracle does ~880 config-2 particle-updates/s per host thread (measured on the GPU box);
            # a window sample costs about the same in every config
            ns = a.cpu_sample if a.cpu_sample > 0 else max(256, int(12 * 880 * host_threads() * 1.64e6 / b_pu))
            out["cpu_baseline"] = cpu_baseline(sc, cfg, min(ns, n_global), host_threads())
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
