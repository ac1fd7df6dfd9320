#!/usr/bin/env python3
"""bench.py -- ECG iterations/s on MI355X (+ SpMM roofline, + CPU baseline, + parity line).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).
With --gpus N > 1 and no launcher environment (RANK unset) the script starts its own N
ranks (`python -m torch.distributed.run --nproc-per-node N ...` as a child process, decided
before anything touches the GPU) and relays their output; under a launcher it checks that
WORLD_SIZE equals --gpus and exits non-zero otherwise.

A step = one full ECG iteration of the reference driver loop
(examples/test_ecg_prealps_op.c:208-221): Iterate(rci 0) -> stopping test ->
block-Jacobi apply -> Iterate(rci 1) -> SpMM, through the C ABI of
libprealps_hip.so.  Default workload = what BASELINE.json's metric is quoted on:
3-D elasticity, n ~ 1M dofs (the reference's Q1 element on 70^3 nodes, N = 1,029,000,
nnz = 80,990,208), t = 4, block-Jacobi, fp64, inputs resident in HBM before the
timed region.  `--workload poisson` runs BASELINE configs[1] (7-pt Poisson 100^3).  If the
solve converges inside the timed region it is restarted from the same rhs (the restart is
inside the timing).  After the timed region: the SpMM against the HBM roofline (HIP events on
the library stream), the per-phase device times of the iteration in the layout of
examples/test_ecg_bench_fused.c:300-336 (stderr + `phases`), and -- rank 0, N = 1 -- the CPU
ports on a bounded sample with the parity line of BASELINE.md section 3 (max relative
difference of the first residuals, GPU vs CPU oracle).
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time


def host_cpu_share():
    """CPUs this process can really use: affinity mask and cgroup CPU quota (a GPU box gives a
    one-GPU job 16 CPUs' worth of quota on a 256-thread host; 128 OpenMP threads then only take
    turns -- the CPU baseline measured 5.9 iterations/s with 128 threads and 24 with 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except (OSError, ValueError):
            pass
    return n


# OpenMP reads this when libgomp is loaded (the CPU oracle of the baseline leg, the library's setup loops).
# torch.distributed.run gives its children OMP_NUM_THREADS=1 unless told otherwise: the ranks of one
# node then set up their shards single-threaded; give each rank its part of the CPU share instead.
_lws = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
if "OMP_NUM_THREADS" not in os.environ or (_lws > 1 and os.environ["OMP_NUM_THREADS"] == "1"):
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(host_cpu_share(), 128) // max(1, _lws)))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
ROUND = "r04"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", type=str, default="elasticity", choices=["poisson", "elasticity"])
    ap.add_argument("--n", type=int, default=0, help="grid points (nodes) per side; default 100 (poisson) / 70 (elasticity)")
    ap.add_argument("--t", type=int, default=4, help="enlarging factor")
    ap.add_argument("--box", type=str, default="", help="subdomain box in nodes; default 5,5,10 (poisson) / 2,4,8 (elasticity)")
    ap.add_argument("--nparts", type=int, default=0,
                    help="number of subdomains from the library's own graph partitioner instead of --box")
    ap.add_argument("--alg", type=str, default="auto", choices=["auto", "odir", "omin", "fused"],
                    help="auto = odir for any number of GPUs.  (The one-collective fused Orthodir of "
                         "examples/test_ecg_bench_fused.c saves an all-reduce per iteration but costs 42-48 us more "
                         "device time per iteration on a 1/8 shard -- profiles/r03_shard8_*.json -- which is more than "
                         "a small all-reduce takes.)")
    ap.add_argument("--reduce", action="store_true", help="dynamic reduction of the search directions (-r 1 of the "
                    "reference driver: D-Odir with --alg odir, BF-Omin with --alg omin), BASELINE configs[3]")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="rehearse ONE rank of a G-GPU run on this one GPU (with --shard r): the rank's rows, plan, "
                         "kernels and stream choreography; sums are local, halo rows arrive as zeros")
    ap.add_argument("--shard", type=int, default=0)
    ap.add_argument("--graphs", action="store_true", help="replay the two halves of an iteration from HIP graphs "
                    "(measured slower than plain launches on ROCm 7.2: off by default)")
    ap.add_argument("--survey-cpu-iters", type=int, default=4,
                    help="CPU baselines at the survey's subdomain count: iterations per port (0 skips them)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--spmm-reps", type=int, default=50)
    ap.add_argument("--phase-iters", type=int, default=20, help="iterations timed phase by phase (hipEvents)")
    ap.add_argument("--survey-nparts", type=int, default=64,
                    help="also time the same problem cut into this many cubic subdomains (SURVEY 8d asks for 64: the "
                         "reference's one large block per rank); 0 skips it")
    return ap.parse_args()


def launch_ranks(a):
    """--gpus N without a launcher: become the launcher.  Nothing has touched the GPU yet."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def run_iterations(prob, e, rhs, L, nsteps, state):
    """Advance the reference driver loop by nsteps full iterations (the loop itself is C:
    preAlps_ECGAdvance, the same calls examples/test_ecg_prealps_op.c:208-221 makes)."""
    from prealps_amd.lib import check
    restarts, last_it, last_res = C.c_int(0), C.c_int(state["last_iters"]), C.c_double(state["last_res"])
    check(L.preAlps_ECGAdvance(C.byref(e), rhs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(state["rci"]),
                               nsteps, C.byref(restarts), C.byref(last_it), C.byref(last_res)), "ECGAdvance")
    state["restarts"] += restarts.value
    state["last_iters"], state["last_res"] = last_it.value, last_res.value


PHASE_KEYS = ("operator", "precond", "gram", "trsm", "update", "small", "comm")


def phase_table(title, ranks, iters, ph, ecg_fields, out=sys.stderr):
    """The report of examples/test_ecg_bench_fused.c:300-336 for one run."""
    print("=== %s ===" % title, file=out)
    print("\t# ranks     : %d" % ranks, file=out)
    print("\titerations  : %d" % iters, file=out)
    print("\ttotal   : %e s" % ph.get("total", float("nan")), file=out)
    print("\toperator: %e s" % ph.get("operator", float("nan")), file=out)
    print("\tprecond : %e s" % ph.get("precond", float("nan")), file=out)
    for k in ("tot_t", "comm_t", "trsm_t", "gemm_t", "potrf_t", "pstrf_t", "lapmt_t", "gesvd_t", "geqrf_t",
              "ormqr_t", "copy_t"):
        if k in ecg_fields:
            print("\t%-8s: %e s" % ({"tot_t": "tot_iter"}.get(k, k[:-2]), ecg_fields[k]), file=out)
    print("", file=out)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))
    if a.shard_of and (a.gpus != 1 or not (0 <= a.shard < a.shard_of)):
        raise SystemExit("--shard-of G needs --gpus 1 and 0 <= --shard < G")
    if a.alg == "auto":
        a.alg = "odir"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    # rehearsal knobs for a one-GPU box: PREALPS_BENCH_BACKEND=gloo PREALPS_BENCH_ONE_DEVICE=1
    backend = os.environ.get("PREALPS_BENCH_BACKEND", "nccl")
    if os.environ.get("PREALPS_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import prealps_amd
    import prealps_amd.lib as pl
    from prealps_amd import gen
    from prealps_amd.lib import check

    if a.n == 0:
        a.n = 100 if a.workload == "poisson" else 70
    if not a.box:
        a.box = "5,5,10" if a.workload == "poisson" else "2,4,8"
    box = tuple(int(x) for x in a.box.split(","))
    if a.workload == "poisson":
        rowptr, colind, val = gen.poisson3d_csr(a.n)
        part, nparts = gen.box_partition(a.n, box)
        N, wname = a.n ** 3, "BASELINE configs[1]: synthetic 7-pt 3-D Poisson SPD CSR %d^3" % a.n
    else:
        rowptr, colind, val = gen.elasticity3d_csr(a.n)
        part, nparts = gen.box_partition_nodes(a.n, box)
        N, wname = 3 * a.n ** 3, ("Q1 3-D elasticity, %d^3 nodes, the reference's element matrix (nu=0.25), stiff/soft "
                                  "inclusions (examples/test_ecg_petsc_ela.c:217-347)" % a.n)
    if a.nparts > 0:
        part, nparts = None, a.nparts      # the library's k-way graph partitioner (operator.c / partition.c)
    nnz = len(val)
    t_setup = time.perf_counter()
    prob = prealps_amd.EcgProblem(rowptr, colind, val, nparts, part, scale=True, device=local_rank,
                                  distributed=distributed, partitioner=(a.nparts > 0),
                                  shard=(a.shard, a.shard_of) if a.shard_of > 1 else None)
    L = prob.L
    L.preAlps_hip_graphs(1 if a.graphs else 0)
    prob.create_block_jacobi()
    check(L.preAlps_hip_prepare_operator(a.t), "prepare_operator")   # the SpMM plan is part of the setup
    t_setup = time.perf_counter() - t_setup
    if distributed:
        # every rank must be on the same binding, and for the nccl backend it has to be the native one
        kinds = [None] * world
        dist.all_gather_object(kinds, prob.comm_kind)
        if len(set(kinds)) != 1:
            raise SystemExit("bench.py: ranks disagree on the communication binding: %s" % kinds)
        if backend == "nccl" and prob.comm_kind != "rccl" and os.environ.get("PREALPS_COMM") != "torch":
            # a multi-GPU number measured through Python callbacks would not be the product's: refuse it
            raise SystemExit("bench.py: the nccl backend is in use but the library's native RCCL binding is not "
                             "(comm = %s); PREALPS_COMM=torch measures the torch.distributed hooks on purpose"
                             % prob.comm_kind)
    rhs = prob.reference_rhs()
    alg = {"odir": pl.ORTHODIR, "omin": pl.ORTHOMIN, "fused": pl.ORTHODIR_FUSED}[a.alg]
    e = prob.new_ecg(a.t, alg, pl.ADAPT_BS if a.reduce else pl.NO_BS_RED, 1e-5, 100000)
    rci = C.c_int(0)
    prhs = rhs.ctypes.data_as(C.POINTER(C.c_double))
    check(L.preAlps_ECGInitialize(C.byref(e), prhs, C.byref(rci)), "ECGInitialize")
    check(L.preAlps_BlockJacobiApply(e.R, e.P), "BlockJacobiApply")
    if alg != pl.ORTHODIR_FUSED:     # (the fused loop starts every iteration with the product, test_ecg_bench_fused.c:252)
        check(L.preAlps_BlockOperator(e.P, e.AP), "BlockOperator")
    state = {"rci": rci, "restarts": 0, "last_iters": 0, "last_res": float("nan")}

    def barrier():
        prob.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    run_iterations(prob, e, rhs, L, a.warmup, state)
    barrier()
    dev_s = C.c_double()
    gram_timed0 = prob.stat("spmm_gram_launches")
    bjg_timed0 = prob.stat("bj_gram_applies")
    check(L.preAlps_hip_timer_start(), "timer_start")      # hipEvents on the library stream around the same region
    t0 = time.perf_counter()
    run_iterations(prob, e, rhs, L, a.steps, state)
    solver_forms_gram = prob.stat("spmm_gram_launches") - gram_timed0 >= a.steps - 1
    solver_bj_forms_gram = prob.stat("bj_gram_applies") - bjg_timed0 >= a.steps - 1
    check(L.preAlps_hip_timer_stop(C.byref(dev_s)), "timer_stop")
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    its = a.steps / dt

    # ---- the same loop phase by phase: hipEvent pairs around every phase (one stream sync each,
    #      so this is a separate, slower pass), accumulated by the library and in preAlps_ECG_t
    phases, ecg_fields = {}, {}
    if a.phase_iters > 0:
        for k in ("tot_t", "comm_t", "trsm_t", "gemm_t", "potrf_t", "copy_t"):
            setattr(e, k, 0.0)
        L.preAlps_hip_timing(1)
        L.preAlps_hip_timing_reset()
        barrier()
        tp0 = time.perf_counter()
        run_iterations(prob, e, rhs, L, a.phase_iters, state)
        barrier()
        phases["total"] = time.perf_counter() - tp0
        L.preAlps_hip_timing(0)
        sec = C.c_double()
        for k in PHASE_KEYS:
            L.preAlps_hip_get_time(k.encode(), C.byref(sec))
            phases[k] = sec.value
        ecg_fields = {k: getattr(e, k) for k in ("tot_t", "comm_t", "trsm_t", "gemm_t", "potrf_t", "pstrf_t",
                                                 "lapmt_t", "gesvd_t", "geqrf_t", "ormqr_t", "copy_t")}
        if distributed:   # MPI_Allreduce(MAX) of the reference's report
            keys = sorted(phases)
            tt = torch.tensor([phases[k] for k in keys], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            phases = dict(zip(keys, [float(v) for v in tt.tolist()]))

    # ---- dominant kernel (SpMM) against the HBM roofline: HIP events on the library stream
    m_loc, nnz_loc = int(prob.stat("rows_local")), int(prob.stat("nnz_local"))
    halo = int(prob.stat("halo_rows"))
    sec = C.c_double()
    # (a) in context: between two SpMM launches of the solver lies a whole iteration that streams the
    #     block-Jacobi factors and seven panels through the 256 MiB Infinity Cache, so each timed launch
    #     here is preceded by one (untimed) preconditioner apply; one event pair per launch.
    tot = 0.0
    gram0 = prob.stat("spmm_gram_launches")
    for _ in range(a.spmm_reps):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "BlockJacobiApply")
        check(L.preAlps_hip_timer_start(), "timer_start")
        check(L.preAlps_BlockOperator(e.P, e.AP), "BlockOperator")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
        tot += sec.value
    spmm_s = tot / a.spmm_reps
    # The product P -> AP of the solver also leaves the Gram block [AP | R]^T P behind (k_spmm_runs_gram, the
    # default at t = 4): it reads the m x t rows of R on top of the product's bytes.  The request is made by the
    # solver's own loop (preAlps_ECGIterate), so the stand-alone launches above are the PLAIN product; the
    # solver's launch is timed where it happens -- the hipEvent pair of the library's `operator` phase, one pair
    # per iteration of the phase-by-phase pass, same stream, same cache state by construction.
    with_gram = bool(solver_forms_gram) and phases.get("operator", 0.0) > 0.0 and a.phase_iters > 0
    plain_s = None
    dx, dy = prob.panel(a.t, a.t), prob.panel(a.t, a.t)
    prob.to_device(dx, np.random.default_rng(1).standard_normal((m_loc, a.t)), a.t)
    if with_gram:
        plain_s = spmm_s
        spmm_s = phases["operator"] / a.phase_iters
    # (b) back to back (matrix partly resident in the Infinity Cache): reported for comparison only
    check(L.preAlps_hip_timer_start(), "timer_start")
    for _ in range(a.spmm_reps):
        check(L.preAlps_BlockOperator(e.P, e.AP), "BlockOperator")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
    spmm_b2b_s = sec.value / a.spmm_reps
    # 12 B per nonzero + 4 B per row pointer + read X + write AX (8*t B per row each)
    spmm_bytes = 12.0 * nnz_loc + 4.0 * (m_loc + 1) + 8.0 * (m_loc + halo) * a.t + 8.0 * m_loc * a.t
    plain_bytes = spmm_bytes
    if with_gram:
        spmm_bytes += 8.0 * m_loc * a.t + 256.0 * prob.stat("spmm_blocks")     # rows of R read, one 8 x 4 block per workgroup written
    spmm_gbs = spmm_bytes / spmm_s / 1e9
    # block-Jacobi apply, same stopwatch
    bjg0 = prob.stat("bj_gram_applies")
    check(L.preAlps_hip_timer_start(), "timer_start")
    for _ in range(a.spmm_reps):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "BlockJacobiApply")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
    bj_solver_s = sec.value / a.spmm_reps
    bj_with_gram = prob.stat("bj_gram_applies") > bjg0
    # the solver's own apply AP -> Z also forms [AP | AP_prev]^T Z (requested by its loop, not by these stand-alone
    # calls): timed where it happens, the hipEvent pair of the library's `precond` phase
    if solver_bj_forms_gram and a.phase_iters > 0 and phases.get("precond", 0.0) > 0.0:
        bj_solver_s = phases["precond"] / a.phase_iters
        bj_with_gram = True
    # the block solve alone (other panels): this is the kernel the roofline figures below are about
    check(L.preAlps_hip_timer_start(), "timer_start")
    for _ in range(a.spmm_reps):
        check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "BlockJacobiApply")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
    bj_s = sec.value / a.spmm_reps
    prob.panel_free(dx); prob.panel_free(dy)
    bj_bytes = prob.stat("bj_factor_bytes") + 16.0 * m_loc * a.t + 8.0 * m_loc
    # streaming ceilings of this very device (calibration kernels of the library, 1 GiB buffers)
    copy_gbs, read_gbs = C.c_double(), C.c_double()
    check(L.preAlps_hip_hbm_probe(1 << 30, 20, C.byref(copy_gbs), C.byref(read_gbs)), "hbm_probe")
    # HBM bytes per launch from the rocprofv3 PMC passes of this same command (profiles/): the
    # counters cannot be read from inside the process, so the committed summary is quoted when
    # the workload is the one it was collected on.
    traffic, traffic_src, bj_traffic = None, None, None
    profiled = {("elasticity", 70, 4, "2,4,8"): "pmc_hbm_traffic_elasticity.json",
                ("poisson", 100, 4, "5,5,10"): "pmc_hbm_traffic_poisson.json"}
    pmc = profiled.get((a.workload, a.n, a.t, a.box)) if a.nparts == 0 else None
    if world == 1 and pmc:
        for rnd in (ROUND, "r03", "r01"):
            path = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, pmc))
            if os.path.exists(path):
                with open(path) as f:
                    pmc_doc = json.load(f)
                traffic = pmc_doc["k_spmm"]["traffic_bytes_per_launch"]
                bj_traffic = (pmc_doc.get("k_bj") or {}).get("traffic_bytes_per_launch")
                traffic_src = "profiles/%s_%s (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)" % (rnd, pmc)
                break
    halo_rows = [halo]
    if distributed:
        halo_rows = [None] * world
        dist.all_gather_object(halo_rows, halo)
    per_it = {k: 1e6 * v / a.phase_iters for k, v in phases.items()} if phases else {}
    out = {
        "metric": "ECG iters/sec + SpMM HBM GB/s (% roofline), 3D-elasticity n~1M t=4",
        "value": its, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "device_ms_per_step": 1e3 * dev_s.value / a.steps,
        "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s (N=%d, nnz=%d), ECG %s%s + block-Jacobi, t=%d, tol 1e-5" % (wname, N, nnz, a.alg, " with direction reduction (-r 1)" if a.reduce else "", a.t),
                   "nparts": int(nparts), "partition": "library k-way" if a.nparts > 0 else "boxes of %s nodes" % (list(box),),
                   "parallelism": "rows x%d" % world,
                   "comm": prob.comm_kind, "hip_graphs": bool(a.graphs), "halo_rows_per_rank": halo_rows,
                   "restarts_in_timed_region": state["restarts"],
                   "iterations_to_converge": state["last_iters"], "setup_seconds": t_setup,
                   "setup_breakdown_s": {k: prob.stat("setup_" + k + "_s") for k in ("build", "plan", "bj_factor", "bj_layout")},
                   "bj_max_bandwidth": int(prob.stat("bj_max_bandwidth")),
                   "spmm_blocks": int(prob.stat("spmm_blocks"))},
        "roofline": {"kernel": ("k_spmm_runs_gram<3>" if prob.stat("spmm_runs") else ("k_spmm_runs_gram<1>" if prob.stat("spmm_staged") else "k_spmm_gram")) if with_gram else ("k_spmm_runs<.,.,3>" if prob.stat("spmm_runs") else ("k_spmm_runs<.,.,1>" if prob.stat("spmm_staged") else "k_spmm")), "bound": "hbm", "achieved": spmm_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": spmm_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": spmm_bytes, "avg_launch_us": 1e6 * spmm_s,
                     "back_to_back_launch_us": 1e6 * spmm_b2b_s,
                     "measured_copy_ceiling_GBs": copy_gbs.value, "measured_read_ceiling_GBs": read_gbs.value,
                     # what the kernel really moves (PMC traffic of the committed profile) over this launch's time, against
                     # what a plain read kernel streams on this very device: the honest distance to the ceiling.  (The
                     # algorithmic bytes count 12 B per nonzero where the kernel reads 8.67, so they may exceed it.)
                     "traffic_GBs": None if traffic is None else traffic / spmm_s / 1e9,
                     "traffic_frac_of_measured_read_ceiling": None if traffic is None else traffic / spmm_s / 1e9 / read_gbs.value,
                     "algorithmic_over_measured_read_ceiling": spmm_gbs / read_gbs.value,
                     "plain_product": None if plain_s is None else {
                         "kernel": "k_spmm_runs" if (prob.stat("spmm_runs") or prob.stat("spmm_staged")) else "k_spmm", "avg_launch_us": 1e6 * plain_s, "algorithmic_bytes_per_launch": plain_bytes,
                         "achieved": plain_bytes / plain_s / 1e9, "frac": plain_bytes / plain_s / 1e9 / HBM_PEAK_GBS,
                         "note": "the same product without the Gram block, stand-alone launches behind one block solve each (what rounds 1-2 reported)"},
                     "note": ("the solver's own launches: hipEvent pair of the `operator` phase, one per iteration of the phase-by-phase pass; "
                              "the product also forms [AP | R]^T P (rows of R counted in the bytes; `traffic` is the PMC figure of this kernel)"
                              if with_gram else "each timed launch follows one preconditioner apply (cache state of the solver loop)")},
        "block_jacobi": {"avg_apply_us": 1e6 * bj_s, "solver_apply_us": 1e6 * bj_solver_s, "solver_apply_forms_gram_block": bool(bj_with_gram),
                         "factor_bytes": prob.stat("bj_factor_bytes"),
                         "traffic": bj_traffic, "frac": bj_bytes / bj_s / 1e9 / HBM_PEAK_GBS,
                         "achieved_GBs": bj_bytes / bj_s / 1e9,
                         # panels of up to 4 columns: ONE stored copy of the band (bj_g4.hip), streamed once by each of
                         # the two sweeps; else the paired copy of both sweeps' records, or the plain two copies
                         "streamed_record_bytes": (2.0 * prob.stat("bj_g4_bytes") if a.t <= 4 and prob.stat("bj_g4_bytes") > 0 else
                                                   prob.stat("bj_pairs_bytes") if a.t <= 4 and prob.stat("bj_pairs_bytes") > 0
                                                   else prob.stat("bj_factor_bytes")),
                         "stored_record_bytes": (prob.stat("bj_g4_bytes") if a.t <= 4 and prob.stat("bj_g4_bytes") > 0 else
                                                 prob.stat("bj_factor_bytes")),
                         "streamed_GBs": ((2.0 * prob.stat("bj_g4_bytes") if a.t <= 4 and prob.stat("bj_g4_bytes") > 0 else
                                           prob.stat("bj_factor_bytes")) + 16.0 * m_loc * a.t) / bj_s / 1e9,
                         "note": "achieved_GBs counts the plain two-sweep factor of SURVEY 8(d) (algorithmic bytes) + the "
                                 "panels; streamed_GBs what the kernel really reads and writes; avg_apply_us = the block solve on "
                                 "panels of its own, solver_apply_us = the solver's AP -> Z, which also reads AP again and AP_prev "
                                 "for the Gram block beta (16 B x t per row more; the PMC traffic averages both kinds of launch)"},
        "phases": {"iterations": a.phase_iters, "device_us_per_iteration": per_it,
                   "ecg_struct_timers_s": ecg_fields,
                   "note": "hipEvent pairs per phase (max over ranks); dense = gram + trsm + update + small; the pass "
                           "syncs after every phase, so its total is above ms_per_step"},
    }
    if a.shard_of > 1:
        kern = sum(per_it.get(k, 0.0) for k in PHASE_KEYS)
        out["scaling"] = "rehearsal"
        out["shard"] = {"of": a.shard_of, "rank": a.shard, "rows": m_loc, "nnz": nnz_loc, "halo_rows": halo,
                        "send_rows": int(prob.stat("send_rows")), "bj_blocks": int(prob.stat("bj_parts_local")),
                        "iteration_device_us": 1e6 * dev_s.value / a.steps,
                        "sum_of_phase_device_us": kern, "launch_gap_us": 1e6 * dev_s.value / a.steps - kern,
                        "halo_overlap": os.environ.get("PREALPS_HALO_OVERLAP", "auto"),
                        "note": "one rank of a %d-GPU run on one GPU (preAlps_hip_loopback): its rows, SpMM plan, pack, "
                                "exchange (main stream in front of one SpMM launch for small interiors, else side stream "
                                "beside the interior blocks), block solve and reductions with the collectives replaced "
                                "by no-ops; sum_of_phase_device_us is measured with a sync after every phase and exceeds "
                                "the iteration; value = iterations/s of THIS shard alone, not a multi-GPU measurement"
                                % a.shard_of}
    if rank == 0 and phases:
        phase_table("%s on %d x MI355X (device time, %d iterations)" % ({"odir": "ODIR", "omin": "OMIN", "fused": "F-ODIR"}[a.alg], world, a.phase_iters),
                    world, a.phase_iters, phases, ecg_fields)

    # ---- CPU baseline on the host cores, rank 0, N=1.  Two ports of the same algorithm are timed on
    #      a bounded sample and the FASTER one is reported: (a) oracle/ecg_oracle.c, plain C + OpenMP,
    #      one thread per group of subdomains (the reference's one-rank-per-core layout); (b) the
    #      reference's own kernels, mkl_dcsrmm + MKL PARDISO + BLAS, threaded inside one process.
    if rank == 0 and world == 1 and not a.no_cpu:
        from oracle import oracle as O
        from oracle import mkl_path as M
        import scipy.sparse as sp
        # the first residuals of a fresh solve on the GPU (same rhs, same partition) for the parity line
        gpu = prob.solve(rhs, a.t, ortho_alg=alg, bs_red=pl.ADAPT_BS if a.reduce else pl.NO_BS_RED, max_iter=a.cpu_iters)
        A = sp.csr_matrix((val, colind.astype(np.int32), rowptr), shape=(N, N))
        B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), prob.part_vector(), nparts)
        rhs_cpu = O.reference_rhs(rowpos)
        tf0 = time.perf_counter()
        ecg = O.ECG(B, rowpos, a.t, {"odir": O.ORTHODIR, "omin": O.ORTHOMIN, "fused": O.ORTHODIR_FUSED}[a.alg],
                    O.ADAPT_BS if a.reduce else O.NO_BS_RED, 1e-5, a.cpu_iters)
        tfac = time.perf_counter() - tf0
        r = ecg.solve(rhs_cpu)
        k = min(len(gpu.res), len(r["res"]))
        rel = np.abs(gpu.res[:k] - r["res"][:k]) / np.abs(r["res"][:k])
        out["parity"] = {"max_rel_diff_res": float(rel.max()) if k else None, "iterations_compared": int(k),
                         "normb_rel_diff": abs(gpu.normb - r["normb"]) / r["normb"],
                         "gpu_res": [float(x) for x in gpu.res[:k]], "cpu_res": [float(x) for x in r["res"][:k]],
                         "note": "residual norm after each of the first iterations, HIP path vs oracle/ecg_oracle.c "
                                 "on the same matrix, partition and rhs (BASELINE.md section 3)"}
        phase_table("ODIR on the host CPU, C/OpenMP port (%d threads)" % O.lib().orc_num_threads(), 1, r["iters"],
                    {"total": r["t_total"], "operator": r["t_op"], "precond": r["t_prec"]}, {})
        cands = [(r["iters"] / r["t_total"], O.lib().orc_num_threads(),
                  "C/OpenMP port (oracle/ecg_oracle.c): %d iterations, operator %.3fs precond %.3fs of %.3fs, "
                  "factorisation %.1fs outside the rate" % (r["iters"], r["t_op"], r["t_prec"], r["t_total"], tfac))]
        if a.alg == "odir" and M.load_mkl() is not None:
            try:
                e_cpu = M.MklEcg(B, rowpos, a.t, 1e-5, a.cpu_iters, threads=min(host_cpu_share(), 128))
                rm = e_cpu.solve(rhs_cpu)
                phase_table("ODIR on the host CPU, MKL kernels (%d threads)" % int(rm["threads"]), 1, rm["iters"],
                            {"total": rm["t_total"], "operator": rm["t_op"], "precond": rm["t_prec"]},
                            {"gemm_t": rm["t_dense"]})
                cands.append((rm["iters"] / rm["t_total"], int(rm["threads"]),
                              "MKL kernels (mkl_dcsrmm %.3fs, PARDISO solves %.3fs, BLAS dense %.3fs of %.3fs for %d "
                              "iterations, PARDISO factorisation %.1fs outside the rate)"
                              % (rm["t_op"], rm["t_prec"], rm["t_dense"], rm["t_total"], rm["iters"], e_cpu.t_factor)))
            except Exception as ex:  # the baseline must never take the GPU result down with it
                cands.append((0.0, 0, "MKL path failed: %s" % ex))
        else:
            cands.append((0.0, 0, "libmkl_rt not on this host"))
        best = max(cands, key=lambda c: c[0])
        out["cpu_baseline"] = {"value": best[0], "unit": "iterations/s", "cores": best[1], "kind": "port",
                               "sample": "same workload (matrix, scaling, partition, rhs), %d ECG iterations each; reported = the "
                                         "faster of: [%s] = %.2f it/s; [%s] = %.2f it/s"
                                         % (a.cpu_iters, cands[0][2], cands[0][0], cands[1][2], cands[1][0]),
                               "host_cores_online": os.cpu_count(), "host_cpu_share": host_cpu_share()}
    prob.close()
    # ---- the same problem with the subdomain count SURVEY 8(d) names (64): large blocks, sparse
    #      nested-dissection factors (nd.c) instead of bands.  Iterations/s there and at the tuned
    #      subdomain size are different quantities (fewer, more expensive iterations), so this goes
    #      beside the headline value, not into it.
    if a.survey_nparts > 0 and world == 1 and a.nparts == 0:
        k = max(1, round(a.survey_nparts ** (1.0 / 3.0)))
        if k ** 3 == a.survey_nparts:
            edge = -(-a.n // k)
            if a.workload == "poisson":
                part2, np2 = gen.box_partition(a.n, (edge, edge, edge))
            else:
                part2, np2 = gen.box_partition_nodes(a.n, (edge, edge, edge))
            t1 = time.perf_counter()
            prob2 = prealps_amd.EcgProblem(rowptr, colind, val, np2, part2, scale=True, device=local_rank)
            prob2.create_block_jacobi()
            check(L.preAlps_hip_prepare_operator(a.t), "prepare_operator")
            setup2 = time.perf_counter() - t1
            rhs2 = prob2.reference_rhs()
            e2 = prob2.new_ecg(a.t, alg, pl.NO_BS_RED, 1e-5, 100000)
            rci2 = C.c_int(0)
            p2 = rhs2.ctypes.data_as(C.POINTER(C.c_double))
            check(L.preAlps_ECGInitialize(C.byref(e2), p2, C.byref(rci2)), "ECGInitialize")
            check(L.preAlps_BlockJacobiApply(e2.R, e2.P), "BlockJacobiApply")
            if alg != pl.ORTHODIR_FUSED:
                check(L.preAlps_BlockOperator(e2.P, e2.AP), "BlockOperator")
            st2 = {"rci": rci2, "restarts": 0, "last_iters": 0, "last_res": float("nan")}
            n2 = max(20, a.steps // 5)
            run_iterations(prob2, e2, rhs2, L, 3, st2)
            prob2.sync()
            t1 = time.perf_counter()
            run_iterations(prob2, e2, rhs2, L, n2, st2)
            prob2.sync()
            dt2 = time.perf_counter() - t1
            check(L.preAlps_hip_timer_start(), "timer_start")
            for _ in range(10):
                check(L.preAlps_BlockJacobiApply(e2.AP, e2.Z), "BlockJacobiApply")
            check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
            apply_s = sec.value / 10
            fbytes = prob2.stat("bj_factor_bytes")
            # fetched bytes per apply from the committed two-pass PMC run of this configuration
            fetched, fsrc = None, None
            for rnd in (ROUND, "r03", "r02"):
                path = os.path.join(ROOT, "profiles", "%s_nd_apply_pmc.txt" % rnd)
                if a.workload == "elasticity" and a.n == 70 and a.t == 4 and os.path.exists(path):
                    import re
                    mm = re.search(r"per apply \(\d+ applies\): ([\d.]+) MB", open(path).read())
                    if mm:
                        fetched = 1e6 * float(mm.group(1))
                        fsrc = "profiles/%s_nd_apply_pmc.txt (2 x FETCH_SIZE of every k_nd_* launch, per apply)" % rnd
                        break
            out["survey_nparts"] = {"nparts": int(np2), "subdomain_box": [edge, edge, edge],
                                    "iterations_per_s": n2 / dt2, "ms_per_step": 1e3 * dt2 / n2, "steps": n2,
                                    "block_solve_us": 1e6 * apply_s, "factor_bytes": fbytes,
                                    "sparse_factor_blocks": int(prob2.stat("bj_nd_blocks")),
                                    "bj_max_bandwidth": int(prob2.stat("bj_max_bandwidth")), "setup_seconds": setup2,
                                    "roofline": {"kernel": "k_nd_forward + k_nd_backward (one launch per tree level)",
                                                 "bound": "hbm", "achieved": fbytes / apply_s / 1e9, "peak": HBM_PEAK_GBS,
                                                 "unit": "GB/s", "frac": fbytes / apply_s / 1e9 / HBM_PEAK_GBS,
                                                 "algorithmic_bytes_per_apply": fbytes, "traffic": fetched,
                                                 "traffic_source": fsrc, "avg_apply_us": 1e6 * apply_s,
                                                 "note": "algorithmic bytes = both stored copies of the panels, each streamed "
                                                         "once per apply"},
                                    "note": "SURVEY 8(d) subdomain count; blocks of this size get the nested-dissection "
                                            "factor (nd.c), not the band kernels"}
            if not a.no_cpu and a.survey_cpu_iters > 0:
                # CPU ports in the reference's own regime (one large sparse-factored block per rank,
                # src/preconditioners/block_jacobi.c:26-63,93-109) + the parity line, same matrix / partition / rhs
                from oracle import oracle as O
                from oracle import mkl_path as M
                import scipy.sparse as sp
                k2 = a.survey_cpu_iters
                gpu2 = prob2.solve(rhs2, a.t, ortho_alg=alg, max_iter=k2)
                A2 = sp.csr_matrix((val, colind.astype(np.int32), rowptr), shape=(N, N))
                B2, perm2, rowpos2 = O.permute_by_part(O.symrac_scale(A2), prob2.part_vector(), np2)
                rhs2_cpu = O.reference_rhs(rowpos2)
                sv = out["survey_nparts"]
                cands2 = []
                tf0 = time.perf_counter()
                ecg2 = O.ECG(B2, rowpos2, a.t, {"odir": O.ORTHODIR, "omin": O.ORTHOMIN, "fused": O.ORTHODIR_FUSED}[a.alg],
                             O.NO_BS_RED, 1e-5, k2)
                tfac2 = time.perf_counter() - tf0
                r2 = ecg2.solve(rhs2_cpu)
                kk = min(len(gpu2.res), len(r2["res"]))
                rel2 = np.abs(gpu2.res[:kk] - r2["res"][:kk]) / np.abs(r2["res"][:kk])
                sv["parity"] = {"max_rel_diff_res": float(rel2.max()) if kk else None, "iterations_compared": int(kk),
                                "gpu_res": [float(x) for x in gpu2.res[:kk]], "cpu_res": [float(x) for x in r2["res"][:kk]]}
                cands2.append((r2["iters"] / r2["t_total"], O.lib().orc_num_threads(),
                               "C/OpenMP port (envelope Cholesky per block): %d iterations, operator %.3fs precond %.3fs of "
                               "%.3fs, factorisation %.1fs outside the rate" % (r2["iters"], r2["t_op"], r2["t_prec"], r2["t_total"], tfac2)))
                del ecg2
                if a.alg == "odir" and M.load_mkl() is not None:
                    try:
                        e_cpu2 = M.MklEcg(B2, rowpos2, a.t, 1e-5, k2, threads=min(host_cpu_share(), 128))
                        rm2 = e_cpu2.solve(rhs2_cpu)
                        cands2.append((rm2["iters"] / rm2["t_total"], int(rm2["threads"]),
                                       "MKL kernels, the reference's configuration (mkl_dcsrmm %.3fs, PARDISO solves %.3fs, BLAS "
                                       "dense %.3fs of %.3fs for %d iterations, PARDISO factorisation %.1fs outside the rate)"
                                       % (rm2["t_op"], rm2["t_prec"], rm2["t_dense"], rm2["t_total"], rm2["iters"], e_cpu2.t_factor)))
                    except Exception as ex:
                        cands2.append((0.0, 0, "MKL path failed: %s" % ex))
                best2 = max(cands2, key=lambda c: c[0])
                sv["cpu_baseline"] = {"value": best2[0], "unit": "iterations/s", "cores": best2[1], "kind": "port",
                                      "sample": "same workload at %d subdomains, %d ECG iterations per port; " % (np2, k2) +
                                                "; ".join("[%s] = %.2f it/s" % (c[2], c[0]) for c in cands2)}
            prob2.close()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
