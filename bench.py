#!/usr/bin/env python3
"""bench.py -- ECG iterations/s on MI355X (+ SpMM roofline, + CPU baseline).

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under
torch.distributed.run, one rank per GPU) prints ONE JSON line on rank 0.

A step = one full ECG iteration of the reference driver loop
(examples/test_ecg_prealps_op.c:208-221): Iterate(rci 0) -> stopping test ->
block-Jacobi apply -> Iterate(rci 1) -> SpMM, through the C ABI of
libprealps_hip.so.  Default workload = what BASELINE.json's metric is quoted on:
3-D elasticity, n ~ 1M dofs (Q1 hexahedra on 70^3 nodes, N = 1,029,000,
nnz = 80,990,208), t = 4, block-Jacobi, fp64, inputs resident in HBM before the
timed region.  `--workload poisson` runs BASELINE configs[1] (7-pt Poisson 100^3).  If the solve converges inside the timed region it
is restarted from the same rhs (the restart is inside the timing).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", type=str, default="elasticity", choices=["poisson", "elasticity"])
    ap.add_argument("--n", type=int, default=0, help="grid points (nodes) per side; default 100 (poisson) / 70 (elasticity)")
    ap.add_argument("--t", type=int, default=4, help="enlarging factor")
    ap.add_argument("--box", type=str, default="", help="subdomain box in nodes; default 5,5,10 (poisson) / 2,4,8 (elasticity)")
    ap.add_argument("--alg", type=str, default="odir", choices=["odir", "omin", "fused"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--spmm-reps", type=int, default=50)
    return ap.parse_args()


def run_iterations(prob, e, rhs, L, nsteps, state):
    """Advance the reference driver loop by nsteps full iterations (the loop itself is C:
    preAlps_ECGAdvance, the same calls examples/test_ecg_prealps_op.c:208-221 makes)."""
    from prealps_amd.lib import check
    restarts, last_it, last_res = C.c_int(0), C.c_int(state["last_iters"]), C.c_double(state["last_res"])
    check(L.preAlps_ECGAdvance(C.byref(e), rhs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(state["rci"]),
                               nsteps, C.byref(restarts), C.byref(last_it), C.byref(last_res)), "ECGAdvance")
    state["restarts"] += restarts.value
    state["last_iters"], state["last_res"] = last_it.value, last_res.value


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        N, wname = 3 * a.n ** 3, "Q1 3-D elasticity, %d^3 nodes, nu=0.25, stiff/soft inclusions (SURVEY A.3 structure)" % a.n
    nnz = len(val)
    t_setup = time.perf_counter()
    prob = prealps_amd.EcgProblem(rowptr, colind, val, nparts, part, scale=True, device=local_rank,
                                  distributed=distributed)
    L = prob.L
    prob.create_block_jacobi()
    check(L.preAlps_hip_prepare_operator(a.t), "prepare_operator")   # the SpMM plan is part of the setup
    t_setup = time.perf_counter() - t_setup
    rhs = prob.reference_rhs()
    alg = {"odir": pl.ORTHODIR, "omin": pl.ORTHOMIN, "fused": pl.ORTHODIR_FUSED}[a.alg]
    if alg == pl.ORTHODIR_FUSED:
        raise SystemExit("bench.py times the two-phase RCI loop; use --alg odir|omin")
    e = prob.new_ecg(a.t, alg, pl.NO_BS_RED, 1e-5, 100000)
    rci = C.c_int(0)
    check(L.preAlps_ECGInitialize(C.byref(e), rhs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(rci)), "ECGInitialize")
    check(L.preAlps_BlockJacobiApply(e.R, e.P), "BlockJacobiApply")
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
    t0 = time.perf_counter()
    run_iterations(prob, e, rhs, L, a.steps, state)
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    its = a.steps / dt

    # ---- dominant kernel (SpMM) against the HBM roofline: HIP events on the library stream
    ts = L.preAlps_hip_panel_stride(a.t)
    m_loc, nnz_loc = int(prob.stat("rows_local")), int(prob.stat("nnz_local"))
    halo = int(prob.stat("halo_rows"))
    sec = C.c_double()
    # (a) in context: between two SpMM launches of the solver lies a whole iteration that streams the
    #     block-Jacobi factors and seven panels through the 256 MiB Infinity Cache, so each timed launch
    #     here is preceded by one (untimed) preconditioner apply; one event pair per launch.
    tot = 0.0
    for _ in range(a.spmm_reps):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "BlockJacobiApply")
        check(L.preAlps_hip_timer_start(), "timer_start")
        check(L.preAlps_BlockOperator(e.P, e.AP), "BlockOperator")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
        tot += sec.value
    spmm_s = tot / a.spmm_reps
    # (b) back to back (matrix partly resident in the Infinity Cache): reported for comparison only
    check(L.preAlps_hip_timer_start(), "timer_start")
    for _ in range(a.spmm_reps):
        check(L.preAlps_BlockOperator(e.P, e.AP), "BlockOperator")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
    spmm_b2b_s = sec.value / a.spmm_reps
    # 12 B per nonzero + 4 B per row pointer + read X + write AX (8*t B per row each)
    spmm_bytes = 12.0 * nnz_loc + 4.0 * (m_loc + 1) + 8.0 * (m_loc + halo) * a.t + 8.0 * m_loc * a.t
    spmm_gbs = spmm_bytes / spmm_s / 1e9
    # block-Jacobi apply, same stopwatch
    check(L.preAlps_hip_timer_start(), "timer_start")
    for _ in range(a.spmm_reps):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "BlockJacobiApply")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "timer_stop")
    bj_s = sec.value / a.spmm_reps
    bj_bytes = prob.stat("bj_factor_bytes") + 16.0 * m_loc * a.t + 8.0 * m_loc
    # streaming ceilings of this very device (calibration kernels of the library, 1 GiB buffers)
    copy_gbs, read_gbs = C.c_double(), C.c_double()
    check(L.preAlps_hip_hbm_probe(1 << 30, 20, C.byref(copy_gbs), C.byref(read_gbs)), "hbm_probe")
    # HBM bytes per launch from the rocprofv3 PMC passes of this same command (profiles/): the
    # counters cannot be read from inside the process, so the committed summary is quoted when
    # the workload is the one it was collected on.
    traffic, traffic_src = None, None
    profiled = {("elasticity", 70, 4, "2,4,8"): "r01_pmc_hbm_traffic_elasticity.json",
                ("poisson", 100, 4, "5,5,10"): "r01_pmc_hbm_traffic_poisson.json"}
    pmc = profiled.get((a.workload, a.n, a.t, a.box))
    if world == 1 and pmc and os.path.exists(os.path.join(ROOT, "profiles", pmc)):
        with open(os.path.join(ROOT, "profiles", pmc)) as f:
            traffic = json.load(f)["k_spmm"]["traffic_bytes_per_launch"]
        traffic_src = "profiles/%s (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)" % pmc
    out = {
        "metric": "ECG iters/sec + SpMM HBM GB/s (% roofline), 3D-elasticity n~1M t=4",
        "value": its, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s (N=%d, nnz=%d), ECG %s + block-Jacobi, t=%d, tol 1e-5" % (wname, N, nnz, a.alg, a.t),
                   "nparts": int(nparts), "subdomain_box": list(box), "parallelism": "rows x%d" % world,
                   "comm": prob.comm_kind,
                   "restarts_in_timed_region": state["restarts"],
                   "iterations_to_converge": state["last_iters"], "setup_seconds": t_setup,
                   "setup_breakdown_s": {k: prob.stat("setup_" + k + "_s") for k in ("build", "plan", "bj_factor", "bj_layout")},
                   "bj_max_bandwidth": int(prob.stat("bj_max_bandwidth")),
                   "spmm_blocks": int(prob.stat("spmm_blocks"))},
        "roofline": {"kernel": "k_spmm_runs" if prob.stat("spmm_runs") else ("k_spmm_staged" if prob.stat("spmm_staged") else "k_spmm"), "bound": "hbm", "achieved": spmm_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": spmm_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": spmm_bytes, "avg_launch_us": 1e6 * spmm_s,
                     "back_to_back_launch_us": 1e6 * spmm_b2b_s,
                     "measured_copy_ceiling_GBs": copy_gbs.value, "measured_read_ceiling_GBs": read_gbs.value,
                     "frac_of_measured_read_ceiling": spmm_gbs / read_gbs.value,
                     "note": "each timed launch follows one preconditioner apply (cache state of the solver loop)"},
        "block_jacobi": {"avg_apply_us": 1e6 * bj_s, "factor_bytes": prob.stat("bj_factor_bytes"),
                         "achieved_GBs": bj_bytes / bj_s / 1e9},
    }

    # ---- CPU baseline on the host cores, rank 0, N=1.  Two ports of the same algorithm are timed on
    #      a bounded sample and the FASTER one is reported: (a) oracle/ecg_oracle.c, plain C + OpenMP,
    #      one thread per group of subdomains (the reference's one-rank-per-core layout); (b) the
    #      reference's own kernels, mkl_dcsrmm + MKL PARDISO + BLAS, threaded inside one process.
    if rank == 0 and world == 1 and not a.no_cpu:
        from oracle import oracle as O
        from oracle import mkl_path as M
        import scipy.sparse as sp
        A = sp.csr_matrix((val, colind.astype(np.int32), rowptr), shape=(N, N))
        B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, nparts)
        rhs_cpu = O.reference_rhs(rowpos)
        tf0 = time.perf_counter()
        ecg = O.ECG(B, rowpos, a.t, O.ORTHODIR if a.alg == "odir" else O.ORTHOMIN, O.NO_BS_RED, 1e-5, a.cpu_iters)
        tfac = time.perf_counter() - tf0
        r = ecg.solve(rhs_cpu)
        cands = [(r["iters"] / r["t_total"], O.lib().orc_num_threads(),
                  "C/OpenMP port (oracle/ecg_oracle.c): %d iterations, operator %.3fs precond %.3fs of %.3fs, "
                  "factorisation %.1fs outside the rate" % (r["iters"], r["t_op"], r["t_prec"], r["t_total"], tfac))]
        if a.alg == "odir" and M.load_mkl() is not None:
            try:
                e_cpu = M.MklEcg(B, rowpos, a.t, 1e-5, a.cpu_iters, threads=min(os.cpu_count() or 1, 128))
                rm = e_cpu.solve(rhs_cpu)
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
                               "host_cores_online": os.cpu_count()}
    if rank == 0:
        print(json.dumps(out))
    prob.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
