#!/usr/bin/env python3
"""Does the SpMM time drift with how long the GPU has been busy?  Builds the bench
workload once and prints the in-context SpMM and block-solve times every few seconds."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
import prealps_amd.lib as pl
from prealps_amd import gen
from prealps_amd.lib import check

n, t = 70, 4
rowptr, colind, val = gen.elasticity3d_csr(n)
part, nparts = gen.box_partition_nodes(n, (2, 4, 8))
prob = prealps_amd.EcgProblem(rowptr, colind, val, nparts, part, scale=True, device=0)
L = prob.L
prob.create_block_jacobi()
rhs = prob.reference_rhs()
e = prob.new_ecg(t, pl.ORTHODIR, pl.NO_BS_RED, 1e-5, 100000)
rci = C.c_int(0)
check(L.preAlps_ECGInitialize(C.byref(e), rhs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(rci)), "init")
check(L.preAlps_BlockJacobiApply(e.R, e.P), "bj")
check(L.preAlps_BlockOperator(e.P, e.AP), "op")
sec = C.c_double()
t0 = time.perf_counter()
reps = int(os.environ.get("PROBE_REPS", "20"))
for rep in range(reps):
    tot = 0.0
    for _ in range(50):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "bj")
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(e.P, e.AP), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        tot += sec.value
    check(L.preAlps_hip_timer_start(), "ts")
    for _ in range(50):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "bj")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
    print("t=%6.1fs spmm %.1f us  bj %.1f us" % (time.perf_counter() - t0, 1e6 * tot / 50, 1e6 * sec.value / 50), flush=True)
    idle = float(os.environ.get("PROBE_IDLE", "0"))
    if idle:
        time.sleep(idle)
prob.close()
