#!/usr/bin/env python3
"""Do the SpMM / block-solve times depend on where the operator's arrays land in HBM?  Builds
the bench workload several times in one process (optionally with a dummy allocation of varying
size in between) and prints the in-context kernel times of each instance."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import prealps_amd
import prealps_amd.lib as pl
from prealps_amd import gen
from prealps_amd.lib import check

n, t = 70, 4
rowptr, colind, val = gen.elasticity3d_csr(n)
part, nparts = gen.box_partition_nodes(n, (2, 4, 8))
keep = []
for inst in range(int(os.environ.get("PROBE_INSTANCES", "5"))):
    pad_mb = int(os.environ.get("PROBE_PAD_MB", "0")) * inst
    if pad_mb:
        keep.append(torch.empty(pad_mb << 20, dtype=torch.uint8, device="cuda"))
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
    tot = 0.0
    for _ in range(60):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "bj")
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(e.P, e.AP), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        tot += sec.value
    check(L.preAlps_hip_timer_start(), "ts")
    for _ in range(60):
        check(L.preAlps_BlockJacobiApply(e.AP, e.Z), "bj")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
    print("instance %d (pad %d MB): spmm %.1f us  bj %.1f us" % (inst, pad_mb, 1e6 * tot / 60, 1e6 * sec.value / 60), flush=True)
    sol = (C.c_double * prob.m)()
    check(L.preAlps_ECGFinalize(C.byref(e), sol), "fin")
    prob.close()
