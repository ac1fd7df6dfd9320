#!/usr/bin/env python3
"""DEV: the SpMM variants behind PREALPS_SPMM_OLD (0 new, 1 round 3, 2.. experiments), one process, each launch
behind a block solve."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check
t = 4
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
L = prob.L
prob.create_block_jacobi()
check(L.preAlps_hip_prepare_operator(t), "prep")
X = np.random.default_rng(0).standard_normal((prob.m, t))
dx, dy, dz = (prob.panel(t, t) for _ in range(3))
prob.to_device(dx, X, t)
sec = C.c_double()
def timed(reps=20):
    tot = 0.0
    for i in range(reps + 3):
        check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dz)), "bj")
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 3: tot += sec.value
    return 1e6 * tot / reps
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2", "3", "4", "5", "6", "7"]
for rnd in range(3):
    out = []
    for k in variants:
        os.environ["PREALPS_SPMM_OLD"] = k
        out.append("%s: %.1f" % (k, timed()))
    print("round %d  " % rnd + "  ".join(out), flush=True)
prob.close()
