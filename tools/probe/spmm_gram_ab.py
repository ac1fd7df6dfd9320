#!/usr/bin/env python3
"""Same process, alternating: the SpMM of the headline problem with and without the Gram block
(k_spmm_runs_gram vs k_spmm_runs<4, 4>), each launch behind one block solve (the solver's cache state)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check
n, t = 70, 4
rp, ci, v = gen.elasticity3d_csr(n); part, P = gen.box_partition_nodes(n, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
L = prob.L
lib = C.CDLL(os.path.join(os.path.dirname(prealps_amd.__file__), "libprealps_hip.so"))
lib.pa_rt_malloc.restype = C.c_void_p; lib.pa_rt_malloc.argtypes = [C.c_size_t]
lib.pa_k_spmm_gram_arm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
prob.create_block_jacobi()
check(L.preAlps_hip_prepare_operator(t), "prep")
X = np.random.default_rng(0).standard_normal((prob.m, t))
dx, dy, dr = prob.panel(t, t), prob.panel(t, t), prob.panel(t, t)
prob.to_device(dx, X, t); prob.to_device(dr, X[::-1].copy(), t)
nblk = int(prob.stat("spmm_blocks"))
parts = lib.pa_rt_malloc((nblk + 64) * 32 * 8)
px, py, pr = (C.cast(d.val, C.c_void_p) for d in (dx, dy, dr))
sec = C.c_double()
def time_spmm(armed, reps=20):
    tot = 0.0
    for i in range(reps + 3):
        check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
        if armed: lib.pa_k_spmm_gram_arm(px, py, pr, parts, nblk)
        else: lib.pa_k_spmm_gram_disarm(None)
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 3: tot += sec.value
    return 1e6 * tot / reps
for rnd in range(4):
    a, b = time_spmm(False), time_spmm(True)
    print("round %d: plain %.1f us, with the Gram block %.1f us (+%.1f)" % (rnd, a, b, b - a), flush=True)
lib.pa_k_spmm_gram_disarm(None)
prob.close()
