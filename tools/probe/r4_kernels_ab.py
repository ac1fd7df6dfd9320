#!/usr/bin/env python3
"""Round-4 kernel timings in ONE process (the only comparison the process-to-process spread allows): the
SpMM and the block solve of the headline problem in the solver's cache state (each timed launch behind the
other kernel), the block solve plain and with the Gram block armed (pa_k_bj_gram_arm, what the solver's
launch does), then 800-iteration solves.  HIP events on the library stream.  (The commit "SpMM: batched X staging ..."
of round 4 still carried the round-3 SpMM behind a dev switch: profiles/r04_spmm_variants_ab.txt was taken there.)
usage: r4_kernels_ab.py [rounds]   (R4_AB_WORKLOAD=poisson: BASELINE configs[1]; R4_AB_T=8: eight columns)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t = int(os.environ.get("R4_AB_T", "4"))
if os.environ.get("R4_AB_WORKLOAD") == "poisson":
    rp, ci, v = gen.poisson3d_csr(100); part, P = gen.box_partition(100, (5, 5, 10))
else:
    rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
L = prob.L
lib = C.CDLL(os.path.join(os.path.dirname(prealps_amd.__file__), "libprealps_hip.so"))
lib.pa_rt_malloc.restype = C.c_void_p; lib.pa_rt_malloc.argtypes = [C.c_size_t]
lib.pa_k_bj_gram_arm.argtypes = [C.c_void_p] * 4 + [C.c_int]
lib.pa_bj_gram_blocks.restype = C.c_int
prob.create_block_jacobi()
check(L.preAlps_hip_prepare_operator(t), "prep")
ts = max(2, 1 << (t - 1).bit_length())
X = np.random.default_rng(0).standard_normal((prob.m, t))
dx, dy, dz, dp = (prob.panel(t, t) for _ in range(4))
prob.to_device(dx, X, t); prob.to_device(dp, X[::-1].copy(), t)
cap = lib.pa_bj_gram_blocks()
parts = lib.pa_rt_malloc((max(cap, 1) + 64) * 32 * 8)
px, pz, pp = (C.cast(d.val, C.c_void_p) for d in (dx, dz, dp))
sec = C.c_double()

def timed(fn, before, reps=20):
    tot = 0.0
    for i in range(reps + 3):
        before()
        check(L.preAlps_hip_timer_start(), "ts")
        fn()
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 3: tot += sec.value
    return 1e6 * tot / reps

def spmm(): check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
def bj(): check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dz)), "bj")
def bj_gram():
    lib.pa_k_bj_gram_arm(px, pz, pp, parts, cap)
    check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dz)), "bj")

for rnd in range(rounds):
    sp_ = timed(spmm, bj)
    a = timed(bj, spmm)
    b = timed(bj_gram, spmm) if (t == 4 and cap > 0) else float("nan")
    print("round %d: SpMM %.1f us | block solve %.1f us, with the Gram block %.1f" % (rnd, sp_, a, b), flush=True)

rhs = prob.reference_rhs()
prob.solve(rhs, t, tol=1e-30, max_iter=50)
for rnd in range(rounds):
    r = prob.solve(rhs, t, tol=1e-30, max_iter=800)
    print("solve: %d iterations, %.1f us per iteration" % (r.iters, 1e6 * r.seconds / r.iters), flush=True)
prob.close()
