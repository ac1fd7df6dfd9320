#!/usr/bin/env python3
"""Spread study: does the SpMM's mode depend on where the PANELS lie?  One process, the headline
operator, the product timed in the solver's cache state on several freshly allocated (X, Y) pairs -- the earlier pairs
stay allocated, so every pair lies on other pages -- and on mixed pairs (old X with new Y and the other way round)."""
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
prob.create_block_jacobi()
check(L.preAlps_hip_prepare_operator(t), "prep")
X = np.random.default_rng(0).standard_normal((prob.m, t))
sec = C.c_double()
def time_spmm(dx, dy, scratch, reps=20):
    tot = 0.0
    for i in range(reps + 5):
        check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(scratch)), "bj")     # the solver's cache state
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 5: tot += sec.value
    return 1e6 * tot / reps
addr = lambda d: C.cast(d.val, C.c_void_p).value
scratch = prob.panel(t, t)
pairs = []
for k in range(int(os.environ.get("PAIRS", "6"))):
    dx, dy = prob.panel(t, t), prob.panel(t, t)
    prob.to_device(dx, X, t)
    pairs.append((dx, dy))
    print("pair %d: spmm %.1f us   x %#x y %#x" % (k, time_spmm(dx, dy, scratch), addr(dx), addr(dy)), flush=True)
for k in range(1, len(pairs)):
    print("x of pair 0, y of pair %d: %.1f us;  x of pair %d, y of pair 0: %.1f us" % (
        k, time_spmm(pairs[0][0], pairs[k][1], scratch), k, time_spmm(pairs[k][0], pairs[0][1], scratch)), flush=True)
print("pair 0 again: %.1f us" % time_spmm(pairs[0][0], pairs[0][1], scratch), flush=True)
prob.close()
