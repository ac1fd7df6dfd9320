#!/usr/bin/env python3
"""BASELINE configs[1] (7-point Poisson 100^3, boxes of 5 x 5 x 10), one process: the window SpMM plan (round 3's
default for short rows) against the staged plan on round 4's kernel (PREALPS_SPMM_STAGED=1), SpMM launches behind a
block solve and 800-iteration solves."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check
t = 4
rp, ci, v = gen.poisson3d_csr(100); part, P = gen.box_partition(100, (5, 5, 10))
for rnd in range(2):
    for env in ({"PREALPS_SPMM_STAGED": "0"}, {"PREALPS_SPMM_STAGED": "1"}, {}):
        os.environ.pop("PREALPS_SPMM_STAGED", None)
        os.environ.update(env)
        prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
        L = prob.L
        prob.create_block_jacobi()
        check(L.preAlps_hip_prepare_operator(t), "prep")
        X = np.random.default_rng(0).standard_normal((prob.m, t))
        dx, dy, dz = (prob.panel(t, t) for _ in range(3))
        prob.to_device(dx, X, t)
        sec = C.c_double()
        tot = 0.0
        for i in range(23):
            check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dz)), "bj")
            check(L.preAlps_hip_timer_start(), "ts")
            check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
            check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
            if i >= 3: tot += sec.value
        rhs = prob.reference_rhs()
        prob.solve(rhs, t, tol=1e-30, max_iter=50)
        r = prob.solve(rhs, t, tol=1e-30, max_iter=800)
        print("round %d %-28s staged %d runs %d: SpMM %.1f us; solve %.1f us per iteration" % (
            rnd, env, prob.stat("spmm_staged"), prob.stat("spmm_runs"), 1e6 * tot / 20, 1e6 * r.seconds / r.iters), flush=True)
        prob.close()
