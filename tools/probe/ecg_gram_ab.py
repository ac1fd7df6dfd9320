#!/usr/bin/env python3
"""Same process, alternating: ECG iterations of the headline problem with PREALPS_SPMM_GRAM = 0 / 1
(the switch is read when a solver is created).  Per-iteration time = difference of a 500- and a
200-iteration solve (tol far below reach), so set-up and wrap-up cancel.
NOTE: superseded by abs_ab.py / abs_ab_sync.py -- the first solve after a switch pays for new allocations, which
bends the slope this script computes (it read 11-12 us for a change that absolute 800-iteration times put at 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
n, t = 70, 4
rp, ci, v = gen.elasticity3d_csr(n); part, P = gen.box_partition_nodes(n, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
prob.solve(rhs, t, tol=1e-30, max_iter=50)
def per_iteration(gram):
    os.environ["PREALPS_SPMM_GRAM"] = gram
    a = prob.solve(rhs, t, tol=1e-30, max_iter=200)
    b = prob.solve(rhs, t, tol=1e-30, max_iter=500)
    assert a.iters >= 200 and b.iters >= 500, (a.iters, b.iters)
    return 1e6 * (b.seconds - a.seconds) / (b.iters - a.iters), b.res[100]
for rnd in range(4):
    (t0, r0), (t1, r1) = per_iteration("0"), per_iteration("1")
    print("round %d: %.1f us per iteration without, %.1f us with the Gram block from the SpMM (%+.1f); residual 100: %.6e / %.6e"
          % (rnd, t0, t1, t1 - t0, r0, r1), flush=True)
prob.close()
