#!/usr/bin/env python3
"""Same process, alternating: ECG iterations of the headline problem under two settings of a switch
that the library reads when a solver is created.  usage: ecg_ab.py NAME A B [t]   (NAME = an environment
variable, or GRAPHS for preAlps_hip_graphs(A / B)).  Per-iteration time = difference of a 500- and a
200-iteration solve (tol far below reach), so set-up and wrap-up cancel.
NOTE: superseded by abs_ab.py / abs_ab_sync.py -- the first solve after a switch pays for new allocations, which
bends the slope this script computes (it read 11-12 us for a change that absolute 800-iteration times put at 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
name, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
t = int(sys.argv[4]) if len(sys.argv) > 4 else 4
if os.environ.get("ECG_AB_WORKLOAD") == "poisson":       # BASELINE configs[1]
    n = 100
    rp, ci, v = gen.poisson3d_csr(n); part, P = gen.box_partition(n, (5, 5, 10))
else:
    n = 70
    rp, ci, v = gen.elasticity3d_csr(n); part, P = gen.box_partition_nodes(n, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
prob.solve(rhs, t, tol=1e-30, max_iter=50)
def per_iteration(val):
    if name == "GRAPHS": prob.L.preAlps_hip_graphs(int(val))
    else: os.environ[name] = val
    a = prob.solve(rhs, t, tol=1e-30, max_iter=200)
    b = prob.solve(rhs, t, tol=1e-30, max_iter=500)
    assert a.iters >= 200 and b.iters >= 500, (a.iters, b.iters)
    return 1e6 * (b.seconds - a.seconds) / (b.iters - a.iters)
per_iteration(va); per_iteration(vb)
for rnd in range(4):
    t0, t1 = per_iteration(va), per_iteration(vb)
    print("round %d: %s=%s %.1f us per iteration, %s=%s %.1f us (%+.1f)" % (rnd, name, va, t0, name, vb, t1, t1 - t0), flush=True)
prob.close()
