#!/usr/bin/env python3
"""A plain 300-iteration solve of BASELINE configs[1] (7-point Poisson 100^3, 4000 boxes of 5 x 5 x 10) for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import prealps_amd
from prealps_amd import gen
t = int(os.environ.get("R4_AB_T", "4"))
rp, ci, v = gen.poisson3d_csr(100); part, P = gen.box_partition(100, (5, 5, 10))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
prob.solve(rhs, t, tol=1e-30, max_iter=20)
r = prob.solve(rhs, t, tol=1e-30, max_iter=300)
print("%d iterations, %.1f us per iteration" % (r.iters, 1e6 * r.seconds / r.iters))
prob.close()
