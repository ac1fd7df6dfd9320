#!/usr/bin/env python3
"""D-Odir (block-size reduction on) at t = 8 on the headline matrix for rocprofv3: 150 iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import prealps_amd
from prealps_amd import gen
import prealps_amd.lib as pl
t = int(os.environ.get("R4_AB_T", "8"))
alg = pl.ORTHOMIN if os.environ.get("R4_ALG") == "omin" else pl.ORTHODIR
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
prob.solve(rhs, t, ortho_alg=alg, bs_red=pl.ADAPT_BS, tol=1e-30, max_iter=10)
r = prob.solve(rhs, t, ortho_alg=alg, bs_red=pl.ADAPT_BS, tol=1e-30, max_iter=150)
print("%d iterations, %.1f us per iteration" % (r.iters, 1e6 * r.seconds / r.iters))
prob.close()
