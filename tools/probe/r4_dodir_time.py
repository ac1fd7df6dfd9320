#!/usr/bin/env python3
"""Per-iteration time of the variants with block-size reduction (configs[3]: t = 8) against plain Orthodir / Orthomin
on the headline matrix, 300 iterations each (no reduction happens that early), one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import prealps_amd
from prealps_amd import gen
import prealps_amd.lib as pl
t = int(os.environ.get("R4_AB_T", "8"))
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
prob.solve(rhs, t, tol=1e-30, max_iter=30)
for rnd in range(2):
    for name, alg, red in (("Odir", pl.ORTHODIR, pl.NO_BS_RED), ("D-Odir", pl.ORTHODIR, pl.ADAPT_BS), ("Omin", pl.ORTHOMIN, pl.NO_BS_RED), ("BF-Omin", pl.ORTHOMIN, pl.ADAPT_BS)):
        r = prob.solve(rhs, t, ortho_alg=alg, bs_red=red, tol=1e-30, max_iter=300)
        print("round %d %-8s %d iterations, %.1f us per iteration, final block size %d" % (rnd, name, r.iters, 1e6 * r.seconds / r.iters, r.bs[-1] if len(r.bs) else -1), flush=True)
prob.close()
