#!/usr/bin/env python3
"""One process, alternating environments: absolute times of 800-iteration solves of the headline problem (the
library reads these switches at solver reset).  usage: r4_solve_env_ab.py "A=1,B=0" "A=0" ...   (R4_AB_T: columns)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import prealps_amd
from prealps_amd import gen
t = int(os.environ.get("R4_AB_T", "4"))
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
prob.solve(rhs, t, tol=1e-30, max_iter=50)
cfgs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
keys = sorted({k for c in cfgs for k in c})
for rnd in range(3):
    for c in cfgs:
        for k in keys: os.environ.pop(k, None)
        os.environ.update(c)
        r = prob.solve(rhs, t, tol=1e-30, max_iter=800)
        print("round %d %-40s %d iterations, %.1f us per iteration" % (rnd, c, r.iters, 1e6 * r.seconds / r.iters), flush=True)
prob.close()
