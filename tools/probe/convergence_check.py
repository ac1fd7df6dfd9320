import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
on = {"PREALPS_SPMM_GRAM": "1", "PREALPS_BJ_GRAM": "1", "PREALPS_ECG_LAZY_STOP": "1"}
off = {"PREALPS_SPMM_GRAM": "0", "PREALPS_BJ_GRAM": "0", "PREALPS_ECG_LAZY_STOP": "0"}
for env in (on, off, on, off, on, off):      # (the first solve of a process also pays for first-use allocations)
    os.environ.update(env)
    r = prob.solve(rhs, 4, tol=1e-5, max_iter=3000)
    print(env, "iterations", r.iters, "final res/normb %.3e" % (r.final_res / r.normb), "seconds %.3f" % r.seconds, "gram launches", prob.stat("spmm_gram_launches"), prob.stat("bj_gram_applies"), flush=True)
prob.close()
