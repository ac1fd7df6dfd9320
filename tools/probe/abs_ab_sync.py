"""Same process, absolute times of 800-iteration solves: how the host learns the residual norm (event, polled\nword) and HIP-graph replay, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, prealps_amd
from prealps_amd import gen
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
prob.solve(rhs, 4, tol=1e-30, max_iter=50)
def run(name, env=None, graphs=0):
    for k in ("PREALPS_ECG_POLL",): os.environ.pop(k, None)
    if env: os.environ.update(env)
    prob.L.preAlps_hip_graphs(graphs)
    prob.solve(rhs, 4, tol=1e-30, max_iter=100)
    r = prob.solve(rhs, 4, tol=1e-30, max_iter=800)
    print("%-14s 800 its %.4f s (%.1f us/it)" % (name, r.seconds, 1e6 * r.seconds / r.iters), flush=True)
for rnd in range(3):
    run("defaults"); run("poll", {"PREALPS_ECG_POLL": "1"}); run("graphs", None, 1)
prob.close()
