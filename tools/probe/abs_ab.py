"""Same process: absolute times of 200- and 800-iteration solves (tol far below reach) of the headline problem
(ECG_AB_WORKLOAD=poisson: BASELINE configs[1]) with the round-3 switches on / off one at a time.  The 800-iteration
times repeat to 0.1 us per iteration; slopes between two solves do not (the first solve after a switch pays for
new allocations), which is what made tools/probe/ecg_ab.py over-estimate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, prealps_amd
from prealps_amd import gen
if os.environ.get("ECG_AB_WORKLOAD") == "poisson":       # BASELINE configs[1]
    rp, ci, v = gen.poisson3d_csr(100); part, P = gen.box_partition(100, (5, 5, 10))
else:
    rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
on = {"PREALPS_SPMM_GRAM": "1", "PREALPS_BJ_GRAM": "1", "PREALPS_ECG_LAZY_STOP": "1"}
off = {"PREALPS_SPMM_GRAM": "0", "PREALPS_BJ_GRAM": "0", "PREALPS_ECG_LAZY_STOP": "0"}
prob.solve(rhs, 4, tol=1e-30, max_iter=50)
cfgs = [("defaults    ", None), ("all on      ", on), ("SpMM Gram off", dict(on, PREALPS_SPMM_GRAM="0")), ("BJ Gram off  ", dict(on, PREALPS_BJ_GRAM="0")),
        ("lazy off     ", dict(on, PREALPS_ECG_LAZY_STOP="0")), ("all off      ", off)]
for rnd in range(3):
    for name, env in cfgs:
        for k in on: os.environ.pop(k, None)
        if env: os.environ.update(env)
        out = []
        for it in (200, 800):
            r = prob.solve(rhs, 4, tol=1e-30, max_iter=it)
            out.append((r.iters, r.seconds))
        print(name, "800 its %.4f s (%.1f us/it) | slope 200->800: %.1f us/it" % (out[1][1], 1e6 * out[1][1] / out[1][0], 1e6 * (out[1][1] - out[0][1]) / (out[1][0] - out[0][0])), flush=True)
prob.close()
