#!/usr/bin/env python3
"""Subdomain-count sweep SURVEY 8(d) asks for ({8, 64, 512} + the tuned size), one JSON line per
run: solve to tol 1e-5 with preAlps_ECGSolve on one MI355X, cubic boxes (or the library's graph
partitioner with --kway).  usage: nparts_sweep.py poisson|elasticity n t [--kway | --shuffle] [--no8]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
from prealps_amd import gen
from prealps_amd.solver import partition_kway

wl, n, t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kway = "--kway" in sys.argv or "--shuffle" in sys.argv
rp, ci, v = gen.poisson3d_csr(n) if wl == "poisson" else gen.elasticity3d_csr(n)
shuffle = "--shuffle" in sys.argv          # a random symmetric permutation: nothing left of the grid numbering
if shuffle:
    import scipy.sparse as sp
    N = len(rp) - 1
    pm = np.random.default_rng(3).permutation(N)
    if "--nodes" in sys.argv and wl == "elasticity":      # renumber the nodes, keep a node's 3 dofs together (what FE codes write)
        pn = np.random.default_rng(3).permutation(N // 3)
        pm = (3 * pn[:, None] + np.arange(3)[None, :]).ravel()
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))[pm][:, pm]
    A.sort_indices()
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    kway = True
tuned = (5, 5, 10) if wl == "poisson" else (2, 4, 8)
cases = [("tuned", tuned)] + [(str(k ** 3), (-(-n // k),) * 3) for k in (8, 4, 2)]
if "--no8" in sys.argv:
    cases = cases[:-1]
for name, box in cases:
    part, P = (gen.box_partition(n, box) if wl == "poisson" else gen.box_partition_nodes(n, box))
    tp = 0.0
    if kway:
        t0 = time.time(); part = partition_kway(rp, ci, P); tp = time.time() - t0
    t0 = time.time()
    prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
    prob.create_block_jacobi()
    setup = time.time() - t0
    r = prob.solve(prob.reference_rhs(), t, max_iter=5000)
    print(json.dumps({"workload": "%s %d^3" % (wl, n), "t": t, "nparts": int(P), "partition": (("library k-way, nodes randomly renumbered" if "--nodes" in sys.argv else "library k-way, rows randomly renumbered") if shuffle else "library k-way") if kway else "boxes %s" % (box,),
                      "iterations": int(r.iters), "solve_seconds": r.seconds, "iterations_per_s": r.iters / r.seconds,
                      "setup_seconds": setup, "partition_seconds": tp, "factor_GB": prob.stat("bj_factor_bytes") / 1e9,
                      "sparse_factor_blocks": int(prob.stat("bj_nd_blocks")), "bj_max_bandwidth": int(prob.stat("bj_max_bandwidth")),
                      "final_res_over_normb": r.final_res / r.normb}), flush=True)
    prob.close()
