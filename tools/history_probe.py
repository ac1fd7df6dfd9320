#!/usr/bin/env python3
"""How far the HIP path and the CPU oracle stay together over a whole solve (run on the GPU box):
prints the relative difference of the residual norm along the iterations for the elasticity
matrices, where coefficient jumps of 1e10 amplify rounding differences."""
import sys, os
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd as pa
from prealps_amd import gen
from prealps_amd.solver import partition_kway
from oracle import oracle as O

def run(name, rp, ci, v, part, P, t, maxit):
    N = len(rp) - 1
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))
    prob = pa.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
    rhs = prob.reference_rhs()
    got = prob.solve(rhs, t, max_iter=maxit)
    ref = O.ECG(B, rowpos, t, O.ORTHODIR, O.NO_BS_RED, 1e-5, maxit).solve(rhs)
    k = min(len(got.res), len(ref["res"]))
    rel = np.abs(got.res[:k] - ref["res"][:k]) / ref["res"][:k]
    print(name, "iters gpu %d cpu %d; rel diff at 1,5,10,20,40,80,end:" % (got.iters, ref["iters"]),
          " ".join("%.1e" % rel[min(i, k - 1)] for i in (0, 4, 9, 19, 39, 79, k - 1)),
          "| res/normb at end gpu %.3e cpu %.3e" % (got.res[-1] / got.normb, ref["res"][-1] / ref["normb"]), flush=True)
    prob.close()

rp, ci, v = gen.elasticity3d_csr((12, 10, 10))
run("12x10x10 kway P=8 t=4", rp, ci, v, partition_kway(rp, ci, 8), 8, 4, 1000)
rp, ci, v = gen.elasticity3d_csr(9)
part, P = gen.box_partition_nodes(9, (3, 3, 3))
run("9^3 boxes P=27 t=4", rp, ci, v, part, P, 4, 400)
run("9^3 boxes P=27 t=8", rp, ci, v, part, P, 8, 400)
if len(sys.argv) > 1:
    n = int(sys.argv[1])
    rp, ci, v = gen.elasticity3d_csr(n)
    part, P = gen.box_partition_nodes(n, (2, 4, 8))
    run("%d^3 boxes 2x4x8 t=4" % n, rp, ci, v, part, P, 4, int(sys.argv[2]))
