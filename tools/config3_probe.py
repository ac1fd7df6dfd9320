#!/usr/bin/env python3
"""BASELINE configs[3] at full size (Q1 elasticity 88^3 nodes, N = 2 044 416, t = 8, dynamic reduction
of the search directions): the first iterations of the HIP path against the CPU oracle -- residuals
and the block-size sequence.  usage: config3_probe.py [nn] [iterations] [odir|omin]"""
import sys, os, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd as pa
from prealps_amd import gen
from oracle import oracle as O

nn = int(sys.argv[1]) if len(sys.argv) > 1 else 88
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 12
alg = sys.argv[3] if len(sys.argv) > 3 else "odir"
rp, ci, v = gen.elasticity3d_csr(nn)
part, P = gen.box_partition_nodes(nn, (2, 4, 8))
N = len(rp) - 1
A = sp.csr_matrix((v, ci, rp), shape=(N, N))
prob = pa.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
rhs = prob.reference_rhs()
a_gpu, a_cpu = (pa.ORTHODIR, O.ORTHODIR) if alg == "odir" else (pa.ORTHOMIN, O.ORTHOMIN)
got = prob.solve(rhs, 8, ortho_alg=a_gpu, bs_red=pa.ADAPT_BS, max_iter=maxit)
t0 = time.time()
ref = O.ECG(B, rowpos, 8, a_cpu, O.ADAPT_BS, 1e-5, maxit).solve(rhs)
print("N = %d, %d parts, %s with reduction, %d iterations (oracle %.1f s)" % (N, P, alg, maxit, time.time() - t0))
k = min(len(got.res), len(ref["res"]))
def changes(b):
    return " ".join("%d@%d" % (b[i], i + 1) for i in range(len(b)) if i == 0 or b[i] != b[i - 1])
print("block size (value@iteration) gpu:", changes(got.bs[:k]))
print("block size (value@iteration) cpu:", changes(ref["bs"][:k]))
rel = np.abs(np.array(got.res[:k]) - np.array(ref["res"][:k])) / np.array(ref["res"][:k])
print("res/normb gpu:", " ".join("%.4e" % (x / got.normb) for x in got.res[:k:max(1, k // 12)]))
print("res/normb cpu:", " ".join("%.4e" % (x / ref["normb"]) for x in ref["res"][:k:max(1, k // 12)]))
print("rel diff of the residual norm along the run:", " ".join("%.1e" % x for x in rel[::max(1, k // 12)]), "max %.2e" % rel.max())
prob.close()
