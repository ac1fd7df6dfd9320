#!/usr/bin/env python3
"""Which rows of which blocks does the block solve get wrong?  usage: bj_debug2.py [poisson|elasticity] n box t"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
from prealps_amd import gen
from oracle import oracle as O
wl, n, box, t = sys.argv[1], int(sys.argv[2]), tuple(int(x) for x in sys.argv[3].split(",")), int(sys.argv[4])
if wl == "poisson":
    rp, ci, v = gen.poisson3d_csr(n); part, P = gen.box_partition(n, box)
else:
    rp, ci, v = gen.elasticity3d_csr(n); part, P = gen.box_partition_nodes(n, box)
N = len(rp) - 1
A = sp.csr_matrix((v, ci, rp), shape=(N, N))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
prob.create_block_jacobi()
B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
X = np.random.default_rng(1).standard_normal((N, t))
ref = O.BlockJacobi(B, rowpos).apply(X)
got = prob.block_jacobi_apply(X, t)
err = np.abs(got - ref).max(axis=1) / np.abs(ref).max()
print("env", {k: os.environ[k] for k in os.environ if k.startswith("PREALPS_")}, "band", prob.stat("bj_max_bandwidth"), "max err %.3e" % err.max())
for p in range(min(P, 6)):
    e = err[rowpos[p]:rowpos[p + 1]]
    bad = np.where(e > 1e-10)[0]
    print(" block %d rows %d: max %.2e, %d bad rows, first %s last %s" % (p, len(e), e.max(), len(bad), bad[:12], bad[-4:]))
# unit impulses in one block: which outputs differ
p = 0
b = rowpos[1] - rowpos[0]
for j in (0, 1, 4, 5, b - 1):
    X = np.zeros((N, t)); X[rowpos[p] + j, 0] = 1.0
    ref = O.BlockJacobi(B, rowpos).apply(X); got = prob.block_jacobi_apply(X, t)
    d = np.abs(got - ref)[rowpos[p]:rowpos[p + 1], 0]
    bad = np.where(d > 1e-12 * max(1.0, np.abs(ref).max()))[0]
    print(" impulse at row %d: %d rows differ, first %s; got[%d]=%.6g ref=%.6g" % (j, len(bad), bad[:10], j, got[rowpos[p] + j, 0], ref[rowpos[p] + j, 0]))
prob.close()
