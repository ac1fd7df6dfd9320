#!/usr/bin/env python3
"""Forward sweep of the block solve alone (PREALPS_BJ_G4_EARLY=3) against scipy: first wrong row per block."""
import os, sys
os.environ["PREALPS_BJ_G4_EARLY"] = os.environ.get("PREALPS_BJ_G4_EARLY", "3")
import numpy as np, scipy.sparse as sp, scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
from prealps_amd import gen
from oracle import oracle as O
n, box, t = int(sys.argv[1]), tuple(int(x) for x in sys.argv[2].split(",")), 4
rp, ci, v = gen.poisson3d_csr(n); part, P = gen.box_partition(n, box)
N = len(rp) - 1
A = sp.csr_matrix((v, ci, rp), shape=(N, N))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
prob.create_block_jacobi()
B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
X = np.random.default_rng(1).standard_normal((N, t))
for rep in range(3):
    got = prob.block_jacobi_apply(X, t)
    out = []
    for p in range(min(P, 8)):
        r0, r1 = rowpos[p], rowpos[p + 1]
        Ab = B[r0:r1][:, r0:r1].toarray()
        Lc = np.linalg.cholesky(Ab)
        ref = sl.solve_triangular(Lc, X[r0:r1], lower=True) / np.diag(Lc)[:, None]
        e = np.abs(got[r0:r1] - ref).max(axis=1) / np.abs(ref).max()
        bad = np.where(e > 1e-11)[0]
        out.append("blk%d:%s" % (p, ("ok" if len(bad) == 0 else "first bad row %d (%d bad, max %.1e)" % (bad[0], len(bad), e.max()))))
    print("rep", rep, "; ".join(out))
prob.close()
