#!/usr/bin/env python3
"""Stress of the sparse block solve on matrices that are no grids: random sparse SPD blocks (unstructured,
some disconnected), a few thousand to tens of thousands of rows each, against scipy's sparse LU of the same
diagonal blocks.  usage: nd_stress.py [rows per block] [blocks] [t]"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PREALPS_BJ_ND", "2")
import prealps_amd as pa

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(7)
N = nb * P
blocks = []
for p in range(P):
    if p % 3 == 2:      # two disconnected halves with different densities
        h = nb // 2
        M = sp.block_diag([sp.random(h, h, density=6.0 / h, random_state=rng), sp.random(nb - h, nb - h, density=14.0 / nb, random_state=rng)], format="csr")
    elif p % 3 == 1:    # a geometric graph: points in the unit cube linked to their nearest neighbours (3-D mesh-like)
        from scipy.spatial import cKDTree
        pts = rng.random((nb, 3))
        d, idx = cKDTree(pts).query(pts, k=9)
        rows = np.repeat(np.arange(nb), 8)
        M = sp.csr_matrix((rng.random(8 * nb), (rows, idx[:, 1:].ravel())), shape=(nb, nb))
    else:               # plain random pattern (expander-like: the worst case for nested dissection)
        M = sp.random(nb, nb, density=5.0 / nb, random_state=rng, format="csr")
    blocks.append(M + M.T)
A = sp.block_diag(blocks, format="lil")
# a few couplings between the blocks
for k in range(200):
    i, j = rng.integers(0, N, 2)
    A[i, j] = A[j, i] = 0.1
A = sp.csr_matrix(A)
A = sp.csr_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 0.5))
A.sort_indices()
part = (np.arange(N) // nb).astype(np.int32)
rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
t0 = time.time()
prob = pa.EcgProblem(rp, ci, v, P, part, scale=False, device=0)
prob.create_block_jacobi()
ts = time.time() - t0
X = rng.standard_normal((N, t))
t0 = time.time()
got = prob.block_jacobi_apply(X, t)
ref = np.zeros_like(X)
for p in range(P):
    sl = slice(p * nb, (p + 1) * nb)
    ref[sl] = spl.splu(sp.csc_matrix(A[sl, sl])).solve(X[sl])
err = np.abs(got - ref).max() / np.abs(ref).max()
print("N = %d in %d blocks, t = %d: sparse-factor blocks %d, factor %.2f GB, setup %.2f s, inverse check %.1e, max error vs splu %.2e"
      % (N, P, t, prob.stat("bj_nd_blocks"), prob.stat("bj_factor_bytes") / 1e9, ts, prob.stat("bj_nd_inverse_dev"), err))
r = prob.solve(prob.reference_rhs(), t, max_iter=500)
print("ECG: %d iterations, res/normb %.2e" % (r.iters, r.final_res / r.normb))
assert err < 1e-9 and r.final_res <= 1e-5 * r.normb
prob.close()
