#!/usr/bin/env python3
"""D-Odir (dynamic reduction of the search directions, -o 0 -r 1) at full size: does the HIP path drop the
same directions at the same iterations as the CPU oracle?  Elasticity n^3 nodes, 2x4x8-node subdomains,
t = 8; both run `iters` iterations from the same rhs; prints the block-size sequences around every change.
usage (GPU box): dodir_fullsize_check.py [n=70] [iters=420]"""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd as pa
from prealps_amd import gen
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 420
t = 8
rp, ci, v = gen.elasticity3d_csr(n)
part, P = gen.box_partition_nodes(n, (2, 4, 8))
N = len(rp) - 1
A = sp.csr_matrix((v, ci, rp), shape=(N, N))
prob = pa.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
rhs = prob.reference_rhs()
t0 = time.time()
got = prob.solve(rhs, t, ortho_alg=pa.ORTHODIR, bs_red=pa.ADAPT_BS, tol=1e-5, max_iter=iters)
tg = time.time() - t0
B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
t0 = time.time()
ref = O.ECG(B, rowpos, t, O.ORTHODIR, O.ADAPT_BS, 1e-5, iters).solve(rhs)
tc = time.time() - t0
k = min(len(got.bs), len(ref["bs"]))
chg_g = [(i + 1, int(got.bs[i])) for i in range(1, len(got.bs)) if got.bs[i] != got.bs[i - 1]]
chg_c = [(i + 1, int(ref["bs"][i])) for i in range(1, len(ref["bs"])) if ref["bs"][i] != ref["bs"][i - 1]]
rel = np.abs(got.res[:k] - ref["res"][:k]) / ref["res"][:k]
first = chg_c[0][0] if chg_c else k
print("elasticity %d^3 (N = %d), %d subdomains, t = %d, D-Odir, %d iterations: HIP %.1fs, oracle %.1fs" % (n, N, P, t, k, tg, tc))
print("  block-size changes (iteration, new size)  HIP   : %s" % chg_g)
print("  block-size changes (iteration, new size)  oracle: %s" % chg_c)
print("  same sequence over the %d iterations compared: %s" % (k, bool(np.array_equal(got.bs[:k], ref["bs"][:k]))))
print("  rel. diff of the residual norm at 1, 20, 100, just before the first reduction (%d), end: %s" % (
    first, " ".join("%.1e" % rel[min(i, k - 1)] for i in (0, 19, 99, max(first - 2, 0), k - 1))))
print("  res/normb at the end: HIP %.3e oracle %.3e" % (got.res[k - 1] / got.normb, ref["res"][k - 1] / ref["normb"]))
prob.close()
