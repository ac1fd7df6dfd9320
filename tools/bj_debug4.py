#!/usr/bin/env python3
"""Block solve and ECG on one box partition, several panel widths, against the oracle."""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
from prealps_amd import gen
from oracle import oracle as O
n, box = int(sys.argv[1]), tuple(int(x) for x in sys.argv[2].split(","))
rp, ci, v = gen.poisson3d_csr(n); part, P = gen.box_partition(n, box)
N = len(rp) - 1
A = sp.csr_matrix((v, ci, rp), shape=(N, N))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
prob.create_block_jacobi()
B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
print("env", {k: os.environ[k] for k in os.environ if k.startswith("PREALPS_")}, "band", prob.stat("bj_max_bandwidth"), "g4 bytes", prob.stat("bj_g4_bytes"))
for t in (4, 8):
    X = np.random.default_rng(t).standard_normal((N, t))
    zr = O.BlockJacobi(B, rowpos).apply(X)
    got = prob.block_jacobi_apply(X, t)
    e = np.abs(got - zr).max(axis=0) / np.abs(zr).max()
    print(" t=%d block solve: max err per column %s" % (t, " ".join("%.1e" % x for x in e)))
    rhs = prob.reference_rhs()
    g = prob.solve(rhs, t)
    r = O.ECG(B, rowpos, t).solve(rhs)
    k = min(len(g.res), len(r["res"]))
    rel = np.abs(g.res[:k] - r["res"][:k]) / r["res"][:k]
    print(" t=%d ECG: iters %d / %d, max rel diff of the history %.2e (at %d)" % (t, g.iters, r["iters"], rel.max(), rel.argmax()))
prob.close()
