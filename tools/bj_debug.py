import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
from oracle import oracle as O

def run(A, P, part, t, tag):
    part = O.contiguous_partition(A.shape[0], P) if part is None else part
    rp, ci, v = O.as_csr(A)
    prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
    X = np.random.default_rng(1).standard_normal((B.shape[0], t))
    got = prob.block_jacobi_apply(X, t)
    ref = O.BlockJacobi(B, rowpos).apply(X)
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max()
    bad = np.where(err > 1e-9)[0]
    sizes = np.diff(rowpos)
    print(tag, "t", t, "w", prob.stat("bj_max_bandwidth"), "parts", P, "sizes", sizes.min(), sizes.max(), "max err %.2e" % err.max(),
          "bad rows", len(bad), "in parts", sorted(set(np.searchsorted(rowpos, bad, side="right") - 1))[:10])
    prob.close()

def tri(n, w):
    d = [np.full(n, 2.0 * w + 1)]
    offs = [0]
    for k in range(1, w + 1):
        d += [np.full(n - k, -1.0 / k)] * 2
        offs += [k, -k]
    return sp.diags(d, offs, format="csr")

for t in (8,):
    run(tri(64, 5), 1, None, t, "band5 b=64")
    run(tri(48, 5), 1, None, t, "band5 b=48")
    run(tri(41, 5), 1, None, t, "band5 b=41")
    run(tri(41, 20), 1, None, t, "band20 b=41")
    run(tri(100, 40), 1, None, t, "band40 b=100")
    run(tri(200, 2), 1, None, t, "band2 b=200")
    run(tri(200, 1), 1, None, t, "band1 b=200")
    run(sp.identity(100, format="csr") * 3.0, 2, None, t, "diag b=50")
    run(tri(300, 70), 2, None, t, "band70 b=150")
    run(tri(300, 100), 1, None, t, "band100 b=300")
