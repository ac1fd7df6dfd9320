#!/usr/bin/env python3
"""Block solve at full size against its definition: blockdiag(A)^-1 (blockdiag(A) X) = X, and timing.
usage: bj_check.py [elasticity|poisson] n box t [kway-parts]"""
import ctypes as C, os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
from prealps_amd import gen
from oracle import oracle as O

wl, n, box, t = sys.argv[1], int(sys.argv[2]), tuple(int(x) for x in sys.argv[3].split(",")), int(sys.argv[4])
kway = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if wl == "poisson":
    rp, ci, v = gen.poisson3d_csr(n); part, P = gen.box_partition(n, box)
else:
    rp, ci, v = gen.elasticity3d_csr(n); part, P = gen.box_partition_nodes(n, box)
if kway:
    from prealps_amd.solver import partition_kway
    part, P = partition_kway(rp, ci, kway), kway
N = len(rp) - 1
A = sp.csr_matrix((v, ci, rp), shape=(N, N))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
prob.create_block_jacobi()
B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
pid = np.repeat(np.arange(P), np.diff(rowpos))
coo = B.tocoo()
keep = pid[coo.row] == pid[coo.col]
D = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=B.shape)
X = np.random.default_rng(1).standard_normal((N, t))
back = prob.block_jacobi_apply(D @ X, t)
err = np.abs(back - X).max() / np.abs(X).max()
bad = np.unique(pid[np.where(np.abs(back - X).max(axis=1) > 1e-8)[0]])
print("%s n=%d box=%s t=%d parts=%d env=%s: max error %.3e, %d blocks off (first %s), band %d, si bytes %.0f MB" % (
    wl, n, box, t, P, {k: os.environ[k] for k in os.environ if k.startswith("PREALPS_")}, err, len(bad), bad[:8],
    prob.stat("bj_max_bandwidth"), prob.stat("bj_g4_bytes") / 1e6), flush=True)
prob.close()
