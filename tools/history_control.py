#!/usr/bin/env python3
"""CPU-vs-CPU control of the residual-history drift on the elasticity matrices (no GPU): the same
ECG recurrence in two fp64 implementations -- the C/OpenMP oracle (oracle/ecg_oracle.c) and the
reference's own kernels (mkl_dcsrmm + PARDISO + BLAS, oracle/mkl_path.py) -- on the same matrix,
partition and rhs.  If their residual histories separate like the GPU-vs-oracle ones recorded in
profiles/ (tools/history_probe.py), the drift is a property of the recurrence on this matrix
(coefficient jumps of 1e10), not of the HIP path.
usage: history_control.py [nodes_per_side=30] [iterations=300]   -> one line per case on stdout"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prealps_amd import gen
from prealps_amd.solver import partition_kway
from oracle import oracle as O
from oracle import mkl_path as M

MARKS = (0, 4, 9, 19, 39, 79)


def control(name, rp, ci, v, part, P, t, maxit, out=sys.stdout):
    N = len(rp) - 1
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
    rhs = O.reference_rhs(rowpos)
    a = O.ECG(B, rowpos, t, O.ORTHODIR, O.NO_BS_RED, 1e-5, maxit).solve(rhs)
    b = M.MklEcg(B, rowpos, t, 1e-5, maxit).solve(rhs)
    k = min(len(a["res"]), len(b["res"]))
    rel = np.abs(a["res"][:k] - b["res"][:k]) / a["res"][:k]
    marks = [min(i, k - 1) for i in MARKS] + [k - 1]
    print(name, "iters oracle %d mkl %d; rel diff at 1,5,10,20,40,80,end:" % (a["iters"], b["iters"]),
          " ".join("%.1e" % rel[i] for i in marks),
          "| res/normb at end oracle %.3e mkl %.3e" % (a["res"][-1] / a["normb"], b["res"][-1] / b["normb"]), file=out, flush=True)
    return dict(iters=(a["iters"], b["iters"]), rel=rel, marks=marks)


def cases(n=30, maxit=300):
    rp, ci, v = gen.elasticity3d_csr((12, 10, 10))
    yield "12x10x10 kway P=8 t=4", rp, ci, v, partition_kway(rp, ci, 8), 8, 4, 1000
    rp, ci, v = gen.elasticity3d_csr(n)
    part, P = gen.box_partition_nodes(n, (2, 4, 8))
    yield "%d^3 boxes 2x4x8 t=4" % n, rp, ci, v, part, P, 4, maxit


if __name__ == "__main__":
    if M.load_mkl() is None:
        raise SystemExit("libmkl_rt is not on this host")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    it = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    for c in cases(n, it):
        control(*c)
