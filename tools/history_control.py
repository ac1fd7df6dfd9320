#!/usr/bin/env python3
"""CPU-vs-CPU control of the residual-history drift on the elasticity matrices (no GPU): the same
ECG recurrence in two fp64 implementations -- the C/OpenMP oracle (oracle/ecg_oracle.c) and the
reference's own kernels (mkl_dcsrmm + PARDISO + BLAS, oracle/mkl_path.py) -- on the same matrix,
partition and rhs.  If their residual histories separate like the GPU-vs-oracle ones recorded in
profiles/ (tools/history_probe.py), the drift is a property of the recurrence on this matrix
(coefficient jumps of 1e10), not of the HIP path.
usage: history_control.py [nodes_per_side=30] [iterations=300]   -> one line per case on stdout
       history_control.py --dodir [nodes_per_side=30] [t=8]     -> when do two CPU paths reduce their directions?
       history_control.py --tail                                -> the rank-deficient end of the t = 8 Poisson case"""
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


def dodir_control(n=30, t=8, maxit=1000, out=sys.stdout):
    """D-Odir (-o 0 -r 1, src/solvers/ecg.c:445-497): the iteration at which the block size drops is a threshold on
    the singular values of a t x t block of a history that is chaotic on this matrix.  Two CPU paths -- oracle and
    MKL kernels -- from the same rhs: block-size sequences, the iterations of each reduction, the residual gap."""
    rp, ci, v = gen.elasticity3d_csr(n)
    part, P = gen.box_partition_nodes(n, (2, 4, 8))
    N = len(rp) - 1
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
    rhs = O.reference_rhs(rowpos)
    a = O.ECG(B, rowpos, t, O.ORTHODIR, O.ADAPT_BS, 1e-5, maxit).solve(rhs)
    b = M.MklEcg(B, rowpos, t, 1e-5, maxit).solve_dodir(rhs)

    def drops(bs):
        return [(int(i) + 1, int(bs[i])) for i in range(1, len(bs)) if bs[i] < bs[i - 1]]
    da, db = drops(a["bs"]), drops(b["bs"])
    k = min(len(a["res"]), len(b["res"]))
    rel = np.abs(a["res"][:k] - b["res"][:k]) / a["res"][:k]
    first = min(da[0][0] if da else k, db[0][0] if db else k)
    print("D-Odir %d^3 boxes 2x4x8 t=%d: iterations oracle %d mkl %d" % (n, t, a["iters"], b["iters"]), file=out)
    print("  reductions (iteration, new block size): oracle %s" % da, file=out)
    print("                                           mkl    %s" % db, file=out)
    print("  same sequence: %s; residual gap at iteration 20 / 100 / first reduction (%d): %.1e / %.1e / %.1e" % (
        list(a["bs"]) == list(b["bs"]), first, rel[min(19, k - 1)], rel[min(99, k - 1)], rel[min(first - 1, k - 1)]), file=out, flush=True)
    return dict(drops=(da, db), iters=(a["iters"], b["iters"]), rel=rel)


def tail_control(out=sys.stdout):
    """The t = 8 case of tests/test_gpu_configs.py (Poisson 16^3 on 16 slabs, Orthodir): 8 directions on 16 slabs
    lose rank as the solve converges; how far apart are two CPU paths at the last residuals?"""
    A = O.poisson3d(16)
    part = O.contiguous_partition(16 ** 3, 16)
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, 16)
    rhs = O.reference_rhs(rowpos)
    a = O.ECG(B, rowpos, 8).solve(rhs)
    b = M.MklEcg(B, rowpos, 8, 1e-5, 1000).solve(rhs)
    k = min(len(a["res"]), len(b["res"]))
    rel = np.abs(a["res"][:k] - b["res"][:k]) / a["res"][:k]
    print("Poisson 16^3 slabs t=8: iterations oracle %d mkl %d; rel diff of the residuals: all but the last two %.1e, last two %s" % (
        a["iters"], b["iters"], rel[:-2].max(), " ".join("%.1e" % x for x in rel[-2:])), file=out, flush=True)
    return dict(iters=(a["iters"], b["iters"]), rel=rel)


if __name__ == "__main__":
    if M.load_mkl() is None:
        raise SystemExit("libmkl_rt is not on this host")
    if len(sys.argv) > 1 and sys.argv[1] == "--dodir":
        dodir_control(int(sys.argv[2]) if len(sys.argv) > 2 else 30, int(sys.argv[3]) if len(sys.argv) > 3 else 8)
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--tail":
        tail_control()
        raise SystemExit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    it = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    for c in cases(n, it):
        control(*c)
