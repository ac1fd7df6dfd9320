"""Synthetic SPD test matrices and geometric partitions, generated from their
definition (no files, no downloads): the 7-point Poisson operator of SURVEY
Appendix A.2 and sub-box partitions of the grid (the stand-in for METIS k-way
on structured problems)."""
import numpy as np


def poisson3d_csr(n):
    """A = T(x)I(x)I + I(x)T(x)I + I(x)I(x)T, T = tridiag(-1, 2, -1); row (i*n+j)*n+k.
    Returns int32 rowptr, int32 colind, float64 val with sorted columns."""
    N = n ** 3
    idx = np.arange(N, dtype=np.int64)
    i, j, k = idx // (n * n), (idx // n) % n, idx % n
    cols = [idx - n * n, idx - n, idx - 1, idx, idx + 1, idx + n, idx + n * n]
    ok = [i > 0, j > 0, k > 0, np.ones(N, bool), k < n - 1, j < n - 1, i < n - 1]
    vals = [-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0]
    C = np.stack(cols, axis=1)
    M = np.stack(ok, axis=1)
    V = np.broadcast_to(np.array(vals), (N, 7))
    counts = M.sum(axis=1)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(counts)
    return rowptr.astype(np.int32), C[M].astype(np.int32), np.ascontiguousarray(V[M], dtype=np.float64)


def box_partition(n, box):
    """Part id of every node of an n^3 grid cut into boxes of `box` = (bi, bj, bk)
    nodes (the last box of a direction may be smaller).  Parts are numbered
    box-lexicographically, so consecutive parts are neighbours."""
    bi, bj, bk = box
    idx = np.arange(n ** 3, dtype=np.int64)
    i, j, k = idx // (n * n), (idx // n) % n, idx % n
    ni, nj, nk = -(-n // bi), -(-n // bj), -(-n // bk)
    part = ((i // bi) * nj + (j // bj)) * nk + (k // bk)
    return part.astype(np.int32), ni * nj * nk
