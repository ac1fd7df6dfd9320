"""Host side of the sparse block solve for large diagonal blocks (prealps_amd/csrc/nd.c): nested
dissection, symbolic structure and multifrontal Cholesky, checked without a GPU through
preAlps_hip_nd_selfcheck (the factor is multiplied back, L L^T x against A x; the triangular
solves themselves are HIP kernels, tested in test_gpu_configs.py).  Stands where the reference
runs PARDISO's analysis + factorisation (src/preconditioners/block_jacobi.c:48-58)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check


def _check(A, leaf):
    L = prealps_amd.load()
    A = sp.csr_matrix(A)
    A.sort_indices()
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data, dtype=np.float64)
    st = np.zeros(8)
    pi, pd = C.POINTER(C.c_int), C.POINTER(C.c_double)
    check(L.preAlps_hip_nd_selfcheck(A.shape[0], rp.ctypes.data_as(pi), ci.ctypes.data_as(pi), v.ctypes.data_as(pd),
                                     leaf, st.ctypes.data_as(pd)), "nd_selfcheck")
    return dict(supernodes=int(st[0]), doubles=st[1], max_front=int(st[2]), height=int(st[3]), resid=st[4], copies=st[5], inverse=st[6], widest=int(st[7]))


@pytest.mark.parametrize("leaf", [16, 96])
def test_poisson_block(leaf):
    n = 14
    rp, ci, v = gen.poisson3d_csr(n)
    A = sp.csr_matrix((v, ci, rp), shape=(n ** 3, n ** 3))
    r = _check(A, leaf)
    assert r["resid"] < 1e-13 and r["copies"] < 1e-14
    assert r["inverse"] < 1e-13               # the selective-inversion panels the solve kernels multiply with
    assert r["supernodes"] > 8 and r["height"] >= 3
    band = n ** 3 * (n * n + 1)
    assert r["doubles"] < band                # sparser than the band factor of the same block (more so for larger blocks)


def test_elasticity_block_with_coefficient_jumps():
    nn = 8
    rp, ci, v = gen.elasticity3d_csr(nn)
    N = 3 * nn ** 3
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))
    r = _check(A, 48)
    assert r["resid"] < 1e-12 and r["copies"] < 1e-13
    assert r["inverse"] < 1e-11 and 48 <= r["widest"] <= 512   # coefficient jumps of 1e10: the triangles stay well conditioned
    assert r["max_front"] <= N // 2 and r["doubles"] < 1.5 * N * (3 * nn * nn + 3)   # (tiny block: no gain over the band yet)


def test_unstructured_and_disconnected():
    rng = np.random.default_rng(5)
    M = sp.random(900, 900, density=0.006, random_state=rng, format="csr")
    A = M + M.T
    A = sp.csr_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0))
    r = _check(A, 32)
    assert r["resid"] < 1e-13
    # two disconnected copies: the top separator is empty
    r2 = _check(sp.block_diag([A, A], format="csr"), 32)
    assert r2["resid"] < 1e-13 and r2["supernodes"] >= 2 * r["supernodes"]
    # small matrix below the leaf size: one dense supernode
    r3 = _check(A[:40][:, :40] + 10 * sp.identity(40), 96)
    assert r3["supernodes"] == 1 and r3["resid"] < 1e-14


def test_indefinite_matrix_is_reported():
    n = 8
    rp, ci, v = gen.poisson3d_csr(n)
    A = sp.lil_matrix(sp.csr_matrix((v, ci, rp), shape=(n ** 3, n ** 3)))
    A[100, 100] = -5.0
    with pytest.raises(prealps_amd.PreAlpsError, match="not SPD"):
        _check(sp.csr_matrix(A), 32)


def test_one_sided_pattern_is_symmetrised():
    """A `general` file may store an explicit zero at (i, j) and nothing at (j, i) (operator.c accepts such
    patterns): the dissection and the lower triangle work on pattern(A) + pattern(A^T)."""
    n = 14
    rp, ci, v = gen.poisson3d_csr(n)
    N = n ** 3
    A = sp.csr_matrix((v, ci, rp), shape=(N, N)).tocoo()
    rng = np.random.default_rng(11)
    i = rng.integers(0, N, 400)
    j = (i + rng.integers(1, N, 400)) % N
    keep = np.array([A.tocsr()[a, b] == 0 and A.tocsr()[b, a] == 0 for a, b in zip(i, j)])
    i, j = i[keep], j[keep]
    rows = np.concatenate([A.row, i]); cols = np.concatenate([A.col, j]); vals = np.concatenate([A.data, np.zeros(len(i))])
    # build the CSR by hand: scipy would drop nothing, but sum duplicates -- there are none
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rp2 = np.zeros(N + 1, dtype=np.int32); np.add.at(rp2, rows + 1, 1); rp2 = np.cumsum(rp2).astype(np.int32)
    L = prealps_amd.load()
    st = np.zeros(8)
    pi, pd = C.POINTER(C.c_int), C.POINTER(C.c_double)
    ci2, v2 = cols.astype(np.int32), np.ascontiguousarray(vals, dtype=np.float64)
    check(L.preAlps_hip_nd_selfcheck(N, rp2.ctypes.data_as(pi), ci2.ctypes.data_as(pi), v2.ctypes.data_as(pd), 32,
                                     st.ctypes.data_as(pd)), "nd_selfcheck")
    assert st[4] < 1e-13 and st[6] < 1e-13
