"""ctypes front end of the CPU restatement (oracle/ecg_oracle.c) plus the
numpy restatement of the reference's *setup* steps.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (prealps_amd/) never imports it.

Setup restated here (reference file:line under /root/reference):
  symrac_scale     utils/cplm_light/cplm_matcsr.c:1461-1554
  permute_by_part  utils/cplm_v0/cplm_v0_metis_utils.c:22-43,197-222 and
                   utils/cplm_v0/cplm_v0_matcsr.c:941-1022
  reference_rhs    examples/test_ecg_prealps_op.c:172-184 (every rank calls
                   srand(0); element 0 of each rank is not scaled)
  load_mtx         utils/cplm_light/cplm_matcsr.c:96-243 (symmetric files are
                   expanded to the full pattern)
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORTHOMIN, ORTHODIR, ORTHODIR_FUSED = 0, 1, 2
ADAPT_BS, NO_BS_RED = 0, 1

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liborc.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.orc_num_threads.restype = C.c_int
    L.orc_spmm.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_bj_create.restype = C.c_void_p
    L.orc_bj_create.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, _ip]
    L.orc_bj_info.argtypes = [C.c_void_p]
    L.orc_bj_factor_bytes.restype = C.c_double
    L.orc_bj_factor_bytes.argtypes = [C.c_void_p]
    L.orc_bj_apply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_bj_free.argtypes = [C.c_void_p]
    L.orc_ecg_create.restype = C.c_void_p
    L.orc_ecg_create.argtypes = [C.c_int, C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    L.orc_ecg_destroy.argtypes = [C.c_void_p]
    L.orc_ecg_initialize.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int)]
    L.orc_ecg_iterate.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.orc_ecg_stopping.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.orc_ecg_finalize.argtypes = [C.c_void_p, _dp]
    L.orc_ecg_panel.restype = C.c_void_p
    L.orc_ecg_panel.argtypes = [C.c_void_p, C.c_int]
    for f in ("bs", "kbs", "iter", "pn"):
        getattr(L, "orc_ecg_" + f).argtypes = [C.c_void_p]
        getattr(L, "orc_ecg_" + f).restype = C.c_int
    for f in ("res", "normb"):
        getattr(L, "orc_ecg_" + f).argtypes = [C.c_void_p]
        getattr(L, "orc_ecg_" + f).restype = C.c_double
    L.orc_ecg_solve.restype = C.c_int
    L.orc_ecg_solve.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp, C.c_void_p, _dp, _dp, _dp, _ip, C.c_int, _dp]
    _LIB = L
    return L


# ---------------------------------------------------------------- setup
def as_csr(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return (A.indptr.astype(np.int32), A.indices.astype(np.int32),
            np.ascontiguousarray(A.data, dtype=np.float64))


def load_mtx(path):
    """Coordinate real general|symmetric MatrixMarket -> full CSR (scipy)."""
    import scipy.io
    A = sp.csr_matrix(scipy.io.mmread(path))
    A.sum_duplicates()
    A.sort_indices()
    return A


def symrac_scale(A):
    """A <- D A D, D = diag(1/sqrt(max_j |a_ij|))."""
    A = sp.csr_matrix(A, dtype=np.float64)
    rmax = np.zeros(A.shape[0])
    absA = abs(A)
    rmax = np.asarray(absA.max(axis=1).todense()).ravel()
    if rmax.min() == 0.0:
        raise ValueError("Impossible to scale the matrix, rcmin=0")
    d = np.sqrt(1.0 / rmax)
    indptr, indices, data = A.indptr, A.indices, A.data.copy()
    rows = np.repeat(np.arange(A.shape[0]), np.diff(indptr))
    data = d[rows] * data * d[indices]      # same association as R[i]*a*C[j]
    B = sp.csr_matrix((data, indices.copy(), indptr.copy()), shape=A.shape)
    B.sort_indices()
    return B


def contiguous_partition(N, P):
    """part[r] = floor(r*P/N): the stand-in used for the BASELINE.md probes."""
    return (np.arange(N, dtype=np.int64) * P // N).astype(np.int32)


def permute_by_part(A, part, P):
    """Rows grouped part by part, original order inside a part; symmetric
    permutation with columns re-sorted.  Returns (B, perm, rowpos)."""
    part = np.asarray(part)
    perm = np.argsort(part, kind="stable").astype(np.int64)
    counts = np.bincount(part, minlength=P)
    rowpos = np.zeros(P + 1, dtype=np.int32)
    rowpos[1:] = np.cumsum(counts)
    B = sp.csr_matrix(A)[perm][:, perm].tocsr()
    B.sort_indices()
    return B, perm.astype(np.int32), rowpos


def reference_rhs(rowpos):
    """Concatenation over ranks of the driver's rhs (glibc srand(0)/rand())."""
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    P = len(rowpos) - 1
    mmax = int(np.max(np.diff(rowpos)))
    libc.srand(0)
    stream = np.array([libc.rand() for _ in range(mmax)], dtype=np.float64) / 2147483647.0
    rhs = np.empty(int(rowpos[-1]))
    normb2 = 0.0
    for p in range(P):
        m = int(rowpos[p + 1] - rowpos[p])
        rhs[rowpos[p]:rowpos[p + 1]] = stream[:m]
        s = 0.0
        for v in stream[:m]:             # sequential sum like the C loop
            s += v * v
        normb2 += s
    normb = np.sqrt(normb2)
    for p in range(P):
        rhs[rowpos[p] + 1:rowpos[p + 1]] /= normb
    return rhs


def poisson3d(n):
    """7-point Laplacian on an n^3 grid, row (i*n+j)*n+k."""
    T = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    I = sp.identity(n)
    A = sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


# ---------------------------------------------------------------- solver
class BlockJacobi:
    def __init__(self, A, rowpos):
        self.L = lib()
        rp, ci, v = as_csr(A)
        self.rowpos = np.ascontiguousarray(rowpos, dtype=np.int32)
        self.n = A.shape[0]
        self.h = self.L.orc_bj_create(self.n, rp, ci, v, len(self.rowpos) - 1, self.rowpos)
        if self.L.orc_bj_info(self.h) != 0:
            raise ValueError("diagonal block not SPD at row %d" % (self.L.orc_bj_info(self.h) - 1))

    def factor_bytes(self):
        return self.L.orc_bj_factor_bytes(self.h)

    def apply(self, X):
        """X: (n, t) Fortran-ordered float64 -> Z same shape."""
        X = np.asfortranarray(X, dtype=np.float64)
        Z = np.zeros_like(X, order="F")
        self.L.orc_bj_apply(self.h, X.shape[1], X.ctypes.data, self.n, Z.ctypes.data, self.n)
        return Z

    def __del__(self):
        try:
            self.L.orc_bj_free(self.h)
        except Exception:
            pass


def spmm(A, X):
    rp, ci, v = as_csr(A)
    X = np.asfortranarray(X, dtype=np.float64)
    Y = np.zeros_like(X, order="F")
    n = A.shape[0]
    lib().orc_spmm(n, rp, ci, v, X.shape[1], X.ctypes.data, n, Y.ctypes.data, n)
    return Y


class ECG:
    """Holds one problem; solve() replays the reference driver loop."""

    def __init__(self, A, rowpos, t, ortho_alg=ORTHODIR, bs_red=NO_BS_RED, tol=1e-5, max_iter=1000):
        self.L = lib()
        self.A = A
        self.csr = as_csr(A)
        self.n = A.shape[0]
        self.rowpos = np.ascontiguousarray(rowpos, dtype=np.int32)
        self.t = t
        self.bj = BlockJacobi(A, rowpos)
        self.e = self.L.orc_ecg_create(self.n, len(self.rowpos) - 1, self.rowpos, t, ortho_alg, bs_red, tol, max_iter)
        if not self.e:
            raise ValueError("orc_ecg_create failed")
        self.ortho_alg = ortho_alg

    def solve(self, rhs, maxhist=4096):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        sol = np.zeros(self.n)
        res = np.zeros(maxhist)
        bs = np.zeros(maxhist, dtype=np.int32)
        timers = np.zeros(3)
        rp, ci, v = self.csr
        nh = self.L.orc_ecg_solve(self.e, self.n, rp, ci, v, self.bj.h, rhs, sol, res, bs, maxhist, timers)
        if nh < 0:
            raise RuntimeError("oracle ECG failed (%d)" % nh)
        return dict(x=sol, res=res[:nh].copy(), bs=bs[:nh].copy(), iters=self.L.orc_ecg_iter(self.e),
                    final_res=self.L.orc_ecg_res(self.e), normb=self.L.orc_ecg_normb(self.e),
                    final_bs=self.L.orc_ecg_bs(self.e),
                    t_total=timers[0], t_op=timers[1], t_prec=timers[2])

    def panel(self, which, ncol):
        ptr = self.L.orc_ecg_panel(self.e, which)
        buf = (C.c_double * (self.n * ncol)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float64).reshape((self.n, ncol), order="F")

    def __del__(self):
        try:
            self.L.orc_ecg_destroy(self.e)
        except Exception:
            pass
