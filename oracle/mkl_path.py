"""The reference's CPU *kernels* behind the same ECG algorithm: Intel MKL
`mkl_dcsrmm` for the SpMM (utils/cplm_light/cplm_kernels.c:650) and MKL PARDISO
(phase 12 / 33, cplm_kernels.c:764,833) for the block-Jacobi solves, BLAS/LAPACK
through numpy/scipy for the tall-skinny and t x t work.  This is the "MKL-CSR CPU
path" that bench.py times as `cpu_baseline` when libmkl_rt is loadable on the host
(otherwise it falls back to the plain-C oracle).

TEST INFRASTRUCTURE / BASELINE ONLY (like everything under oracle/): imported by
tests/ and by the cpu_baseline leg of bench.py, never by prealps_amd/.

One process plays all P ranks: the SpMM runs over the whole matrix and PARDISO
factors the block-diagonal matrix blockdiag(A_pp) in one call (its blocks are
independent, which is what P single-rank PARDISO instances compute).  MKL is
called through its Fortran-style by-reference entry points exactly as the
reference does (1-based indices, column-major panels).
"""
import ctypes as C
import os
import time

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

_CANDIDATES = ["libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1",
               "/opt/conda/lib/libmkl_rt.so"]
_mkl = None


def load_mkl():
    """Return the MKL runtime or None when it is not on this machine."""
    global _mkl
    if _mkl is not None:
        return _mkl or None
    for name in _CANDIDATES:
        try:
            _mkl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            _mkl.MKL_Get_Max_Threads.restype = C.c_int
            # the process already runs GNU OpenMP (numpy/torch/liborc): keep MKL on the same
            # runtime instead of loading Intel's next to it (3 = MKL_THREADING_GNU)
            try:
                _mkl.MKL_Set_Threading_Layer(C.c_int(3))
            except Exception:
                pass
            return _mkl
        except OSError:
            continue
    _mkl = False
    return None


def _ref(x, t=C.c_int):
    return C.byref(t(x))


class MklEcg:
    """ECG Orthodir without block-size reduction (src/solvers/ecg.c:402-530 with the driver
    loop of examples/test_ecg_prealps_op.c:203-223) on MKL kernels."""

    def __init__(self, A, rowpos, t, tol=1e-5, max_iter=1000, threads=None):
        mkl = load_mkl()
        if mkl is None:
            raise RuntimeError("libmkl_rt is not available on this host")
        self.mkl = mkl
        if threads:
            mkl.MKL_Set_Num_Threads(C.c_int(int(threads)))
        self.threads = mkl.MKL_Get_Max_Threads()
        A = sp.csr_matrix(A)
        A.sort_indices()
        self.n = n = A.shape[0]
        self.t, self.tol, self.max_iter = t, tol, max_iter
        self.rowpos = np.asarray(rowpos)
        # SpMM operands, 1-based for the column-major form of mkl_dcsrmm
        self.a_val = np.ascontiguousarray(A.data, dtype=np.float64)
        self.a_col = (A.indices + 1).astype(np.int32)
        self.a_pb = (A.indptr[:-1] + 1).astype(np.int32)
        self.a_pe = (A.indptr[1:] + 1).astype(np.int32)
        # block-diagonal preconditioner matrix, upper triangle (block_jacobi.c:48-58)
        part = np.repeat(np.arange(len(rowpos) - 1), np.diff(rowpos))
        coo = A.tocoo()
        keep = (part[coo.row] == part[coo.col]) & (coo.col >= coo.row)
        M = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
        M.sort_indices()
        self.m_val = np.ascontiguousarray(M.data, dtype=np.float64)
        self.m_ia = M.indptr.astype(np.int32)      # zero-based: iparm[34] = 1
        self.m_ja = M.indices.astype(np.int32)
        self.pt = (C.c_void_p * 64)()
        self.iparm = np.zeros(64, dtype=np.int32)
        self.iparm[0] = 1      # no solver defaults
        self.iparm[1] = 2      # nested dissection (METIS)
        self.iparm[9] = 13
        self.iparm[34] = 1     # zero-based indexing  (cplm_kernels.c:677-694)
        self.t_factor = self._pardiso(12, None, None)

    def _pardiso(self, phase, b, x):
        mkl, n = self.mkl, self.n
        err = C.c_int(0)
        nrhs = 1 if b is None else b.shape[1]
        dummy = np.zeros(1)
        t0 = time.perf_counter()
        mkl.pardiso(self.pt, _ref(1), _ref(1), _ref(2), _ref(phase), _ref(n),
                    self.m_val.ctypes.data_as(C.c_void_p), self.m_ia.ctypes.data_as(C.c_void_p),
                    self.m_ja.ctypes.data_as(C.c_void_p), dummy.ctypes.data_as(C.c_void_p), _ref(nrhs),
                    self.iparm.ctypes.data_as(C.c_void_p), _ref(0),
                    (dummy if b is None else b).ctypes.data_as(C.c_void_p),
                    (dummy if x is None else x).ctypes.data_as(C.c_void_p), C.byref(err))
        if err.value != 0:
            raise RuntimeError("pardiso phase %d failed with error %d" % (phase, err.value))
        return time.perf_counter() - t0

    def spmm(self, X):
        """Y = A X with mkl_dcsrmm, column-major panels."""
        n, k = self.n, X.shape[1]
        Y = np.zeros((n, k), order="F")
        self.mkl.mkl_dcsrmm(C.c_char_p(b"N"), _ref(n), _ref(k), _ref(n), _ref(1.0, C.c_double),
                            C.c_char_p(b"G  F  "), self.a_val.ctypes.data_as(C.c_void_p),
                            self.a_col.ctypes.data_as(C.c_void_p), self.a_pb.ctypes.data_as(C.c_void_p),
                            self.a_pe.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p), _ref(n),
                            _ref(0.0, C.c_double), Y.ctypes.data_as(C.c_void_p), _ref(n))
        return Y

    def precond(self, X):
        Z = np.zeros_like(X, order="F")
        self._pardiso(33, np.asfortranarray(X), Z)
        return Z

    def solve(self, rhs):
        n, t = self.n, self.t
        tm = dict(op=0.0, prec=0.0, dense=0.0)

        def timed(key, f, *a):
            t0 = time.perf_counter()
            r = f(*a)
            tm[key] += time.perf_counter() - t0
            return r

        normb = float(np.sqrt(sum(np.sum(rhs[self.rowpos[p]:self.rowpos[p + 1]] ** 2)
                                  for p in range(len(self.rowpos) - 1))))
        R = np.zeros((n, t), order="F")
        for p in range(len(self.rowpos) - 1):
            R[self.rowpos[p]:self.rowpos[p + 1], p % t] = rhs[self.rowpos[p]:self.rowpos[p + 1]]
        X = np.zeros((n, t), order="F")
        Pprev = np.zeros((n, t), order="F")
        APprev = np.zeros((n, t), order="F")
        t_all = time.perf_counter()
        P = timed("prec", self.precond, R)
        AP = timed("op", self.spmm, P)
        res_hist, it = [], 0
        while True:
            t0 = time.perf_counter()
            W = AP.T @ P
            U = sla.cholesky(W, lower=False, check_finite=False)
            P = np.asfortranarray(sla.solve_triangular(U, P.T, trans="T", lower=False, check_finite=False).T)
            AP = np.asfortranarray(sla.solve_triangular(U, AP.T, trans="T", lower=False, check_finite=False).T)
            alpha = P.T @ R
            X += P @ alpha
            R -= AP @ alpha
            it += 1
            res = float(np.sqrt(np.sum(R * R)))
            tm["dense"] += time.perf_counter() - t0
            res_hist.append(res)
            if not (res > normb * self.tol and it < self.max_iter):
                break
            Z = timed("prec", self.precond, AP)
            t0 = time.perf_counter()
            beta1, beta2 = AP.T @ Z, APprev.T @ Z
            Z -= P @ beta1 + Pprev @ beta2
            Pprev, APprev, P = P, AP, Z
            tm["dense"] += time.perf_counter() - t0
            AP = timed("op", self.spmm, P)
        total = time.perf_counter() - t_all
        return dict(x=X.sum(axis=1), iters=it, res=np.array(res_hist), normb=normb, t_total=total,
                    t_op=tm["op"], t_prec=tm["prec"], t_dense=tm["dense"], threads=self.threads)

    def solve_dodir(self, rhs):
        """Orthodir with the dynamic reduction of the search directions (-o 0 -r 1: src/solvers/ecg.c:445-497 of the
        reference, restated in oracle/ecg_oracle.c: odir_reduce / iterate_odir) on this path's kernels -- mkl_dcsrmm,
        PARDISO and LAPACK through numpy instead of the oracle's own loops.  A CONTROL, not an oracle: a second
        fp64 implementation of the same recurrence, to see how far two CPU paths drift apart in WHEN the
        reduction fires (tools/history_control.py --dodir).  Same memory layout as the reference: V = [T current
        slots | T previous slots], the rotated-away columns stay in the basis for one more step."""
        n, T = self.n, self.t
        normb = float(np.sqrt(sum(np.sum(rhs[self.rowpos[p]:self.rowpos[p + 1]] ** 2)
                                  for p in range(len(self.rowpos) - 1))))
        R = np.zeros((n, T), order="F")
        for p in range(len(self.rowpos) - 1):
            R[self.rowpos[p]:self.rowpos[p + 1], p % T] = rhs[self.rowpos[p]:self.rowpos[p + 1]]
        X = np.zeros((n, T), order="F")
        V = np.zeros((n, 2 * T), order="F")
        AV = np.zeros((n, 2 * T), order="F")
        t, kbs = T, 2 * T
        V[:, :T] = self.precond(R)
        AV[:, :T] = self.spmm(np.asfortranarray(V[:, :T]))
        thresh = self.tol * normb / np.sqrt(float(T))
        res_hist, bs_hist, it = [], [], 0
        while True:
            P, AP = V[:, :t], AV[:, :t]
            W = AP.T @ P
            try:
                U = sla.cholesky(W, lower=False, check_finite=False)
            except np.linalg.LinAlgError:
                break
            P[:] = sla.solve_triangular(U, P.T, trans="T", lower=False, check_finite=False).T
            AP[:] = sla.solve_triangular(U, AP.T, trans="T", lower=False, check_finite=False).T
            alpha = P.T @ R                                   # t x T
            Us, sig, _ = np.linalg.svd(alpha, full_matrices=True)
            t1 = 0
            for sv in sig[:t]:
                if sv > thresh:
                    t1 += 1
                else:
                    break
            if 0 < t1 < T and t1 < t:
                Q, _ = np.linalg.qr(Us[:, :t])                # Householder Q of the left singular vectors
                alpha = (Q.T @ alpha)[:t1]
                P[:] = P @ Q
                AP[:] = AP @ Q
                kbs = t + T
                t = t1
            nb = t1 if t1 > 0 else t
            X += V[:, :t] @ alpha[:t]
            R -= AV[:, :t] @ alpha[:t]
            it += 1
            res = float(np.sqrt(np.sum(R * R)))
            res_hist.append(res)
            bs_hist.append(t)
            if not (res > normb * self.tol and it < self.max_iter and t > 0):
                break
            Z = self.precond(np.asfortranarray(AV[:, :t]))
            beta = AV[:, :kbs].T @ Z[:, :nb]
            Z[:, :nb] -= V[:, :kbs] @ beta
            V[:, T:T + t] = V[:, :t]
            AV[:, T:T + t] = AV[:, :t]
            V[:, :t] = Z[:, :t]
            AV[:, :t] = self.spmm(np.asfortranarray(V[:, :t]))
        return dict(x=X.sum(axis=1), iters=it, res=np.array(res_hist), bs=np.array(bs_hist), normb=normb)

    def __del__(self):
        try:
            self._pardiso(-1, None, None)
        except Exception:
            pass
