"""Kernel-level parity of the tall-skinny panel kernels (Gram products, Z update, the fused
triangular solve + iterate update) against numpy on the host, through the C ABI
(preAlps_hip_panel_gram / _update / _trsm_update launch what the solver launches).  Covers the
register-tiled kernels (panel stride 2, 4) and the matrix-core ones (stride 8, 16), column
counts off the stride (3, 5, 12), and row counts that are not multiples of any tile."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Panels:
    def __init__(self):
        import prealps_amd
        from prealps_amd.lib import check
        self.L, self.check = prealps_amd.load(), check
        check(self.L.preAlps_hip_init(0), "init")
        self.live = []

    def up(self, H, t):
        from prealps_amd.lib import CPLM_Mat_Dense_t
        m, n = H.shape
        d = CPLM_Mat_Dense_t()
        self.check(self.L.preAlps_hip_panel_alloc(C.byref(d), m, n, m, n, t), "alloc")
        Hf = np.asfortranarray(H)
        self.check(self.L.preAlps_hip_panel_from_host(C.byref(d), t, _pd(Hf), m), "h2d")
        self.live.append(d)
        return d

    def down(self, d, t):
        out = np.zeros((d.info.m, d.info.n), order="F")
        self.check(self.L.preAlps_hip_panel_to_host(C.byref(d), t, _pd(out), max(d.info.m, 1)), "d2h")
        return out

    def close(self):
        for d in self.live:
            self.L.preAlps_hip_panel_free(C.byref(d))


@pytest.fixture
def panels():
    p = Panels()
    yield p
    p.close()


@pytest.mark.parametrize("m", [1, 1000, 40961])
@pytest.mark.parametrize("t,na1", [(2, 2), (4, 0), (4, 4), (3, 2), (8, 0), (8, 8), (5, 3), (16, 0), (16, 16), (12, 7)])
def test_gram(panels, m, t, na1):
    rng = np.random.default_rng(m + 31 * t + na1)
    A0, B = rng.standard_normal((m, t)), rng.standard_normal((m, t))
    A1 = rng.standard_normal((m, na1)) if na1 else None
    d0, db = panels.up(A0, t), panels.up(B, t)
    d1 = panels.up(A1, t) if na1 else None
    na = t + na1
    out = np.zeros((na, t), order="F")
    panels.check(panels.L.preAlps_hip_panel_gram(C.byref(d0), C.byref(d1) if na1 else None, C.byref(db), _pd(out), na),
                 "gram")
    ref = (np.hstack([A0, A1]) if na1 else A0).T @ B
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("m", [7, 4099])
@pytest.mark.parametrize("t,na1,nc", [(2, 2, 2), (4, 4, 4), (4, 0, 4), (4, 4, 3), (8, 8, 8), (8, 5, 6), (8, 0, 8),
                                      (16, 16, 16), (16, 0, 16), (12, 9, 12)])
def test_update_z(panels, m, t, na1, nc):
    rng = np.random.default_rng(m + t + 5 * na1 + nc)
    Z, V0 = rng.standard_normal((m, nc)), rng.standard_normal((m, t))
    V1 = rng.standard_normal((m, na1)) if na1 else None
    na = t + na1
    beta = np.asfortranarray(rng.standard_normal((na, nc)))
    dz, d0 = panels.up(Z, t), panels.up(V0, t)
    d1 = panels.up(V1, t) if na1 else None
    panels.check(panels.L.preAlps_hip_panel_update(C.byref(dz), C.byref(d0), C.byref(d1) if na1 else None,
                                                  _pd(beta), na), "update")
    ref = Z - (np.hstack([V0, V1]) if na1 else V0) @ beta
    np.testing.assert_allclose(panels.down(dz, t), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("m", [5, 10007])
@pytest.mark.parametrize("t,nc", [(1, 1), (2, 2), (4, 4), (3, 4), (8, 8), (6, 8), (16, 16), (11, 16), (12, 12)])
def test_trsm_update(panels, m, t, nc):
    """t = current block size (columns of P, AP), nc = enlarging factor (columns of X, R)."""
    T = nc
    rng = np.random.default_rng(m + 17 * t + nc)
    P, AP = rng.standard_normal((m, t)), rng.standard_normal((m, t))
    X, R = rng.standard_normal((m, nc)), rng.standard_normal((m, nc))
    U = np.asfortranarray(np.triu(rng.standard_normal((t, t))) + 4.0 * np.eye(t))
    alpha = np.asfortranarray(rng.standard_normal((t, nc)))
    dp, dap, dx, dr = panels.up(P, T), panels.up(AP, T), panels.up(X, T), panels.up(R, T)
    res2 = C.c_double()
    panels.check(panels.L.preAlps_hip_panel_trsm_update(C.byref(dp), C.byref(dap), C.byref(dx), C.byref(dr),
                                                       _pd(U), _pd(alpha), C.byref(res2)), "trsm_update")
    Ui = np.linalg.inv(U)
    Pn, APn = P @ Ui, AP @ Ui
    Xn, Rn = X + Pn @ alpha, R - APn @ alpha
    for got, ref in ((panels.down(dp, T), Pn), (panels.down(dap, T), APn), (panels.down(dx, T), Xn),
                     (panels.down(dr, T), Rn)):
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-11 * np.abs(ref).max())
    assert abs(res2.value - (Rn ** 2).sum()) <= 1e-11 * (Rn ** 2).sum()


@pytest.mark.parametrize("m", [7, 30011])
@pytest.mark.parametrize("n,t", [(2, 2), (4, 4), (4, 3), (3, 1), (8, 8), (8, 5), (5, 5), (16, 16), (16, 9), (12, 0)])
def test_permute_solve(panels, m, n, t):
    """BF-Omin's copy + dlapmt + dtrsm (ecg.c:358-393) in one kernel: against numpy, and bit for bit against the
    three kernels it replaces; the columns behind the rank keep the permuted copy."""
    rng = np.random.default_rng(m + 13 * n + t)
    Z, P0 = rng.standard_normal((m, n)), rng.standard_normal((m, n))
    piv = rng.permutation(n).astype(np.int32)
    U = np.asfortranarray(np.triu(rng.standard_normal((t, t))) + 3.0 * np.eye(t)) if t else np.zeros((1, 1))
    outs = []
    for one_pass in (1, 0):
        dz, dp = panels.up(Z, n), panels.up(P0, n)
        panels.check(panels.L.preAlps_hip_panel_permute_solve(C.byref(dz), C.byref(dp), piv.ctypes.data_as(C.POINTER(C.c_int)),
                                                             t, _pd(U), one_pass), "permute_solve")
        outs.append(panels.down(dp, n))
        np.testing.assert_array_equal(panels.down(dz, n), Z)
    ref = Z[:, piv].copy()
    if t:
        ref[:, :t] = ref[:, :t] @ np.linalg.inv(U)
    np.testing.assert_allclose(outs[0], ref, rtol=1e-11, atol=1e-11 * np.abs(ref).max())
    np.testing.assert_array_equal(outs[0], outs[1])
