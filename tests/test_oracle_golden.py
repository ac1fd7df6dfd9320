"""Pins the CPU restatement (oracle/) against the reference's own recorded
outputs (tests/golden/reference_probe.json <- BASELINE.md section 2)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL_HIST = 1e-9   # MKL vs plain loops differ by summation order only


def _poisson24(P=8):
    A = O.symrac_scale(O.poisson3d(24))
    B, perm, rowpos = O.permute_by_part(A, O.contiguous_partition(A.shape[0], P), P)
    return B, rowpos, O.reference_rhs(rowpos)


def _lfat5(P):
    A = O.symrac_scale(O.load_mtx(os.path.join(GOLD, "LFAT5.mtx")))
    B, perm, rowpos = O.permute_by_part(A, O.contiguous_partition(A.shape[0], P), P)
    return B, rowpos, O.reference_rhs(rowpos)


def test_rhs_stream_and_normb(golden):
    B, rowpos, rhs = _poisson24()
    # glibc srand(0) stream, SURVEY Appendix A.1
    assert abs(rhs[0] - 1804289383 / 2147483647) < 1e-16
    e = O.ECG(B, rowpos, 4)
    r = e.solve(rhs)
    assert abs(r["normb"] - golden["poisson24_np8_t4"]["normb"]) < 1e-14


def test_poisson24_odir_history(golden):
    B, rowpos, rhs = _poisson24()
    g = golden["poisson24_np8_t4"]["odir"]
    r = O.ECG(B, rowpos, 4, O.ORTHODIR, O.NO_BS_RED).solve(rhs)
    assert r["iters"] == g["iters"]
    np.testing.assert_allclose(r["res"], g["res"], rtol=RTOL_HIST)
    assert np.all(r["bs"] == 4)
    # the returned iterate really solves the (scaled, permuted) system:
    # b - A x = sum of the t residual columns, so its norm is <= sqrt(t) ||R||_F
    assert np.linalg.norm(B @ r["x"] - rhs) <= 2.0001 * r["final_res"]


def test_poisson24_omin(golden):
    B, rowpos, rhs = _poisson24()
    g = golden["poisson24_np8_t4"]
    r = O.ECG(B, rowpos, 4, O.ORTHOMIN, O.NO_BS_RED).solve(rhs)
    assert r["iters"] == g["omin"]["iters"]
    assert abs(r["final_res"] / g["omin"]["final_res"] - 1) < RTOL_HIST
    np.testing.assert_allclose(r["res"], g["odir"]["res"], rtol=1e-8)


def test_poisson24_dodir_history_and_block_sizes(golden):
    B, rowpos, rhs = _poisson24()
    g = golden["poisson24_np8_t4"]["dodir"]
    r = O.ECG(B, rowpos, 4, O.ORTHODIR, O.ADAPT_BS).solve(rhs)
    assert r["iters"] == g["iters"]
    np.testing.assert_allclose(r["res"], [x[0] for x in g["res_bs"]], rtol=RTOL_HIST)
    assert list(r["bs"]) == [x[1] for x in g["res_bs"]]


def test_poisson24_bfomin_t8_t1(golden):
    B, rowpos, rhs = _poisson24()
    g = golden["poisson24_np8_t4"]
    r = O.ECG(B, rowpos, 4, O.ORTHOMIN, O.ADAPT_BS).solve(rhs)
    assert (r["iters"], r["final_bs"]) == (g["bfomin"]["iters"], g["bfomin"]["final_bs"])
    assert "%.6e" % r["final_res"] == "%.6e" % g["bfomin"]["final_res_print"]
    r = O.ECG(B, rowpos, 8, O.ORTHODIR, O.ADAPT_BS).solve(rhs)
    assert (r["iters"], r["final_bs"]) == (g["t8_dodir"]["iters"], g["t8_dodir"]["final_bs"])
    assert "%.6e" % r["final_res"] == "%.6e" % g["t8_dodir"]["final_res_print"]
    r = O.ECG(B, rowpos, 1, O.ORTHODIR, O.NO_BS_RED).solve(rhs)
    assert r["iters"] == g["t1"]["iters"]
    assert "%.6e" % r["final_res"] == "%.6e" % g["t1"]["final_res_print"]


def test_lfat5(golden):
    g = golden["lfat5"]
    B, rowpos, rhs = _lfat5(2)
    r = O.ECG(B, rowpos, 2, O.ORTHODIR, O.NO_BS_RED).solve(rhs)
    assert r["iters"] == g["np2_t2_odir"]["iters"]
    assert abs(r["normb"] - g["np2_t2_odir"]["normb"]) < 1e-14
    # the last entry is rounding noise (1e-13); the first four are pinned
    np.testing.assert_allclose(r["res"][:4], g["np2_t2_odir"]["res"][:4], rtol=1e-11)
    assert r["res"][4] < 1e-11
    r = O.ECG(B, rowpos, 2, O.ORTHOMIN, O.NO_BS_RED).solve(rhs)
    assert r["iters"] == g["np2_t2_omin"]["iters"] and r["final_res"] < 1e-11
    B, rowpos, rhs = _lfat5(4)
    r = O.ECG(B, rowpos, 2, O.ORTHODIR, O.NO_BS_RED).solve(rhs)
    assert r["iters"] == g["np4_t2_odir"]["iters"] and r["final_res"] < 1e-11


def test_fused_odir_matches_odir():
    """ecg.c:532-658 is algebraically Odir with one reduction per iteration;
    the fused loop tests the residual *before* the update, so it needs one
    more call than Odir."""
    B, rowpos, rhs = _poisson24()
    ro = O.ECG(B, rowpos, 4, O.ORTHODIR, O.NO_BS_RED).solve(rhs)
    rf = O.ECG(B, rowpos, 4, O.ORTHODIR_FUSED, O.NO_BS_RED).solve(rhs)
    assert rf["iters"] == ro["iters"] + 1
    np.testing.assert_allclose(rf["res"][1:], ro["res"], rtol=1e-7)


def test_enlarging_factor_exceeds_parts():
    B, rowpos, rhs = _lfat5(2)
    with pytest.raises(RuntimeError):
        O.ECG(B, rowpos, 4).solve(rhs)     # ecg.c:178-183 aborts


def test_block_jacobi_exact_and_spmm():
    B, rowpos, rhs = _poisson24()
    rng = np.random.default_rng(0)
    X = rng.standard_normal((B.shape[0], 3))
    np.testing.assert_allclose(O.spmm(B, X), B @ X, rtol=1e-13, atol=1e-13)
    Z = O.BlockJacobi(B, rowpos).apply(X)
    for p in range(8):
        s = slice(rowpos[p], rowpos[p + 1])
        D = B[s, s].toarray()
        np.testing.assert_allclose(D @ Z[s], X[s], rtol=1e-9, atol=1e-10)


def test_mkl_cpu_path_reproduces_the_recorded_reference_history(golden):
    """The CPU baseline of bench.py: the same algorithm on the reference's own kernels
    (mkl_dcsrmm + MKL PARDISO).  Skipped where libmkl_rt is not installed."""
    from oracle import mkl_path as M
    if M.load_mkl() is None:
        pytest.skip("libmkl_rt not available")
    B, rowpos, rhs = _poisson24()
    g = golden["poisson24_np8_t4"]
    e = M.MklEcg(B, rowpos, 4, threads=4)
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((B.shape[0], 4)))
    np.testing.assert_allclose(e.spmm(X), B @ X, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(e.precond(X), O.BlockJacobi(B, rowpos).apply(X), rtol=1e-10, atol=1e-11)
    r = e.solve(rhs)
    assert r["iters"] == g["odir"]["iters"]
    assert abs(r["normb"] - g["normb"]) < 1e-13
    np.testing.assert_allclose(r["res"], g["odir"]["res"], rtol=RTOL_HIST)
