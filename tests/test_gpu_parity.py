"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the
reference's recorded outputs.  fp64 throughout; the reference itself sums in
message-arrival order, so parity is by tolerance:
  single kernels (SpMM, block solve)   1e-12 relative to the operand norms
  residual histories / iterates        1e-8  (MKL vs oracle differ by 1e-11)
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL_HIST = 1e-8


def _problem(A, P, part=None, **kw):
    import prealps_amd
    from oracle import oracle as O
    part = O.contiguous_partition(A.shape[0], P) if part is None else part
    rp, ci, v = O.as_csr(A)
    prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0, **kw)
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
    return prob, B, rowpos


@pytest.fixture
def poisson24():
    # operator and preconditioner are process-global in the library (as in the
    # reference), so every test builds its own
    from oracle import oracle as O
    prob, B, rowpos = _problem(O.poisson3d(24), 8)
    yield prob, B, rowpos
    prob.close()


def _random_spd(n, density, seed):
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="csr")
    A = M + M.T
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


def test_operator_matches_reference_setup(poisson24):
    prob, B, rowpos = poisson24
    assert np.array_equal(prob.rowpos, rowpos)
    rp, ci, v = prob.local_csr()
    assert np.array_equal(rp, B.indptr) and np.array_equal(ci, B.indices)
    np.testing.assert_allclose(v, B.data, rtol=1e-15)
    from oracle import oracle as O
    np.testing.assert_array_equal(prob.reference_rhs(), O.reference_rhs(rowpos))


@pytest.mark.parametrize("t", [1, 2, 4, 8, 16])
def test_spmm_parity(poisson24, t):
    from oracle import oracle as O
    prob, B, rowpos = poisson24
    X = np.random.default_rng(t).standard_normal((B.shape[0], t))
    got = prob.block_operator(X, t)
    ref = O.spmm(B, X)
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())


@pytest.mark.parametrize("t", [1, 4, 8, 16])
def test_block_jacobi_parity(poisson24, t):
    from oracle import oracle as O
    prob, B, rowpos = poisson24
    X = np.random.default_rng(10 + t).standard_normal((B.shape[0], t))
    got = prob.block_jacobi_apply(X, t)
    ref = O.BlockJacobi(B, rowpos).apply(X)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max())


def test_spmm_and_block_solve_on_ragged_rows():
    """Random pattern: rows of very different lengths, an empty-ish row, tiny parts."""
    from oracle import oracle as O
    A = _random_spd(1537, 0.01, 7)
    P = 37
    prob, B, rowpos = _problem(A, P)
    try:
        for t in (4, 8):
            X = np.random.default_rng(t).standard_normal((B.shape[0], t))
            np.testing.assert_allclose(prob.block_operator(X, t), O.spmm(B, X), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(prob.block_jacobi_apply(X, t), O.BlockJacobi(B, rowpos).apply(X),
                                       rtol=1e-10, atol=1e-11)
    finally:
        prob.close()


@pytest.mark.parametrize("t", [2, 4, 8, 16])
@pytest.mark.parametrize("kind", ["random", "poisson", "elasticity_cut"])
def test_spmm_run_plan_on_irregular_patterns(kind, t, monkeypatch):
    """The run plan (one slot per three consecutive staging slots) forced onto patterns it
    would not choose: broken runs, gaps inside a run, runs that straddle the edge of a
    block's own rows, and subdomains that cut a node's three dofs apart."""
    monkeypatch.setenv("PREALPS_SPMM_RUNS", "2")
    from oracle import oracle as O
    from prealps_amd import gen
    if kind == "random":
        A, P, part = _random_spd(1500, 0.004, 11), 7, None
    elif kind == "poisson":
        A, P, part = O.poisson3d(12), 5, None
    else:
        rp, ci, v = gen.elasticity3d_csr(7)
        A, P, part = sp.csr_matrix((v, ci, rp), shape=(3 * 343, 3 * 343)), 8, None   # contiguous rows: 1029/8 cuts nodes
    prob, B, rowpos = _problem(A, P, part)
    try:
        X = np.random.default_rng(5).standard_normal((B.shape[0], t))
        ref = O.spmm(B, X)
        np.testing.assert_allclose(prob.block_operator(X, t), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        assert prob.stat("spmm_runs") == 1.0
    finally:
        prob.close()


def test_ecg_odir_history_vs_recorded_reference(poisson24, golden):
    prob, B, rowpos = poisson24
    g = golden["poisson24_np8_t4"]
    got = prob.solve(prob.reference_rhs(), 4)
    assert got.iters == g["odir"]["iters"]
    assert abs(got.normb - g["normb"]) < 1e-13
    np.testing.assert_allclose(got.res, g["odir"]["res"], rtol=RTOL_HIST)


@pytest.mark.parametrize("alg", ["odir", "omin", "fused"])
@pytest.mark.parametrize("t", [1, 2, 4, 8])
def test_ecg_parity_with_oracle(poisson24, alg, t):
    import prealps_amd as pa
    from oracle import oracle as O
    prob, B, rowpos = poisson24
    algs = {"odir": (pa.ORTHODIR, O.ORTHODIR), "omin": (pa.ORTHOMIN, O.ORTHOMIN),
            "fused": (pa.ORTHODIR_FUSED, O.ORTHODIR_FUSED)}[alg]
    rhs = prob.reference_rhs()
    got = prob.solve(rhs, t, ortho_alg=algs[0])
    ref = O.ECG(B, rowpos, t, algs[1], O.NO_BS_RED).solve(rhs)
    assert got.iters == ref["iters"]
    np.testing.assert_allclose(got.res, ref["res"], rtol=RTOL_HIST)
    np.testing.assert_allclose(got.x, ref["x"], rtol=1e-7, atol=1e-9 * np.abs(ref["x"]).max())
    assert np.all(got.bs == t)


@pytest.mark.parametrize("alg,t", [("odir", 4), ("odir", 8), ("omin", 4), ("fused", 4)])
def test_ecg_block_size_reduction_parity(poisson24, golden, alg, t):
    import prealps_amd as pa
    from oracle import oracle as O
    prob, B, rowpos = poisson24
    algs = {"odir": (pa.ORTHODIR, O.ORTHODIR), "omin": (pa.ORTHOMIN, O.ORTHOMIN),
            "fused": (pa.ORTHODIR_FUSED, O.ORTHODIR_FUSED)}[alg]
    rhs = prob.reference_rhs()
    got = prob.solve(rhs, t, ortho_alg=algs[0], bs_red=pa.ADAPT_BS)
    ref = O.ECG(B, rowpos, t, algs[1], O.ADAPT_BS).solve(rhs)
    assert got.iters == ref["iters"]
    assert list(got.bs) == list(ref["bs"])
    np.testing.assert_allclose(got.res, ref["res"], rtol=1e-7)
    if (alg, t) == ("odir", 4):
        g = golden["poisson24_np8_t4"]["dodir"]
        np.testing.assert_allclose(got.res, [x[0] for x in g["res_bs"]], rtol=1e-7)
        assert list(got.bs) == [x[1] for x in g["res_bs"]]


def test_lfat5_from_matrixmarket_file(golden):
    import prealps_amd
    prob = prealps_amd.EcgProblem.from_mtx(os.path.join(GOLD, "LFAT5.mtx"), nparts=2, partition="contiguous")
    try:
        got = prob.solve(prob.reference_rhs(), 2)
        g = golden["lfat5"]["np2_t2_odir"]
        assert got.iters == g["iters"]
        assert abs(got.normb - g["normb"]) < 1e-13
        np.testing.assert_allclose(got.res[:4], g["res"][:4], rtol=1e-10)
        assert got.res[4] < 1e-11
    finally:
        prob.close()


def test_enlarging_factor_larger_than_parts_is_refused():
    import prealps_amd
    prob = prealps_amd.EcgProblem.from_mtx(os.path.join(GOLD, "LFAT5.mtx"), nparts=2, partition="contiguous")
    try:
        with pytest.raises(prealps_amd.PreAlpsError, match="Enlarging factor"):
            prob.solve(prob.reference_rhs(), 4)
    finally:
        prob.close()


def test_geometric_partition_many_small_blocks():
    """64 sub-cubes of 6^3 on a 24^3 grid: the shape the benchmark uses."""
    import prealps_amd as pa
    from oracle import oracle as O
    n, s = 24, 6
    idx = np.arange(n ** 3)
    i, j, k = idx // (n * n), (idx // n) % n, idx % n
    part = ((i // s) * (n // s) + (j // s)) * (n // s) + (k // s)
    prob, B, rowpos = _problem(O.poisson3d(n), (n // s) ** 3, part.astype(np.int32))
    try:
        rhs = prob.reference_rhs()
        got = prob.solve(rhs, 4)
        ref = O.ECG(B, rowpos, 4).solve(rhs)
        assert got.iters == ref["iters"]
        np.testing.assert_allclose(got.res, ref["res"], rtol=RTOL_HIST)
        # size-independent property: the iterate solves the scaled, permuted system
        r = B @ got.x - rhs
        assert np.linalg.norm(r) <= 2.0001 * got.final_res
    finally:
        prob.close()


def test_c_driver_end_to_end(tmp_path, golden):
    """examples/ecg_driver.c (the reference driver's call sequence in C) on LFAT5."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ecg_driver")
    subprocess.check_call(["gcc", "-std=gnu11", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "ecg_driver.c"), "-L" + os.path.join(root, "prealps_amd"),
                           "-lprealps_hip", "-Wl,-rpath," + os.path.join(root, "prealps_amd"), "-lm", "-o", exe])
    r = subprocess.run([exe, "-m", os.path.join(GOLD, "LFAT5.mtx"), "-e", "2", "-o", "0", "-r", "0"],
                       capture_output=True, text=True, env=dict(os.environ, PREALPS_NPARTS="2", PREALPS_PARTITION="contiguous"), timeout=120)
    assert r.returncode == 0, r.stderr
    it = int(re.search(r"iter: (\d+)", r.stdout).group(1))
    res = float(re.search(r"res : (\S+)", r.stdout).group(1))
    assert it == golden["lfat5"]["np2_t2_odir"]["iters"] and res < 1e-11


@pytest.mark.parametrize("runs", ["0", "1"])
def test_elasticity_q1_parity(runs, monkeypatch):
    """3 dofs per node, 81 nonzeros per interior row, coefficient jumps of 1e10: the
    matrix class of the headline metric (long SELL slices, wider bands); with the staged
    SpMM that shares one LDS slot per run of three columns (default) and the scalar one."""
    monkeypatch.setenv("PREALPS_SPMM_RUNS", runs)
    import prealps_amd as pa
    from prealps_amd import gen
    from oracle import oracle as O
    nn = 9
    rp, ci, v = gen.elasticity3d_csr(nn)
    part, nparts = gen.box_partition_nodes(nn, (3, 3, 3))
    A = sp.csr_matrix((v, ci, rp), shape=(3 * nn ** 3, 3 * nn ** 3))
    assert abs(A - A.T).max() < 1e-12
    prob, B, rowpos = _problem(A, nparts, part)
    try:
        X = np.random.default_rng(3).standard_normal((B.shape[0], 4))
        ref = O.spmm(B, X)
        np.testing.assert_allclose(prob.block_operator(X, 4), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        assert prob.stat("spmm_runs") == float(runs) and prob.stat("spmm_staged") == 1.0
        zr = O.BlockJacobi(B, rowpos).apply(X)
        np.testing.assert_allclose(prob.block_jacobi_apply(X, 4), zr, rtol=1e-8, atol=1e-9 * np.abs(zr).max())
        rhs = prob.reference_rhs()
        got = prob.solve(rhs, 4, max_iter=400)
        refs = O.ECG(B, rowpos, 4, max_iter=400).solve(rhs)
        assert abs(got.iters - refs["iters"]) <= 1
        k = min(len(got.res), len(refs["res"])) - 2
        np.testing.assert_allclose(got.res[:k], refs["res"][:k], rtol=1e-5)
    finally:
        prob.close()


@pytest.mark.parametrize("t,alg", [(3, "odir"), (5, "omin"), (16, "odir"), (12, "odir")])
def test_enlarging_factors_off_the_power_of_two_grid(t, alg):
    """t = 3, 5, 12 run on padded panels (stride 4, 8, 16); t = 16 is the largest supported."""
    import prealps_amd as pa
    from oracle import oracle as O
    n, s3 = 16, 4
    idx = np.arange(n ** 3)
    i, j, k = idx // (n * n), (idx // n) % n, idx % n
    part = (((i // s3) * (n // s3) + (j // s3)) * (n // s3) + (k // s3)).astype(np.int32)
    prob, B, rowpos = _problem(O.poisson3d(n), (n // s3) ** 3, part)
    try:
        algs = {"odir": (pa.ORTHODIR, O.ORTHODIR), "omin": (pa.ORTHOMIN, O.ORTHOMIN)}[alg]
        rhs = prob.reference_rhs()
        got = prob.solve(rhs, t, ortho_alg=algs[0])
        ref = O.ECG(B, rowpos, t, algs[1], O.NO_BS_RED).solve(rhs)
        assert got.iters == ref["iters"]
        np.testing.assert_allclose(got.res, ref["res"], rtol=1e-7)
        np.testing.assert_allclose(got.x, ref["x"], rtol=1e-6, atol=1e-9 * np.abs(ref["x"]).max())
    finally:
        prob.close()


def test_unsupported_sizes_fail_loudly():
    import prealps_amd as pa
    from oracle import oracle as O
    prob, B, rowpos = _problem(O.poisson3d(12), 27, None)
    try:
        with pytest.raises(pa.PreAlpsError, match="enlarging factor"):
            prob.solve(prob.reference_rhs(), 17)
    finally:
        prob.close()


@pytest.mark.parametrize("t", [4, 8, 16])
def test_wide_band_blocks_few_large_subdomains(t):
    """Few, large subdomains (the reference's regime at small rank counts): bands wider
    than one wavefront holds go through the workgroup-resident block solve."""
    import prealps_amd as pa
    from oracle import oracle as O
    n, P = 32, 2                       # 2 slabs of 16 x 32 x 32: bandwidth ~ 500-1000
    prob, B, rowpos = _problem(O.poisson3d(n), P)
    try:
        X = np.random.default_rng(t).standard_normal((B.shape[0], t))
        zr = O.BlockJacobi(B, rowpos).apply(X)
        np.testing.assert_allclose(prob.block_jacobi_apply(X, t), zr, rtol=1e-9, atol=1e-10 * np.abs(zr).max())
        assert prob.stat("bj_max_bandwidth") > 448
        if t == 4:
            rhs = prob.reference_rhs()          # enlarging factor <= number of subdomains
            got = prob.solve(rhs, 2)
            ref = O.ECG(B, rowpos, 2).solve(rhs)
            assert got.iters == ref["iters"]
            np.testing.assert_allclose(got.res, ref["res"], rtol=RTOL_HIST)
    finally:
        prob.close()


def test_generic_preconditioner_handle(poisson24):
    """preAlps_PreconditionerCreate / MatApply / Destroy (preAlps_preconditioner.c:20-76):
    NOPREC copies, BLOCKJACOBI is the block solve, the others are refused loudly."""
    import ctypes as C
    import prealps_amd as pa
    from oracle import oracle as O
    prob, B, rowpos = poisson24
    prob.create_block_jacobi()
    L = prob.L
    X = np.random.default_rng(8).standard_normal((B.shape[0], 4))
    for kind, ref in ((0, X), (1, O.BlockJacobi(B, rowpos).apply(X))):
        h = C.c_void_p()
        assert L.preAlps_PreconditionerCreate(C.byref(h), kind, None) == 0
        dx, dy = prob.panel(4, 4), prob.panel(4, 4)
        try:
            prob.to_device(dx, X, 4)
            assert L.preAlps_PreconditionerMatApply(h, C.byref(dx), C.byref(dy)) == 0
            np.testing.assert_allclose(prob.to_host(dy, 4), ref, rtol=1e-9, atol=1e-10 * np.abs(ref).max())
        finally:
            prob.panel_free(dx)
            prob.panel_free(dy)
        assert L.preAlps_PreconditionerDestroy(C.byref(h)) == 0 and not h.value
    h = C.c_void_p()
    assert L.preAlps_PreconditionerCreate(C.byref(h), 2, None) == 0     # PREALPS_LORASC
    dx, dy = prob.panel(4, 4), prob.panel(4, 4)
    try:
        assert L.preAlps_PreconditionerMatApply(h, C.byref(dx), C.byref(dy)) != 0
        assert b"Unknown preconditioner" in L.preAlps_hip_last_error()
    finally:
        prob.panel_free(dx)
        prob.panel_free(dy)
        L.preAlps_PreconditionerDestroy(C.byref(h))


def test_full_size_properties_of_the_headline_workload():
    """BASELINE.json's metric workload at full size (Q1 elasticity 70^3 nodes, 1.03 M dofs, 81 M
    nonzeros, t = 4, the bench's partition): too large for the oracle's solver in test time, so
    checked through size-independent properties against scipy on the host:
      SpMM                     A X equals the CSR product               (1e-12)
      block solve              blockdiag(A)^-1 (blockdiag(A) X) = X     (1e-8)
      SpMM linearity           A (aX + bY) = a AX + b AY                (1e-12)
      ECG                      the returned iterate satisfies ||b - A x|| <= 2 res, res <= tol ||b||
                               and the residual history decreases to it (the first 16 residuals are
                               compared with the oracle in test_gpu_configs.py)."""
    from oracle import oracle as O
    from prealps_amd import gen
    nn, t = 70, 4
    rp, ci, v = gen.elasticity3d_csr(nn)
    part, nparts = gen.box_partition_nodes(nn, (2, 4, 8))
    N = 3 * nn ** 3
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))
    prob, B, rowpos = _problem(A, nparts, part)
    try:
        assert prob.stat("rows_local") == N and B.nnz == 80990208
        rng = np.random.default_rng(70)
        X, Y = rng.standard_normal((N, t)), rng.standard_normal((N, t))
        AX, AY = prob.block_operator(X, t), prob.block_operator(Y, t)
        ref = B @ X
        np.testing.assert_allclose(AX, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        assert prob.stat("spmm_runs") == 1.0
        lin = prob.block_operator(0.5 * X - 2.0 * Y, t)
        np.testing.assert_allclose(lin, 0.5 * AX - 2.0 * AY, rtol=1e-12, atol=1e-11 * np.abs(AX).max())
        pid = np.repeat(np.arange(nparts), np.diff(rowpos))
        coo = B.tocoo()
        keep = pid[coo.row] == pid[coo.col]
        D = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=B.shape)
        back = prob.block_jacobi_apply(D @ X, t)
        np.testing.assert_allclose(back, X, rtol=1e-8, atol=1e-8 * np.abs(X).max())
        rhs = prob.reference_rhs()
        np.testing.assert_array_equal(rhs, O.reference_rhs(rowpos))
        got = prob.solve(rhs, t, max_iter=3000)
        assert got.iters < 3000, got.iters
        print("full-size elasticity 70^3, t = 4: %d iterations to tol 1e-5" % got.iters)
        assert got.final_res <= 1e-5 * got.normb
        r = rhs - B @ got.x
        assert np.linalg.norm(r) <= 2.0001 * got.final_res
        assert got.res[-1] == got.final_res and got.res[0] > 1e3 * got.final_res
    finally:
        prob.close()


@pytest.mark.parametrize("case", ["small", "mid", "wide"])
def test_block_factorisation_on_device_matches_host(case, monkeypatch):
    """Band Cholesky of the diagonal blocks on the device -- k_bj_factor (bands up to 96, LDS
    window) and k_bj_factor_big (wider, blocked, diagonal-major band) -- against the host
    factorisation of the same blocks: the block solves agree to rounding and both invert
    blockdiag(A)."""
    from oracle import oracle as O
    from prealps_amd import gen
    if case == "small":
        nn = 9
        rp, ci, v = gen.elasticity3d_csr(nn)
        part, nparts = gen.box_partition_nodes(nn, (3, 3, 3))
        A = sp.csr_matrix((v, ci, rp), shape=(3 * nn ** 3, 3 * nn ** 3))
        lo, hi = 1, 96
    elif case == "mid":
        A, nparts, part = O.poisson3d(16), 2, None      # slabs of 8 x 16 x 16: band ~130
        lo, hi = 97, 448
    else:
        A, nparts, part = O.poisson3d(32), 2, None      # slabs of 16 x 32 x 32: band > 448
        lo, hi = 449, 4032
    X = np.random.default_rng(12).standard_normal((A.shape[0], 4))
    out = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("PREALPS_BJ_FACTOR", mode)
        prob, B, rowpos = _problem(A, nparts, part)
        try:
            out[mode] = prob.block_jacobi_apply(X, 4)
            assert lo <= prob.stat("bj_max_bandwidth") <= hi
        finally:
            prob.close()
    zr = O.BlockJacobi(B, rowpos).apply(X)
    for mode in out:
        np.testing.assert_allclose(out[mode], zr, rtol=1e-8, atol=1e-9 * np.abs(zr).max())
    np.testing.assert_allclose(out["device"], out["host"], rtol=1e-9, atol=1e-10 * np.abs(zr).max())


@pytest.mark.parametrize("n,P", [(8, 8), (16, 2)])
def test_indefinite_diagonal_block_is_reported(n, P, monkeypatch):
    """A non-positive pivot in either device factorisation kernel surfaces as the reference's
    "not SPD" failure."""
    import prealps_amd as pa
    from oracle import oracle as O
    monkeypatch.setenv("PREALPS_BJ_FACTOR", "device")
    Ab = sp.lil_matrix(O.poisson3d(n))
    Ab[100, 100] = -5.0
    prob, B, rowpos = _problem(sp.csr_matrix(Ab), P)
    try:
        with pytest.raises(pa.PreAlpsError, match="not SPD"):
            prob.create_block_jacobi()
    finally:
        prob.close()
