"""The MatrixMarket reader of preAlps_OperatorBuild (operator.c: mapped file, parsed by the host threads in
line-aligned pieces) against scipy on the cases of utils/cplm_light/cplm_matcsr.c:96-243: general and
symmetric files, 1- and 0-based indices, repeated entries (summed), blank lines, CRLF line ends, a last
line without newline, and a file large enough that every thread gets a piece."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

import prealps_amd
from prealps_amd.lib import check


def _load(path, nparts=1):
    """Build in plan-only mode (no GPU) with one contiguous part and no scaling effects on the pattern:
    returns the permuted, scaled panel; with nparts = 1 the permutation is the identity."""
    L = prealps_amd.load()
    L.preAlps_hip_plan_only(1)
    os.environ["PREALPS_NPARTS"] = str(nparts)
    os.environ["PREALPS_PARTITION"] = "contiguous"
    try:
        check(L.preAlps_OperatorBuild(path.encode(), 0x44000000), "build")
        A = prealps_amd.CPLM_Mat_CSR_t()
        check(L.preAlps_OperatorGetA(C.byref(A)), "A")
        m, nnz = A.info.m, A.info.lnnz
        rp = np.ctypeslib.as_array(A.rowPtr, shape=(m + 1,)).copy()
        ci = np.ctypeslib.as_array(A.colInd, shape=(nnz,)).copy()
        v = np.ctypeslib.as_array(A.val, shape=(nnz,)).copy()
        return sp.csr_matrix((v, ci, rp), shape=(m, m))
    finally:
        L.preAlps_OperatorFree()
        L.preAlps_hip_plan_only(0)
        os.environ.pop("PREALPS_PARTITION", None)
        os.environ.pop("PREALPS_NPARTS", None)


def _scaled(A):
    A = sp.csr_matrix(A)
    d = 1.0 / np.sqrt(np.abs(A).max(axis=1).toarray().ravel())
    return sp.diags(d) @ A @ sp.diags(d)


def _spd(n, seed, density=0.01):
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="csr")
    A = M + M.T
    return sp.csr_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0))


@pytest.mark.parametrize("style", ["general", "symmetric", "zero_based", "crlf_blank_no_final_newline", "duplicates", "comments_in_body"])
def test_reader_matches_scipy(tmp_path, style):
    A = _spd(300, 1)
    coo = sp.coo_matrix(A)
    rows, cols, vals = coo.row, coo.col, coo.data
    sym, base, eol = False, 1, "\n"
    if style == "symmetric":
        keep = rows >= cols
        rows, cols, vals, sym = rows[keep], cols[keep], vals[keep], True
    if style == "zero_based":
        base = 0
        order = np.lexsort((cols, rows))          # the reference decides on the FIRST entry: make it (0, 0)
        rows, cols, vals = rows[order], cols[order], vals[order]
    if style == "crlf_blank_no_final_newline":
        eol = "\r\n"
    if style == "duplicates":                      # every entry split into two halves that must be summed
        rows, cols, vals = np.concatenate([rows, rows]), np.concatenate([cols, cols]), np.concatenate([0.25 * vals, 0.75 * vals])
    path = str(tmp_path / "a.mtx")
    with open(path, "w", newline="") as f:
        f.write("%%%%MatrixMarket matrix coordinate real %s%s" % ("symmetric" if sym else "general", eol))
        f.write("% a comment" + eol)
        f.write("%d %d %d%s" % (A.shape[0], A.shape[1], len(vals), eol))
        lines = ["%d %d %.17g" % (r + base, c + base, x) for r, c, x in zip(rows, cols, vals)]
        if style == "comments_in_body":            # '%' lines between the entries: skipped, not counted
            lines.insert(3, "% a remark in the middle")
            lines.insert(40, "  % and an indented one")
            lines.append("% and one at the end")
        if style == "crlf_blank_no_final_newline":
            lines.insert(5, "")
            f.write(eol.join(lines))
        else:
            f.write(eol.join(lines) + eol)
    got = _load(path)
    ref = _scaled(A)
    assert abs(got - ref).max() <= 1e-15 * abs(ref).max()
    assert got.nnz == ref.nnz


def test_reader_on_a_file_that_every_thread_shares(tmp_path):
    n = 20000
    A = _spd(n, 7, density=0.0008)               # ~ 340 k entries: more than 4096 lines per thread
    coo = sp.coo_matrix(sp.tril(A))
    perm = np.random.default_rng(3).permutation(coo.nnz)     # entries in no particular order
    path = str(tmp_path / "big.mtx")
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real symmetric\n%d %d %d\n" % (n, n, coo.nnz))
        np.savetxt(f, np.column_stack([coo.row[perm] + 1, coo.col[perm] + 1, coo.data[perm]]), fmt="%d %d %.17g")
    got = _load(path)
    ref = _scaled(A)
    assert got.nnz == ref.nnz and abs(got - ref).max() <= 1e-15 * abs(ref).max()


def test_reader_reports_bad_files(tmp_path):
    L = prealps_amd.load()
    L.preAlps_hip_plan_only(1)
    try:
        for body, msg in (("%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1.0\n2 2 1.0\n", b"bad entry"),
                          ("%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1.0\n2 x 1.0\n3 3 1.0\n", b"bad entry"),
                          ("%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1.0\n2 2 1.0\n4 3 1.0\n", b"out of range"),
                          ("%%MatrixMarket matrix array real general\n3 3\n", b"Only sparse real")):
            p = str(tmp_path / "bad.mtx")
            open(p, "w").write(body)
            assert L.preAlps_OperatorBuild(p.encode(), 0x44000000) != 0
            assert msg in L.preAlps_hip_last_error(), L.preAlps_hip_last_error()
    finally:
        L.preAlps_hip_plan_only(0)
