"""The reference is an MPI program (examples/test_ecg_prealps_op.c:69,158 -> utils/operator.c:38-134):
preAlps_OperatorBuild takes rank and size from the communicator it is handed, rank 0 reads and
partitions, every other rank receives its row panel.  These tests run that under mpiexec on the CPU
(plan-only mode: no GPU): every rank must hold only its rows, and exactly the panel, halo plan and
ordering the replicated build (every process given the whole matrix) produces."""
import ctypes as C
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
MPI_INC, MPI_LIB = "/opt/conda/include", "/opt/conda/lib"
have_mpi = os.path.exists(MPIEXEC) and os.path.exists(os.path.join(MPI_INC, "mpi.h"))
REF_DRIVER = "/root/reference/examples/test_ecg_prealps_op.c"


def _cc(src, exe, extra=()):
    prealps_amd.load()
    subprocess.check_call(["gcc", "-std=gnu99", "-w", "-DPREALPS_USE_SYSTEM_MPI", "-I" + MPI_INC, *extra,
                           "-I" + os.path.join(ROOT, "include"), src, "-L" + os.path.join(ROOT, "prealps_amd"),
                           "-lprealps_hip", os.path.join(MPI_LIB, "libmpi.so.12"),
                           # (conda ships an older libstdc++ next to its MPI: the HIP runtime must keep the system's)
                           "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath," + os.path.join(ROOT, "prealps_amd"),
                           "-Wl,-rpath," + MPI_LIB, "-lm", "-o", exe])


def write_mtx(path, rp, ci, v, symmetric=True):
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = rows >= ci if symmetric else np.ones(len(ci), bool)
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real %s\n%d %d %d\n" % ("symmetric" if symmetric else "general", n, n, keep.sum()))
        for r, c, x in zip(rows[keep], ci[keep], v[keep]):
            f.write("%d %d %.17g\n" % (r + 1, c + 1, x))


def _read_dump(path):
    raw = open(path, "rb").read()
    pos = 0
    out = []
    kinds = "iiidiiiiiiid"       # head rowPtr colInd val rowPos peers srows rrows sidx hcols perm rhs
    for k in kinds:
        (n,) = struct.unpack_from("i", raw, pos)
        pos += 4
        dt = np.int32 if k == "i" else np.float64
        out.append(np.frombuffer(raw, dtype=dt, count=n, offset=pos).copy())
        pos += n * (4 if k == "i" else 8)
    assert pos == len(raw)
    return out


def _replicated(path, rank, size, env):
    """The same build with the whole matrix in one process that is told it is rank `rank` of `size`."""
    code = r"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, %r)
import prealps_amd
from prealps_amd.lib import check
L = prealps_amd.load()
L.preAlps_hip_plan_only(1)
check(L.preAlps_hip_set_world(%d, %d), "world")
check(L.preAlps_OperatorBuild(%r.encode(), 0x44000000), "build")
M, m = C.c_int(), C.c_int(); check(L.preAlps_OperatorGetSizes(C.byref(M), C.byref(m)), "sizes")
A = prealps_amd.CPLM_Mat_CSR_t(); check(L.preAlps_OperatorGetA(C.byref(A)), "A")
pi = C.POINTER(C.c_int)
rowPos, n = pi(), C.c_int(); check(L.preAlps_OperatorGetRowPosPtr(C.byref(rowPos), C.byref(n)), "rp")
npeers, nsend, nhalo = C.c_int(), C.c_int(), C.c_int()
peers, srows, rrows, sidx, hcols = pi(), pi(), pi(), pi(), pi()
check(L.preAlps_OperatorGetHaloPlan(C.byref(npeers), C.byref(peers), C.byref(srows), C.byref(rrows), C.byref(sidx), C.byref(nsend), C.byref(hcols), C.byref(nhalo)), "halo")
perm, nperm = pi(), C.c_int(); check(L.preAlps_OperatorGetPermPtr(C.byref(perm), C.byref(nperm)), "perm")
arr = lambda p, k, t=np.int32: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(t).copy() if k else np.zeros(0, t)
rhs = np.zeros(m.value); check(L.preAlps_hip_reference_rhs(rhs.ctypes.data_as(C.POINTER(C.c_double))), "rhs")
np.savez(%r, head=np.array([%d, %d, M.value, m.value, A.info.lnnz, A.info.nnz]), rowPtr=arr(A.rowPtr, m.value + 1),
         colInd=arr(A.colInd, A.info.lnnz), val=arr(A.val, A.info.lnnz, np.float64), rowPos=arr(rowPos, n.value),
         peers=arr(peers, npeers.value), srows=arr(srows, npeers.value), rrows=arr(rrows, npeers.value),
         sidx=arr(sidx, nsend.value), hcols=arr(hcols, nhalo.value), perm=arr(perm, nperm.value), rhs=rhs)
""" % (ROOT, rank, size, path, path + ".rep%d.npz" % rank, rank, size)
    subprocess.check_call(["python", "-c", code], env=env)
    return np.load(path + ".rep%d.npz" % rank)


@pytest.mark.skipif(not have_mpi, reason="no MPI launcher in this image")
@pytest.mark.parametrize("size,nparts,kind", [(2, 2, "poisson"), (3, 12, "poisson"), (2, 9, "unsym")])
def test_mpi_ranks_hold_their_panel_and_the_replicated_plan(tmp_path, size, nparts, kind):
    n = 10
    rp, ci, v = gen.poisson3d_csr(n)
    sym = True
    if kind == "unsym":          # `general` file with a one-sided pattern: the exchanged send lists must still be exact
        import scipy.sparse as sp
        A0 = sp.coo_matrix(sp.csr_matrix((v, ci, rp), shape=(n ** 3, n ** 3)))
        keep = ~((A0.row > A0.col) & ((A0.row + A0.col) % 3 == 0))
        A0 = sp.csr_matrix((A0.data[keep] * (1.0 + 0.01 * (A0.row[keep] % 7)), (A0.row[keep], A0.col[keep])), shape=A0.shape)
        A0.sort_indices()
        rp, ci, v = A0.indptr.astype(np.int32), A0.indices.astype(np.int32), A0.data.copy()
        sym = False
    mtx = str(tmp_path / "a.mtx")
    write_mtx(mtx, rp, ci, v, symmetric=sym)
    exe = str(tmp_path / "mpi_plan_dump")
    _cc(os.path.join(ROOT, "tests", "c", "mpi_plan_dump.c"), exe)
    env = dict(os.environ, PREALPS_NPARTS=str(nparts), OMP_NUM_THREADS="2")
    r = subprocess.run([MPIEXEC, "-n", str(size), exe, mtx, str(tmp_path / "dump")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    names = ["head", "rowPtr", "colInd", "val", "rowPos", "peers", "srows", "rrows", "sidx", "hcols", "perm", "rhs"]
    rows_seen = 0
    for rank in range(size):
        got = dict(zip(names, _read_dump(str(tmp_path / "dump") + ".%d" % rank)))
        ref = _replicated(mtx, rank, size, env)
        for k in names:
            if k == "val":
                np.testing.assert_array_equal(got[k], ref[k])     # same arithmetic, same order: bitwise
            else:
                np.testing.assert_array_equal(got[k], ref[k], err_msg="%s of rank %d" % (k, rank))
        m = int(got["head"][3])
        assert len(got["rowPtr"]) == m + 1 and got["head"][4] == got["rowPtr"][-1] < got["head"][5]   # only its rows
        rows_seen += m
    assert rows_seen == n ** 3


@pytest.mark.skipif(not (have_mpi and os.path.exists(REF_DRIVER)), reason="needs MPI and the reference tree")
def test_unmodified_reference_driver_plans_under_mpiexec(tmp_path):
    """The reference's own driver, compiled unchanged against the system MPI, started with two ranks:
    both ranks get through preAlps_OperatorBuild (rank and size from MPI_COMM_WORLD, panels from rank 0)
    and then stop loudly at the first call that needs the GPU -- there is no CPU path."""
    rp, ci, v = gen.poisson3d_csr(8)
    mtx = str(tmp_path / "a.mtx")
    write_mtx(mtx, rp, ci, v)
    exe = str(tmp_path / "ref_driver_mpi")
    _cc(REF_DRIVER, exe, extra=("-I" + os.path.join(ROOT, "include", "compat"),))
    env = dict(os.environ, PREALPS_PLAN_ONLY="1", PREALPS_SETUP_TRACE="1", OMP_NUM_THREADS="2")
    r = subprocess.run([MPIEXEC, "-n", "2", exe, "-m", mtx, "-e", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank 0: panels sent" in r.stderr and "panel received" in r.stderr and "peer lists (exchanged)" in r.stderr
    assert "ABORTING from" in r.stderr


# ---- failure semantics (reference: every fatal error -> MPI_Abort(MPI_COMM_WORLD, 1), utils/cplm_core/cplm_utils.c:42-58) ----
@pytest.mark.skipif(not have_mpi, reason="no MPI launcher in this image")
@pytest.mark.parametrize("abort_mode", [1, 0])
def test_failure_on_rank_zero_ends_every_rank(tmp_path, abort_mode):
    """Only rank 0 reads the matrix; here the file does not exist.  Abort mode (default): the banner, then
    MPI_Abort takes all three ranks down (no rank stays inside the broadcast).  Return-code mode: all three
    ranks come back from preAlps_OperatorBuild with 1 and finish normally."""
    exe = str(tmp_path / "mpi_fail_probe")
    _cc(os.path.join(ROOT, "tests", "c", "mpi_fail_probe.c"), exe)
    r = subprocess.run([MPIEXEC, "-n", "3", exe, str(tmp_path / "nothing_here.mtx"), str(abort_mode)],
                       env=dict(os.environ, OMP_NUM_THREADS="2", PREALPS_MPI_TIMEOUT="60"), capture_output=True, text=True, timeout=120)
    if abort_mode:
        assert r.returncode != 0
        assert "ABORTING from" in r.stderr and "[Proc: 0]" in r.stderr
        assert "rc" not in r.stdout                     # nobody got as far as the print
    else:
        assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
        assert sorted(r.stdout.split("\n")[:3]) == ["rank 0: rc 1", "rank 1: rc 1", "rank 2: rc 1"]


def test_open_mpi_launcher_is_refused_in_return_code_mode(tmp_path):
    """Under an Open MPI launcher (its communicators are pointers, the run-time binding speaks the MPICH ABI)
    preAlps_OperatorBuild must fail -- also when the library returns error codes instead of aborting: going on
    would make every rank build and solve the whole problem alone."""
    rp, ci, v = gen.poisson3d_csr(4)
    mtx = str(tmp_path / "a.mtx")
    write_mtx(mtx, rp, ci, v)
    code = r"""
import sys
sys.path.insert(0, %r)
import prealps_amd
L = prealps_amd.load()
L.preAlps_hip_plan_only(1)
L.preAlps_hip_set_abort_mode(0)
rc = L.preAlps_OperatorBuild(%r.encode(), 0x44000000)
print("rc", rc, L.preAlps_hip_last_error().decode()[:60])
""" % (ROOT, mtx)
    r = subprocess.run([os.sys.executable, "-c", code], env=dict(os.environ, OMPI_COMM_WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("rc 1 ") and "Open MPI" in r.stdout, r.stdout


@pytest.mark.skipif(not have_mpi, reason="no MPI launcher in this image")
def test_prealps_mpi_off_leaves_the_process_group_to_the_caller(tmp_path):
    """PREALPS_MPI=0: the library never looks for an MPI -- for a caller that starts its ranks with an MPI launcher
    but describes the process group itself (preAlps_hip_set_world / set_comm).  Under mpiexec -n 2 and without
    such a description every rank is then a process group of one and holds all the rows."""
    n = 6
    rp, ci, v = gen.poisson3d_csr(n)
    mtx = str(tmp_path / "a.mtx")
    write_mtx(mtx, rp, ci, v)
    exe = str(tmp_path / "mpi_plan_dump")
    _cc(os.path.join(ROOT, "tests", "c", "mpi_plan_dump.c"), exe)
    env = dict(os.environ, PREALPS_NPARTS="4", OMP_NUM_THREADS="2", PREALPS_MPI="0")
    r = subprocess.run([MPIEXEC, "-n", "2", exe, mtx, str(tmp_path / "dump")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for rank in range(2):
        head = _read_dump(str(tmp_path / "dump") + ".%d" % rank)[0]
        assert head[2] == head[3] == n ** 3            # M == m: the whole matrix


def test_partition_file_is_taken_as_it_is(tmp_path):
    """PREALPS_PARTITION_FILE (one part id per line, e.g. METIS output -- the reference calls METIS_PartGraphKway
    at utils/cplm_core/cplm_matcsr_core.c:394-457): preAlps_OperatorBuild orders the rows by exactly those parts."""
    n, nparts = 6, 5
    rp, ci, v = gen.poisson3d_csr(n)
    mtx = str(tmp_path / "a.mtx")
    write_mtx(mtx, rp, ci, v)
    part = (np.arange(n ** 3) * 7 + 3) % nparts                      # nothing a partitioner would produce
    pf = str(tmp_path / "parts.txt")
    np.savetxt(pf, part, fmt="%d")
    ref = _replicated(mtx, 0, 1, dict(os.environ, PREALPS_NPARTS=str(nparts), PREALPS_PARTITION_FILE=pf, OMP_NUM_THREADS="2"))
    np.testing.assert_array_equal(np.diff(ref["rowPos"]), np.bincount(part, minlength=nparts))
    np.testing.assert_array_equal(part[ref["perm"]], np.repeat(np.arange(nparts), np.bincount(part, minlength=nparts)))
