"""One MPI rank per GPU, the reference's own mode (examples/test_ecg_prealps_op.c:69,158): drivers
that only call MPI_Init and hand MPI_COMM_WORLD to preAlps_OperatorBuild.  On the one-GPU box both
ranks land on the same device, which RCCL refuses, so the library binds its host-staged MPI hooks
(mpi_glue.c) -- the panel distribution from rank 0, the exchanged send lists, the side-stream halo
exchange and the all-reduces all run for real; results are compared with the single-process oracle."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import prealps_amd
from prealps_amd import gen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
MPI_INC, MPI_LIB = "/opt/conda/include", "/opt/conda/lib"
have_mpi = os.path.exists(MPIEXEC) and os.path.exists(os.path.join(MPI_INC, "mpi.h"))
REF_BIN = os.path.join(ROOT, "tests", "_build", "test_ecg_prealps_op_mpi")


def _problem(n, nparts):
    from oracle import oracle as O
    from prealps_amd.solver import partition_kway
    rp, ci, v = gen.poisson3d_csr(n)
    part = partition_kway(rp, ci, nparts)          # what preAlps_OperatorBuild computes for the same file
    A = sp.csr_matrix((v, ci, rp), shape=(n ** 3, n ** 3))
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, nparts)
    return (rp, ci, v), B, perm, rowpos


def _write_mtx(path, rp, ci, v):
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = rows >= ci
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real symmetric\n%d %d %d\n" % (n, n, keep.sum()))
        for r, c, x in zip(rows[keep], ci[keep], v[keep]):
            f.write("%d %d %.17g\n" % (r + 1, c + 1, x))


@pytest.mark.skipif(not have_mpi, reason="no MPI launcher in this image")
@pytest.mark.parametrize("alg,t,red", [(0, 4, 0), (1, 2, 0), (0, 4, 1), (1, 4, 1)])
def test_mpi_driver_two_ranks_one_gpu_matches_oracle(tmp_path, alg, t, red):
    """examples/ecg_driver.c built against the system MPI: nothing in it but MPI_Init/Finalize and the
    reference's call sequence; iteration count, residual and the gathered solution against the oracle.
    red = 1: `-r 1`, D-Odir / BF-Omin through the driver's own loop on two ranks."""
    from oracle import oracle as O
    n, nparts, world = 12, 8, 2
    (rp, ci, v), B, perm, rowpos = _problem(n, nparts)
    mtx = str(tmp_path / "a.mtx")
    _write_mtx(mtx, rp, ci, v)
    prealps_amd.load()
    exe = str(tmp_path / "ecg_driver_mpi")
    subprocess.check_call(["gcc", "-std=gnu11", "-DPREALPS_USE_SYSTEM_MPI", "-I" + MPI_INC, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ecg_driver.c"), "-L" + os.path.join(ROOT, "prealps_amd"),
                           "-lprealps_hip", os.path.join(MPI_LIB, "libmpi.so.12"), "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu",
                           "-Wl,-rpath," + os.path.join(ROOT, "prealps_amd"), "-Wl,-rpath," + MPI_LIB, "-lm", "-o", exe])
    env = dict(os.environ, PREALPS_NPARTS=str(nparts), PREALPS_SETUP_TRACE="1", OMP_NUM_THREADS="4")
    r = subprocess.run([MPIEXEC, "-n", str(world), exe, "-m", mtx, "-e", str(t), "-o", str(alg), "-r", str(red),
                        "-x", str(tmp_path / "sol")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "hooks: mpi-host-staged" in r.stderr          # two ranks on one device: RCCL is not an option
    it = int(re.search(r"iter: (\d+)", r.stdout).group(1))
    res = float(re.search(r"res : (\S+)", r.stdout).group(1))
    ref = O.ECG(B, rowpos, t, O.ORTHODIR if alg == 0 else O.ORTHOMIN, O.ADAPT_BS if red else O.NO_BS_RED).solve(O.reference_rhs(rowpos))
    if red:
        assert int(re.search(r"bs  : (\d+)", r.stdout).group(1)) == ref["bs"][-1]
    assert it == ref["iters"], (it, ref["iters"])
    np.testing.assert_allclose(res, ref["final_res"], rtol=2e-6)      # (printed with 7 digits)
    x = np.concatenate([np.fromfile(str(tmp_path / "sol") + ".%d" % k) for k in range(world)])
    np.testing.assert_allclose(x, ref["x"], rtol=1e-7, atol=1e-10 * np.abs(ref["x"]).max())


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(not have_mpi, reason="no MPI launcher in this image")
def test_mpi_driver_two_ranks_two_gpus_native_rccl(tmp_path):
    """The same driver with one device per rank: the library must pick RCCL by itself (unique id broadcast
    over MPI_COMM_WORLD, ncclSend / ncclRecv / ncclAllReduce on the library stream).  Needs two devices."""
    if _device_count() < 2:
        pytest.skip("one device: RCCL refuses two ranks on it (the host-staged hooks are tested above)")
    from oracle import oracle as O
    n, nparts, world, t = 16, 16, 2, 4
    (rp, ci, v), B, perm, rowpos = _problem(n, nparts)
    mtx = str(tmp_path / "a.mtx")
    _write_mtx(mtx, rp, ci, v)
    prealps_amd.load()
    exe = str(tmp_path / "ecg_driver_mpi")
    subprocess.check_call(["gcc", "-std=gnu11", "-DPREALPS_USE_SYSTEM_MPI", "-I" + MPI_INC, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ecg_driver.c"), "-L" + os.path.join(ROOT, "prealps_amd"),
                           "-lprealps_hip", os.path.join(MPI_LIB, "libmpi.so.12"), "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu",
                           "-Wl,-rpath," + os.path.join(ROOT, "prealps_amd"), "-Wl,-rpath," + MPI_LIB, "-lm", "-o", exe])
    env = dict(os.environ, PREALPS_NPARTS=str(nparts), PREALPS_SETUP_TRACE="1", OMP_NUM_THREADS="4",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([MPIEXEC, "-n", str(world), exe, "-m", mtx, "-e", str(t), "-o", "0", "-x", str(tmp_path / "sol")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "hooks: rccl" in r.stderr
    it = int(re.search(r"iter: (\d+)", r.stdout).group(1))
    ref = O.ECG(B, rowpos, t).solve(O.reference_rhs(rowpos))
    assert it == ref["iters"], (it, ref["iters"])
    x = np.concatenate([np.fromfile(str(tmp_path / "sol") + ".%d" % k) for k in range(world)])
    np.testing.assert_allclose(x, ref["x"], rtol=1e-7, atol=1e-10 * np.abs(ref["x"]).max())


@pytest.mark.skipif(not have_mpi, reason="no MPI launcher in this image")
@pytest.mark.parametrize("alg", [0, 1])
def test_rccl_binding_three_ranks_one_gpu_through_a_stand_in(tmp_path, alg):
    """The `rccl` branch of pa_mpi_bind with more than one rank: unique id from rank 0 broadcast over MPI,
    ncclCommInitRank on every rank, the self-test, then a whole solve whose halo exchange is a grouped
    ncclSend / ncclRecv with TWO peers on the middle rank and whose sums are ncclAllReduce calls on the library's
    stream.  The real RCCL refuses ranks that share a device, so PREALPS_RCCL_LIB points the binding
    (comm_rccl.hip) at tests/c/rccl_standin.c -- test infrastructure that stages every call through the host and
    MPI.  What is checked is the binding and the library's call sequence, not RCCL."""
    from oracle import oracle as O
    n, nparts, world, t = 12, 9, 3, 4
    (rp, ci, v), B, perm, rowpos = _problem(n, nparts)
    mtx = str(tmp_path / "a.mtx")
    _write_mtx(mtx, rp, ci, v)
    prealps_amd.load()
    standin = str(tmp_path / "librccl_standin.so")
    subprocess.check_call(["gcc", "-w", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + MPI_INC,
                           os.path.join(ROOT, "tests", "c", "rccl_standin.c"), "-o", standin, "-L/opt/rocm/lib", "-lamdhip64",
                           os.path.join(MPI_LIB, "libmpi.so.12"), "-Wl,-rpath," + MPI_LIB, "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu"])
    exe = str(tmp_path / "ecg_driver_mpi")
    subprocess.check_call(["gcc", "-std=gnu11", "-DPREALPS_USE_SYSTEM_MPI", "-I" + MPI_INC, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ecg_driver.c"), "-L" + os.path.join(ROOT, "prealps_amd"),
                           "-lprealps_hip", os.path.join(MPI_LIB, "libmpi.so.12"), "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu",
                           "-Wl,-rpath," + os.path.join(ROOT, "prealps_amd"), "-Wl,-rpath," + MPI_LIB, "-lm", "-o", exe])
    env = dict(os.environ, PREALPS_NPARTS=str(nparts), PREALPS_SETUP_TRACE="1", OMP_NUM_THREADS="4",
               PREALPS_RCCL_LIB=standin, PREALPS_COMM="rccl")
    r = subprocess.run([MPIEXEC, "-n", str(world), exe, "-m", mtx, "-e", str(t), "-o", str(alg), "-x", str(tmp_path / "sol")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stderr.count("hooks: rccl") == world, r.stderr[-3000:]
    it = int(re.search(r"iter: (\d+)", r.stdout).group(1))
    ref = O.ECG(B, rowpos, t, ortho_alg=O.ORTHODIR if alg == 0 else O.ORTHOMIN).solve(O.reference_rhs(rowpos))
    assert it == ref["iters"], (it, ref["iters"])
    x = np.concatenate([np.fromfile(str(tmp_path / "sol") + ".%d" % k) for k in range(world)])
    np.testing.assert_allclose(x, ref["x"], rtol=1e-7, atol=1e-10 * np.abs(ref["x"]).max())
    # the stand-in's own count, printed when the communicator goes: the middle rank talks to two peers
    stats = {int(m.group(1)): tuple(int(g) for g in m.groups()[1:])
             for m in re.finditer(r"\[rccl stand-in\] rank (\d+): (\d+) all-reduces, (\d+) sends, (\d+) receives, (\d+) groups", r.stderr)}
    if stats:
        assert stats[1][0] >= 2 * it and stats[1][3] >= it          # >= two sums and one exchange per iteration
        # (every group but the self-test's ring step sends to and receives from both neighbours)
        assert stats[1][1] >= 2 * (stats[1][3] - 1) and stats[1][2] >= 2 * (stats[1][3] - 1)


@pytest.mark.skipif(not (have_mpi and os.path.exists(REF_BIN)), reason="needs MPI and the prebuilt reference driver (oracle/Makefile)")
def test_unmodified_reference_driver_two_ranks_one_gpu(tmp_path):
    """The reference's examples/test_ecg_prealps_op.c, compiled unchanged where the reference tree
    was present (tests/_build/, built by oracle/Makefile) and started with two ranks on this GPU."""
    from oracle import oracle as O
    n, nparts, world, t = 12, 8, 2, 4
    (rp, ci, v), B, perm, rowpos = _problem(n, nparts)
    mtx = str(tmp_path / "a.mtx")
    _write_mtx(mtx, rp, ci, v)
    env = dict(os.environ, PREALPS_NPARTS=str(nparts), PREALPS_SETUP_TRACE="1", OMP_NUM_THREADS="4")
    r = subprocess.run([MPIEXEC, "-n", str(world), REF_BIN, "-m", mtx, "-e", str(t), "-o", "0", "-r", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    it = int(re.search(r"iter: (\d+)", r.stdout).group(1))
    res = float(re.search(r"res : (\S+)", r.stdout).group(1))
    # the driver's rhs (its lines 172-184): every rank draws its m values from srand(0), the norm is
    # global, element 0 of every rank stays unscaled
    libc = C.CDLL("libc.so.6")
    rhs = np.zeros(n ** 3)
    bounds = [int(rowpos[k * nparts // world]) for k in range(world + 1)]
    for k in range(world):
        libc.srand(0)
        m = bounds[k + 1] - bounds[k]
        rhs[bounds[k]:bounds[k + 1]] = [libc.rand() / 2147483647.0 for _ in range(m)]
    normb = np.sqrt((rhs ** 2).sum())
    for k in range(world):
        rhs[bounds[k] + 1:bounds[k + 1]] /= normb
    ref = O.ECG(B, rowpos, t).solve(rhs)
    assert it == ref["iters"], (it, ref["iters"])
    np.testing.assert_allclose(res, ref["final_res"], rtol=2e-6)      # (printed with 7 digits)
