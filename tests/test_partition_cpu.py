"""The library's k-way graph partitioner (prealps_amd/csrc/partition.c), which stands where the
reference calls METIS_PartGraphKway (utils/cplm_core/cplm_matcsr_core.c:394-457).  Host code:
runs without a GPU.  METIS' output is not unique, so what is checked are the properties the ECG
path relies on: every part non-empty, balance, compactness (edge cut against the geometric box
partition of the same grid), the dofs of a node staying together, locality of the part
numbering, and independence from the numbering of the input matrix."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.csgraph as csg

import prealps_amd
from prealps_amd import gen
from prealps_amd.solver import partition_kway

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cut(rp, ci, part):
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    return int((part[rows] != part[ci]).sum())


def _owner_halo(rp, ci, part, k, G):
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    owner = (part.astype(np.int64) * G) // k
    return sum(len(np.unique(ci[(owner[rows] == g) & (owner[ci] != g)])) for g in range(G))


def test_poisson_grid_many_small_parts():
    n, k = 24, 110
    rp, ci, v = gen.poisson3d_csr(n)
    part = partition_kway(rp, ci, k)
    sizes = np.bincount(part, minlength=k)
    assert part.min() == 0 and part.max() == k - 1 and sizes.min() > 0
    avg = n ** 3 / k
    assert sizes.max() <= 1.15 * avg and sizes.min() >= 0.85 * avg, (sizes.min(), avg, sizes.max())
    pb, kb = gen.box_partition(n, (4, 4, 8))          # 108 boxes of 128 nodes
    assert _cut(rp, ci, part) <= 1.5 * _cut(rp, ci, pb)
    # parts are connected pieces of the grid
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n ** 3, n ** 3))
    for p in range(0, k, 7):
        idx = np.flatnonzero(part == p)
        assert csg.connected_components(A[idx][:, idx], directed=False)[0] == 1
    # contiguous ranges of part ids (what one GPU owns) are compact regions
    assert _owner_halo(rp, ci, part, k, 4) <= 1.5 * _owner_halo(rp, ci, pb, kb, 4)
    assert np.array_equal(part, partition_kway(rp, ci, k))      # deterministic


def test_vector_problem_keeps_the_dofs_of_a_node_together():
    nn, k = 12, 40
    rp, ci, v = gen.elasticity3d_csr(nn)
    part = partition_kway(rp, ci, k)
    assert np.array_equal(part[0::3], part[1::3]) and np.array_equal(part[0::3], part[2::3])
    sizes = np.bincount(part, minlength=k)
    avg = 3 * nn ** 3 / k
    assert sizes.min() > 0 and sizes.max() <= 1.2 * avg and sizes.min() >= 0.8 * avg
    pb, kb = gen.box_partition_nodes(nn, (3, 4, 4))
    assert _cut(rp, ci, part) <= 1.5 * _cut(rp, ci, pb)


def test_quality_does_not_depend_on_the_row_numbering():
    """The same grid after a random symmetric permutation (no geometry left in the ids)."""
    n, k = 20, 64
    rp, ci, v = gen.poisson3d_csr(n)
    N = n ** 3
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))
    q = np.random.default_rng(3).permutation(N)
    B = A[q][:, q].tocsr()
    B.sort_indices()
    part = partition_kway(B.indptr, B.indices, k)
    ref = partition_kway(rp, ci, k)
    sizes = np.bincount(part, minlength=k)
    assert sizes.min() >= 0.85 * N / k and sizes.max() <= 1.15 * N / k
    assert _cut(B.indptr, B.indices, part) <= 1.15 * _cut(rp, ci, ref)


def test_more_components_than_parts_and_tiny_graphs():
    # 10 disconnected chains of 30 vertices, 4 parts
    blocks = [sp.diags([np.ones(29), 2 * np.ones(30), np.ones(29)], [-1, 0, 1])] * 10
    A = sp.block_diag(blocks, format="csr")
    A.sort_indices()
    part = partition_kway(A.indptr, A.indices, 4)
    sizes = np.bincount(part, minlength=4)
    assert sizes.min() > 0 and sizes.sum() == 300
    # a star (one dense row and column): every vertex but the centre is a fragment of its part
    n = 2000
    rows = np.concatenate([np.arange(n), np.zeros(n - 1, int), np.arange(1, n)])
    cols = np.concatenate([np.arange(n), np.arange(1, n), np.zeros(n - 1, int)])
    S = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    S.sort_indices()
    sizes = np.bincount(partition_kway(S.indptr, S.indices, 16), minlength=16)
    assert sizes.min() >= 0.7 * n / 16 and sizes.max() <= 1.3 * n / 16
    # no edges at all
    sizes = np.bincount(partition_kway(sp.identity(1000, format="csr").indptr, np.arange(1000, dtype=np.int32), 10))
    assert sizes.min() == sizes.max() == 100
    # the reference's own 14 x 14 fixture, 2 parts
    from oracle import oracle as O
    L = O.load_mtx(os.path.join(ROOT, "tests", "golden", "LFAT5.mtx"))
    part = partition_kway(L.indptr, L.indices, 2)
    assert sorted(np.bincount(part, minlength=2)) in ([6, 8], [7, 7], [5, 9])
    # as many parts as rows
    part = partition_kway(L.indptr, L.indices, 14)
    assert sorted(part) == list(range(14))
    with pytest.raises(prealps_amd.PreAlpsError):
        partition_kway(L.indptr, L.indices, 15)


def test_same_parts_for_every_thread_count():
    """The halves of a cut are bisected as OpenMP tasks: the part vector must not depend on how many
    threads there are (every rank of a multi-GPU run computes the partition on its own and they must
    agree), nor on the schedule (repeated runs)."""
    import hashlib
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import hashlib\n"
        "from prealps_amd import gen\n"
        "from prealps_amd.solver import partition_kway\n"
        "rp, ci, v = gen.elasticity3d_csr(24)\n"
        "print(hashlib.md5(partition_kway(rp, ci, 300).tobytes()).hexdigest())\n"
        "rp, ci, v = gen.poisson3d_csr(36)\n"
        "print(hashlib.md5(partition_kway(rp, ci, 77).tobytes()).hexdigest())\n") % ROOT
    seen = set()
    for threads in ("1", "3", "8", "8"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, OMP_NUM_THREADS=threads))
        assert r.returncode == 0, r.stderr[-800:]
        seen.add(r.stdout)
    assert len(seen) == 1, seen
