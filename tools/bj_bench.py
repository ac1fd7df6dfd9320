#!/usr/bin/env python3
"""Time the block-Jacobi apply and the SpMM of one problem (HIP events on the library stream).
usage: bj_bench.py [elasticity|poisson] n box t [nparts-for-kway]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check

wl, n, box, t = sys.argv[1], int(sys.argv[2]), tuple(int(x) for x in sys.argv[3].split(",")), int(sys.argv[4])
kway = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if wl == "poisson":
    rp, ci, v = gen.poisson3d_csr(n); part, P = gen.box_partition(n, box)
else:
    rp, ci, v = gen.elasticity3d_csr(n); part, P = gen.box_partition_nodes(n, box)
if os.environ.get("BJ_BENCH_SHUFFLE") == "nodes" and wl != "poisson":    # renumber the nodes at random (dof triples kept)
    import scipy.sparse as sp
    N = len(rp) - 1
    pn = np.random.default_rng(3).permutation(N // 3)
    pm = (3 * pn[:, None] + np.arange(3)[None, :]).ravel()
    A = sp.csr_matrix((v, ci, rp), shape=(N, N))[pm][:, pm]
    A.sort_indices()
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
t0 = time.time()
if kway:
    from prealps_amd.solver import partition_kway
    part, P = partition_kway(rp, ci, kway), kway
tpart = time.time() - t0
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
L = prob.L
t0 = time.time(); prob.create_block_jacobi(); tbj = time.time() - t0
check(L.preAlps_hip_prepare_operator(t), "prep")
m = prob.m
X = np.random.default_rng(0).standard_normal((m, t))
dx, dy = prob.panel(t, t), prob.panel(t, t)
prob.to_device(dx, X, t)
sec = C.c_double()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    check(L.preAlps_hip_timer_start(), "ts")
    for _ in range(reps): fn()
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
    return 1e6 * sec.value / reps
bj = timeit(lambda: check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj"))
sp_ = timeit(lambda: check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "spmm"))
def alt():
    check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj"); check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "spmm")
both = timeit(alt)
print("%s n=%d box=%s t=%d parts=%d%s env=%s: bj %.1f us, spmm %.1f us, alternating pair %.1f us; band %d, factor %.0f MB, bj setup %.2fs, partition %.2fs" % (
    wl, n, box, t, P, " (kway)" if kway else "", {k: os.environ[k] for k in os.environ if k.startswith("PREALPS_")}, bj, sp_, both,
    prob.stat("bj_max_bandwidth"), prob.stat("bj_factor_bytes") / 1e6, tbj, tpart), flush=True)
print("   spmm plan: runs %d staged %d stream bytes %.0f MB, blocks %d, halo rows %d" % (prob.stat("spmm_runs"), prob.stat("spmm_staged"), prob.stat("spmm_stream_bytes") / 1e6, prob.stat("spmm_blocks"), prob.stat("halo_rows")), flush=True)
prob.close()
