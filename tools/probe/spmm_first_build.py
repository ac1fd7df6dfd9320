#!/usr/bin/env python3
"""Is the first operator of a process slower than a rebuilt one?  Builds the headline operator N times in one
process (everything freed in between) and times the SpMM each time.  SPMM_FIRST_DUMMY_MB: allocate and free
that much device memory before the first build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check
t = 4
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
X = np.random.default_rng(0).standard_normal((len(rp) - 1, t))
dummy = int(os.environ.get("SPMM_FIRST_DUMMY_MB", "0"))
if dummy:
    import torch
    z = torch.empty(dummy << 20, dtype=torch.uint8, device="cuda"); z.fill_(1); torch.cuda.synchronize(); del z; torch.cuda.empty_cache()
leak_mb = [int(x) for x in os.environ.get("SPMM_FIRST_LEAK_MB", "").split(",") if x]      # leaked before build k (k >= 1)
leak_where = os.environ.get("SPMM_FIRST_LEAK_WHERE", "before")                          # before the build / before the panels
keep = []
import torch
for k in range(int(os.environ.get("SPMM_FIRST_BUILDS", "5"))):
    if k >= 1 and leak_mb and leak_where == "before": keep.append(torch.empty(leak_mb[(k - 1) % len(leak_mb)] << 20, dtype=torch.uint8, device="cuda"))
    prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
    L = prob.L
    prob.create_block_jacobi()
    check(L.preAlps_hip_prepare_operator(t), "prep")
    if k >= 1 and leak_mb and leak_where == "panels": keep.append(torch.empty(leak_mb[(k - 1) % len(leak_mb)] << 20, dtype=torch.uint8, device="cuda"))
    dx, dy = prob.panel(t, t), prob.panel(t, t)
    prob.to_device(dx, X, t)
    sec = C.c_double(); tot = 0.0; totb = 0.0
    for i in range(25):
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 5: totb += sec.value
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 5: tot += sec.value
    print("build %d: spmm %.1f us  block solve %.1f us  val %#x slot %#x x %#x y %#x" % (k, 1e6 * tot / 20, 1e6 * totb / 20, int(prob.stat("spmm_val_address")),
          int(prob.stat("spmm_slot_address")), C.cast(dx.val, C.c_void_p).value, C.cast(dy.val, C.c_void_p).value), flush=True)
    prob.panel_free(dx); prob.panel_free(dy)
    prob.close()
