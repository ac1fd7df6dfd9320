#!/usr/bin/env python3
"""One process: build the headline operator, time the SpMM in the solver's cache state, print the time
next to the device addresses of the matrix arrays and panels (run-to-run spread study)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check
n, t = 70, 4
rp, ci, v = gen.elasticity3d_csr(n); part, P = gen.box_partition_nodes(n, (2, 4, 8))
prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
L = prob.L
prob.create_block_jacobi()
check(L.preAlps_hip_prepare_operator(t), "prep")
X = np.random.default_rng(0).standard_normal((prob.m, t))
dx, dy = prob.panel(t, t), prob.panel(t, t)
prob.to_device(dx, X, t)
sec = C.c_double(); tot = 0.0
for i in range(40):
    check(L.preAlps_BlockJacobiApply(C.byref(dy), C.byref(dx)), "bj") if i == 0 else None
    check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
    check(L.preAlps_hip_timer_start(), "ts")
    check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
    if i >= 10: tot += sec.value
print("spmm %.1f us  val %#x slot %#x x %#x y %#x" % (1e6 * tot / 30, int(prob.stat("spmm_val_address")), int(prob.stat("spmm_slot_address")),
      C.cast(dx.val, C.c_void_p).value, C.cast(dy.val, C.c_void_p).value), flush=True)
def time_spmm(reps=20):
    tot = 0.0
    for i in range(reps + 5):
        check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 5: tot += sec.value
    return 1e6 * tot / reps
# the matrix arrays on other physical pages of the same process (the old ones stay allocated)
if os.environ.get("SPMM_STATE_MOVES"):
    import ctypes
    mv = L._lib.preAlps_hip_debug_move_plan if hasattr(L, "_lib") else ctypes.CDLL(os.path.join(os.path.dirname(prealps_amd.__file__), "libprealps_hip.so")).preAlps_hip_debug_move_plan
    for k in range(int(os.environ["SPMM_STATE_MOVES"])):
        which = (1, 2, 3)[k % 3]
        assert mv(which) == 0
        print("   move %d (%s): spmm %.1f us  val %#x slot %#x" % (k, {1: "values", 2: "slots", 3: "both"}[which], time_spmm(),
              int(prob.stat("spmm_val_address")), int(prob.stat("spmm_slot_address"))), flush=True)
# the same arrays through other streams (other hardware queues) of the same process
if os.environ.get("SPMM_STATE_STREAMS"):
    import torch
    streams = [torch.cuda.Stream(device=0) for _ in range(int(os.environ["SPMM_STATE_STREAMS"]))]
    for si, st in enumerate(streams):
        check(L.preAlps_hip_set_stream(C.c_void_p(st.cuda_stream)), "set_stream")
        tot = 0.0
        for i in range(30):
            check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
            check(L.preAlps_hip_timer_start(), "ts")
            check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
            check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
            if i >= 10: tot += sec.value
        print("   stream %d (%#x): spmm %.1f us" % (si, st.cuda_stream, 1e6 * tot / 20), flush=True)
    check(L.preAlps_hip_set_stream(C.c_void_p(0)), "set_stream")
prob.close()
