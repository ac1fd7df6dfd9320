#!/usr/bin/env python3
"""One process: the SpMM of the headline matrix under several plan geometries (rows per block, bytes of
staging area), each operator built, timed behind a block solve, and released in turn -- in-process
rebuilds repeat to 0.3 % (tools/placement_probe.py), so the differences are the geometry's.
usage: spmm_plan_sweep.py [t]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check
t = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rp, ci, v = gen.elasticity3d_csr(70); part, P = gen.box_partition_nodes(70, (2, 4, 8))
X = np.random.default_rng(0).standard_normal((len(rp) - 1, t))
def run(rows, stage):
    os.environ.pop("PREALPS_SPMM_BLOCK_ROWS", None); os.environ.pop("PREALPS_SPMM_STAGE_BYTES", None)
    if rows: os.environ["PREALPS_SPMM_BLOCK_ROWS"] = str(rows)
    if stage: os.environ["PREALPS_SPMM_STAGE_BYTES"] = str(stage)
    prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
    L = prob.L
    prob.create_block_jacobi()
    check(L.preAlps_hip_prepare_operator(t), "prep")
    dx, dy = prob.panel(t, t), prob.panel(t, t)
    prob.to_device(dx, X, t)
    sec = C.c_double(); tot = 0.0
    for i in range(25):
        check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
        check(L.preAlps_hip_timer_start(), "ts")
        check(L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "op")
        check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
        if i >= 5: tot += sec.value
    print("rows %4s stage %6s: spmm %.1f us, %d blocks, runs %d, stream %.0f MB" % (rows or "dflt", stage or "dflt", 1e6 * tot / 20,
          prob.stat("spmm_blocks"), prob.stat("spmm_runs"), prob.stat("spmm_stream_bytes") / 1e6), flush=True)
    prob.close()
run(0, 0)
grid = [(192, 0), (256, 0), (320, 0), (256, 24576), (192, 24576), (256, 40960), (320, 40960), (384, 49152), (128, 16384)] if t <= 4 else \
       [(128, 0), (192, 0), (256, 0), (192, 40960), (128, 32768), (256, 65536), (192, 53248)]
for rows, stage in grid: run(rows, stage)
run(0, 0)
