#!/usr/bin/env python3
"""64 blocks of 17.5 k rows (SURVEY 8(d)'s regime), elasticity 70^3: factor bytes and block-solve time against the
leaf size of the nested dissection (PREALPS_ND_LEAF) and the widest supernode (PREALPS_ND_WIDTH).
usage: r4_nd_leaf_sweep.py leaf[:width] ..."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prealps_amd
from prealps_amd import gen
from prealps_amd.lib import check
t = int(os.environ.get("R4_AB_T", "4"))
rp, ci, v = gen.elasticity3d_csr(70)
part, P = gen.box_partition_nodes(70, (18, 18, 18))
for arg in sys.argv[1:]:
    leaf, _, width = arg.partition(":")
    os.environ["PREALPS_ND_LEAF"] = leaf
    if width: os.environ["PREALPS_ND_WIDTH"] = width
    else: os.environ.pop("PREALPS_ND_WIDTH", None)
    prob = prealps_amd.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
    L = prob.L
    t0 = time.time(); prob.create_block_jacobi(); tb = time.time() - t0
    X = np.random.default_rng(0).standard_normal((prob.m, t))
    dx, dy = prob.panel(t, t), prob.panel(t, t)
    prob.to_device(dx, X, t)
    sec = C.c_double()
    for _ in range(3): check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
    check(L.preAlps_hip_timer_start(), "ts")
    for _ in range(10): check(L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "bj")
    check(L.preAlps_hip_timer_stop(C.byref(sec)), "te")
    fb = prob.stat("bj_factor_bytes")
    print("leaf %s width %s: %d blocks, factor %.2f GB, apply %.3f ms = %.2f TB/s, set-up %.2f s" % (
        leaf, width or "default", P, fb / 1e9, 1e3 * sec.value / 10, fb / (sec.value / 10) / 1e12, tb), flush=True)
    prob.close()
