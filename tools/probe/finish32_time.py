#!/usr/bin/env python3
"""Time k_finish32 by itself (partial blocks at rest in HBM) for several block counts."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import prealps_amd
L = prealps_amd.load()
L.pa_rt_malloc.restype = C.c_void_p; L.pa_rt_malloc.argtypes = [C.c_size_t]
L.pa_rt_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
L.pa_k_finish32.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
assert L.preAlps_hip_init(0) == 0
n = 8192
buf = L.pa_rt_malloc((n + 64) * 32 * 8 + 4096)
L.pa_rt_memset(buf, 0, (n + 64) * 32 * 8 + 4096)
out = buf + (n + 64) * 32 * 8
sec = C.c_double()
for nblk in (64, 512, 2048, 4174, 8192):
    for t in (0, 4):
        for _ in range(5): L.pa_k_finish32(buf, nblk, buf + n * 32 * 8, t, t, out, out + 512, out + 1024, out + 2048)
        L.preAlps_hip_timer_start()
        for _ in range(100): L.pa_k_finish32(buf, nblk, buf + n * 32 * 8, t, t, out, out + 512, out + 1024, out + 2048)
        L.preAlps_hip_timer_stop(C.byref(sec))
        print("nblk %5d t %d: %.2f us per launch" % (nblk, t, 1e4 * sec.value), flush=True)
