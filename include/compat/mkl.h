/* compat/mkl.h -- the ECG drivers include <mkl.h> only to pin MKL to one thread
 * (examples/test_ecg_prealps_op.c:149).  libprealps_hip.so needs no MKL. */
#ifndef PREALPS_COMPAT_MKL_H
#define PREALPS_COMPAT_MKL_H
static inline void MKL_Set_Num_Threads(int n) { (void)n; }
#endif
