/* ecg.h -- drop-in for src/solvers/ecg.h of the reference: everything lives in preAlps_abi.h. */
#ifndef ECG_H
#define ECG_H
#include "preAlps_abi.h"
#endif
