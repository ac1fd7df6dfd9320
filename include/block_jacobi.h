/* block_jacobi.h -- drop-in for src/preconditioners/block_jacobi.h of the reference. */
#ifndef BLOCK_JACOBI_H
#define BLOCK_JACOBI_H
#include "preAlps_abi.h"
#endif
