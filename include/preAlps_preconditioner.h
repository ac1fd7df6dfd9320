/* preAlps_preconditioner.h -- drop-in for src/preconditioners/preAlps_preconditioner.h (and
 * preAlps_preconditioner_struct.h) of the reference. */
#ifndef PREALPS_PRECONDITIONER_H
#define PREALPS_PRECONDITIONER_H
#include "preAlps_abi.h"
#endif
