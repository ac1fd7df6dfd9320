/* preAlps_preconditioner_struct.h -- drop-in for the reference header of the same name. */
#ifndef PREALPS_PRECONDITIONER_STRUCT_H
#define PREALPS_PRECONDITIONER_STRUCT_H
#include "preAlps_abi.h"
#endif
