/* operator.h -- drop-in for utils/operator.h of the reference: everything lives in preAlps_abi.h. */
#ifndef OPERATOR_H
#define OPERATOR_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "preAlps_abi.h"
#endif
