// bj_g4.hip compiled for blocks of up to 224 rows (14 register tiles); see there.
#define G4_NT 14
#include "bj_g4.hip"
