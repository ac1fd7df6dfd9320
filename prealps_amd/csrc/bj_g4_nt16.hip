// bj_g4.hip compiled for blocks of up to 256 rows (16 register tiles); see there.
#define G4_NT 16
#include "bj_g4.hip"
