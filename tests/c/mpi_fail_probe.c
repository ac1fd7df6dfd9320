/* Test helper (built by tests/test_mpi_cpu.py with gcc + libmpi): what happens to an MPI run when the set-up
 * fails on rank 0 only (a file that does not exist: only rank 0 reads it).  usage:
 *   mpiexec -n P mpi_fail_probe <file.mtx> <abort_mode 0|1>
 * abort_mode 1 (the library's default): the failing rank prints the reference's abort banner and the whole run
 * ends through MPI_Abort(MPI_COMM_WORLD, 1), as CPLM_Abort does (utils/cplm_core/cplm_utils.c:42-58).
 * abort_mode 0: preAlps_OperatorBuild returns 1 on EVERY rank (nobody is left inside a collective); each rank
 * prints "rank r: rc 1" and the program ends in MPI_Finalize. */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include "preAlps_hip.h"

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  if (argc < 3) MPI_Abort(MPI_COMM_WORLD, 2);
  preAlps_hip_plan_only(1);
  preAlps_hip_set_abort_mode(atoi(argv[2]));
  int rc = preAlps_OperatorBuild(argv[1], MPI_COMM_WORLD);
  printf("rank %d: rc %d\n", rank, rc);
  fflush(stdout);
  MPI_Barrier(MPI_COMM_WORLD);
  MPI_Finalize();
  return 0;
}
