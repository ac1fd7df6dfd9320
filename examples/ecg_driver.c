/*
 * ecg_driver.c -- a C driver for libprealps_hip.so with the call sequence of the
 * reference's examples/test_ecg_prealps_op.c:151-239 (operator -> block-Jacobi ->
 * rhs -> RCI loop -> print), written against include/ only.
 *
 *   gcc -std=gnu11 -Iinclude examples/ecg_driver.c -Lprealps_amd -lprealps_hip \
 *       -Wl,-rpath,$PWD/prealps_amd -lm -o ecg_driver
 *   PREALPS_NPARTS=8 ./ecg_driver -m tests/golden/LFAT5.mtx -e 2 -o 0 -r 0
 *
 * One MPI rank per GPU, like the reference (add -DPREALPS_USE_SYSTEM_MPI -I$MPI/include and link
 * libmpi): the library takes rank and size from the communicator, binds device and RCCL itself and
 * distributes the matrix from rank 0 -- the driver needs nothing but MPI_Init / MPI_Finalize:
 *   PREALPS_NPARTS=4096 mpiexec -n 8 ./ecg_driver -m A.mtx -e 4
 * -x prefix writes each rank's part of the solution to prefix.<rank> (raw doubles).
 */
#include <getopt.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "block_jacobi.h"
#include "ecg.h"
#include "operator.h"
#include "preAlps_hip.h"

int main(int argc, char** argv) {
  double tol = 1e-5;
  int maxIter = 1000, enlFac = 1, ortho_alg = 0, bs_red = 0, c, rank = 0;
  const char* file = NULL;
  const char* dump = NULL;
#ifdef PREALPS_USE_SYSTEM_MPI
  MPI_Init(&argc, &argv);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
#endif
  while ((c = getopt(argc, argv, "e:i:m:o:r:t:x:")) != -1) switch (c) {
      case 'e': enlFac = atoi(optarg); break;
      case 'i': maxIter = atoi(optarg); break;
      case 'm': file = optarg; break;
      case 'o': ortho_alg = atoi(optarg); break;
      case 'r': bs_red = atoi(optarg); break;
      case 't': tol = atof(optarg); break;
      case 'x': dump = optarg; break;
      default: fprintf(stderr, "usage: %s -m A.mtx -e t [-o 0|1] [-r 0|1] [-i maxit] [-t tol]\n", argv[0]); return 2;
    }
  if (!file) { fprintf(stderr, "-m matrix.mtx is required\n"); return 2; }

  CPLM_Mat_CSR_t A = CPLM_MatCSRNULL();
  int M, m, sizeRowPos, sizeColPos;
  int *rowPos = NULL, *colPos = NULL;
  preAlps_OperatorBuild(file, MPI_COMM_WORLD);
  preAlps_OperatorGetA(&A);
  preAlps_OperatorGetSizes(&M, &m);
  preAlps_OperatorGetRowPosPtr(&rowPos, &sizeRowPos);
  preAlps_OperatorGetColPosPtr(&colPos, &sizeColPos);
  preAlps_BlockJacobiCreate(&A, rowPos, sizeRowPos, colPos, sizeColPos);

  double* rhs = (double*)malloc(m * sizeof(double));
  double* sol = (double*)malloc(m * sizeof(double));
  preAlps_hip_reference_rhs(rhs); /* what np = nparts ranks of the reference driver build */

  preAlps_ECG_t ecg;
  ecg.comm = MPI_COMM_WORLD;
  ecg.globPbSize = M; ecg.locPbSize = m;
  ecg.maxIter = maxIter; ecg.enlFac = enlFac; ecg.tol = tol;
  ecg.ortho_alg = (ortho_alg == 0 ? ORTHODIR : ORTHOMIN);
  ecg.bs_red = (bs_red == 0 ? NO_BS_RED : ADAPT_BS);
  int rci_request = 0, stop = 0;
  preAlps_ECGInitialize(&ecg, rhs, &rci_request);
  preAlps_BlockJacobiApply(ecg.R, ecg.P);
  preAlps_BlockOperator(ecg.P, ecg.AP);
  while (stop != 1) {
    preAlps_ECGIterate(&ecg, &rci_request);
    if (rci_request == 0) {
      preAlps_BlockOperator(ecg.P, ecg.AP);
    } else if (rci_request == 1) {
      preAlps_ECGStoppingCriterion(&ecg, &stop);
      if (stop == 1) break;
      if (ecg.ortho_alg == ORTHOMIN) preAlps_BlockJacobiApply(ecg.R, ecg.Z);
      else preAlps_BlockJacobiApply(ecg.AP, ecg.Z);
    }
  }
  preAlps_ECGFinalize(&ecg, sol);
  if (rank == 0) preAlps_ECGPrint(&ecg, 0);
  if (dump) {
    char name[1024];
    snprintf(name, sizeof(name), "%s.%d", dump, rank);
    FILE* f = fopen(name, "wb");
    if (f) { fwrite(sol, sizeof(double), (size_t)m, f); fclose(f); }
  }
  free(rhs); free(sol);
  preAlps_BlockJacobiFree();
  preAlps_OperatorFree();
  preAlps_hip_shutdown();
#ifdef PREALPS_USE_SYSTEM_MPI
  MPI_Finalize();
#endif
  return 0;
}
