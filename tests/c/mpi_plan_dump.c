/* Test helper (built by tests/test_mpi_cpu.py with gcc + libmpi): what one MPI rank holds after
 * preAlps_OperatorBuild(file, MPI_COMM_WORLD) in plan-only mode -- its row panel, the halo plan and
 * the ordering -- written to <prefix>.<rank> as int32 / float64 records for the test to compare with
 * the replicated build.  usage: mpiexec -n P mpi_plan_dump matrix.mtx prefix */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include "preAlps_hip.h"

static void put_i(FILE* f, const int* a, int n) { fwrite(&n, sizeof(int), 1, f); if (n) fwrite(a, sizeof(int), (size_t)n, f); }
static void put_d(FILE* f, const double* a, int n) { fwrite(&n, sizeof(int), 1, f); if (n) fwrite(a, sizeof(double), (size_t)n, f); }

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (argc < 3) { fprintf(stderr, "usage: %s matrix.mtx prefix\n", argv[0]); MPI_Abort(MPI_COMM_WORLD, 2); }
  preAlps_hip_plan_only(1);
  preAlps_OperatorBuild(argv[1], MPI_COMM_WORLD);
  CPLM_Mat_CSR_t A = CPLM_MatCSRNULL();
  int M, m, nrp, npeers, nsend, nhalo, nperm;
  int *rowPos, *peers, *srows, *rrows, *sidx, *hcols, *perm;
  preAlps_OperatorGetA(&A);
  preAlps_OperatorGetSizes(&M, &m);
  preAlps_OperatorGetRowPosPtr(&rowPos, &nrp);
  preAlps_OperatorGetHaloPlan(&npeers, &peers, &srows, &rrows, &sidx, &nsend, &hcols, &nhalo);
  preAlps_OperatorGetPermPtr(&perm, &nperm);
  char name[1024];
  snprintf(name, sizeof(name), "%s.%d", argv[2], rank);
  FILE* f = fopen(name, "wb");
  if (!f) MPI_Abort(MPI_COMM_WORLD, 3);
  int head[6] = {rank, size, M, m, A.info.lnnz, A.info.nnz};
  put_i(f, head, 6);
  put_i(f, A.rowPtr, m + 1); put_i(f, A.colInd, A.info.lnnz); put_d(f, A.val, A.info.lnnz);
  put_i(f, rowPos, nrp); put_i(f, peers, npeers); put_i(f, srows, npeers); put_i(f, rrows, npeers);
  put_i(f, sidx, nsend); put_i(f, hcols, nhalo); put_i(f, perm, nperm);
  double* rhs = (double*)malloc((size_t)(m ? m : 1) * sizeof(double));
  preAlps_hip_reference_rhs(rhs);
  put_d(f, rhs, m);
  free(rhs);
  fclose(f);
  preAlps_OperatorFree();
  MPI_Finalize();
  return 0;
}
