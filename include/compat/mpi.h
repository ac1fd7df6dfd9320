/* compat/mpi.h -- single-process stand-in for the handful of MPI calls the ECG
 * drivers make (examples/test_ecg_prealps_op.c:69,145-146,181,237).  One process
 * drives one GPU; subdomains are PREALPS_NPARTS, not ranks.  Use the system
 * <mpi.h> instead (and -DPREALPS_USE_SYSTEM_MPI) when launching one process per
 * GPU under mpirun. */
#ifndef PREALPS_COMPAT_MPI_H
#define PREALPS_COMPAT_MPI_H
#include <stdlib.h>
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_COMM_WORLD ((MPI_Comm)0x44000000)
#define MPI_DOUBLE ((MPI_Datatype)0x4c00080b)
#define MPI_INT ((MPI_Datatype)0x4c000405)
#define MPI_SUM ((MPI_Op)0x58000003)
#define MPI_MAX ((MPI_Op)0x58000001)
#define MPI_IN_PLACE ((void*)-1)
#define MPI_SUCCESS 0
static inline int MPI_Init(int* argc, char*** argv) { (void)argc; (void)argv; return 0; }
static inline int MPI_Finalize(void) { return 0; }
static inline int MPI_Comm_size(MPI_Comm c, int* size) { (void)c; *size = 1; return 0; }
static inline int MPI_Comm_rank(MPI_Comm c, int* rank) { (void)c; *rank = 0; return 0; }
static inline int MPI_Barrier(MPI_Comm c) { (void)c; return 0; }
static inline int MPI_Abort(MPI_Comm c, int code) { (void)c; exit(code ? code : 1); return 0; }
static inline int MPI_Allreduce(const void* s, void* r, int n, MPI_Datatype d, MPI_Op o, MPI_Comm c) {
  (void)s; (void)r; (void)n; (void)d; (void)o; (void)c; /* one process: in-place sum is the identity */
  return 0;
}
static inline double MPI_Wtime(void);
#include <time.h>
static inline double MPI_Wtime(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
#endif
