/*
 * preAlps_abi.h -- the data types and entry points of the preAlps ECG hot
 * path that libprealps_hip.so exports.  Field order, field names, enum order
 * and signatures follow the reference so that a caller compiled against the
 * reference headers is binary compatible:
 *
 *   CPLM_storage_type_t, CPLM_Info_Dense_t, CPLM_Mat_Dense_t
 *                               <- utils/cplm_light/cplm_matdense.h:21-39
 *   CPLM_Mat_CSR_format_t, Struct_Type, CPLM_Info_t, CPLM_Mat_CSR_t
 *                               <- utils/cplm_core/cplm_matcsr_struct.h:20-71
 *   CPLM_MatCSRNULL             <- utils/cplm_core/cplm_matcsr_core.h:8-11
 *   timing no-op macros         <- utils/cplm_core/cplm_timing.h:4-49
 *   preAlps_ECG_t + enums       <- src/solvers/ecg.h:23-100
 *   preAlps_ECG* entry points   <- src/solvers/ecg.h:116-247
 *   preAlps_Operator*, preAlps_BlockOperator
 *                               <- utils/operator.h:50-110
 *   preAlps_BlockJacobi*        <- src/preconditioners/block_jacobi.h:45-65
 *
 * What differs from the reference, by design:
 *   - every `double* val` inside a CPLM_Mat_Dense_t owned by the solver, and
 *     `work`, `R_p/P_p/AP_p/Z_p`, are DEVICE pointers (HBM).  Panels are stored
 *     row-interleaved: element (i, j) lives at val[i * ts + j], where
 *     ts = preAlps_hip_panel_stride(enlFac); `info` still describes the logical
 *     m x n shape the reference would have.
 *   - the number of subdomains ("ranks" of the reference: Jacobi blocks =
 *     splitting domains) is `nparts`, independent of the number of processes.
 */
#ifndef PREALPS_ABI_H
#define PREALPS_ABI_H

#include <stddef.h>

#ifdef PREALPS_USE_SYSTEM_MPI
#include <mpi.h>
#else
/* No MPI runtime is needed by this library; the communicator field of
 * preAlps_ECG_t is kept (MPICH ABI: MPI_Comm is an int) and ignored. */
#ifndef MPI_COMM_WORLD
typedef int MPI_Comm;
#define MPI_COMM_WORLD ((MPI_Comm)0x44000000)
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense panels ------------------------------------------------------ */
typedef enum { ROW_MAJOR, COL_MAJOR } CPLM_storage_type_t;

typedef struct {
  int M, N;   /* global rows / cols */
  int m, n;   /* local rows / cols */
  int lda;
  int nval;   /* m*n */
  CPLM_storage_type_t stor_type;
} CPLM_Info_Dense_t;

typedef struct {
  double* val;
  CPLM_Info_Dense_t info;
} CPLM_Mat_Dense_t;

#define CPLM_MatDenseNULL() \
  { .val = NULL, .info = { .M = 0, .N = 0, .m = 0, .n = 0, .lda = 0, .nval = 0, .stor_type = ROW_MAJOR } }

int CPLM_MatDenseSetInfo(CPLM_Mat_Dense_t* A, int M, int N, int m, int n,
                         CPLM_storage_type_t storage);

/* ---- CSR row panels ---------------------------------------------------- */
typedef enum { FORMAT_CSR, FORMAT_BCSR, FORMAT_BCSR_VAR } CPLM_Mat_CSR_format_t;
typedef enum { UNSYMMETRIC, SYMMETRIC } Struct_Type;
typedef enum { AVOID_PERMUTE, PERMUTE } Choice_permutation;

typedef struct {
  int M, N, nnz;     /* global */
  int m, n, lnnz;    /* local  */
  int blockSize;
  CPLM_Mat_CSR_format_t format;
  Struct_Type structure;
} CPLM_Info_t;

typedef struct {
  CPLM_Info_t info;
  int* rowPtr;
  int* colInd;
  double* val;
} CPLM_Mat_CSR_t;

#define CPLM_MatCSRNULL() \
  { .info = { .M = 0, .N = 0, .nnz = 0, .m = 0, .n = 0, .lnnz = 0, .blockSize = 0, \
              .format = FORMAT_CSR, .structure = UNSYMMETRIC },                   \
    .rowPtr = NULL, .colInd = NULL, .val = NULL }

/* ---- instrumentation macros: compiled out, as in the reference ---------- */
#ifndef CPLM_TIMING_H
#define CPLM_TIMING_H
#define CPLM_PUSH
#define CPLM_POP
#define CPLM_BEGIN_TIME
#define CPLM_END_TIME
#define CPLM_OPEN_TIMER
#define CPLM_CLOSE_TIMER
#define CPLM_TIC(a, b)
#define CPLM_TAC(a)
#define CPLM_SetEnv()
#define CPLM_printTimer(a)
#define CPLM_resetTimer()
enum { step1 = 1, step2, step3, step4, step5, step6, step7, step8, step9, step10,
       step11, step12, step13, step14, step15, step16, step17, step18, step19, step20,
       step21, step22, step23, step24, step25, step26, step27, step28, step29, step30 };
#endif

/* ---- ECG solver state -------------------------------------------------- */
typedef enum { ORTHOMIN, ORTHODIR, ORTHODIR_FUSED } preAlps_ECG_Ortho_Alg_t;
typedef enum { ADAPT_BS, NO_BS_RED } preAlps_ECG_Block_Size_Red_t;

typedef struct {
  double* b;
  CPLM_Mat_Dense_t* X;
  CPLM_Mat_Dense_t* R;
  CPLM_Mat_Dense_t* V;
  CPLM_Mat_Dense_t* AV;
  CPLM_Mat_Dense_t* Z;
  CPLM_Mat_Dense_t* alpha;
  CPLM_Mat_Dense_t* beta;
  CPLM_Mat_Dense_t* P;
  CPLM_Mat_Dense_t* AP;
  double* R_p;
  double* P_p;
  double* AP_p;
  double* Z_p;
  double* work;
  int* iwork;
  double normb;
  double res;
  int iter;
  int bs;
  int kbs;
  int globPbSize;
  int locPbSize;
  int maxIter;
  int enlFac;
  double tol;
  preAlps_ECG_Ortho_Alg_t ortho_alg;
  preAlps_ECG_Block_Size_Red_t bs_red;
  MPI_Comm comm;
  double tot_t, comm_t, trsm_t, gemm_t, potrf_t, pstrf_t, lapmt_t, gesvd_t,
         geqrf_t, ormqr_t, copy_t;
} preAlps_ECG_t;

/* Reverse-communication solver (src/solvers/ecg.h:116-148).
 * After Initialize: rci = 0 and the caller computes P = M^-1 R, AP = A P.
 * Iterate flips rci 0 -> 1 (caller: StoppingCriterion, then Z = M^-1 AP for
 * Orthodir / M^-1 R for Orthomin) and 1 -> 0 (caller: AP = A P).  For
 * ORTHODIR_FUSED the caller computes AP and Z before every call and rci == 1
 * means converged.  rhs / solution are HOST arrays of locPbSize doubles. */
int preAlps_ECGInitialize(preAlps_ECG_t* ecg, double* rhs, int* rci_request);
int preAlps_ECGIterate(preAlps_ECG_t* ecg, int* rci_request);
int preAlps_ECGStoppingCriterion(preAlps_ECG_t* ecg, int* stop);
int preAlps_ECGFinalize(preAlps_ECG_t* ecg, double* solution);
void preAlps_ECGPrint(preAlps_ECG_t* ecg, int verbosity);
/* "private" entry points of the reference (ecg.h:152-247) */
int _preAlps_ECGMalloc(preAlps_ECG_t* ecg);
int _preAlps_ECGReset(preAlps_ECG_t* ecg, double* rhs, int* rci_request);
int _preAlps_ECGWrapUp(preAlps_ECG_t* ecg, double* solution);
void _preAlps_ECGFree(preAlps_ECG_t* ecg);
int _preAlps_ECGSplit(double* x, CPLM_Mat_Dense_t* XSplit, int colIndex);
int _preAlps_ECGIterateOmin(preAlps_ECG_t* ecg, int* rci_request);
int _preAlps_ECGIterateOdir(preAlps_ECG_t* ecg, int* rci_request);
int _preAlps_ECGIterateOdirFused(preAlps_ECG_t* ecg, int* rci_request);

/* ---- operator (utils/operator.h:50-110) -------------------------------- */
int preAlps_OperatorBuild(const char* matrixFilename, MPI_Comm comm);
void preAlps_OperatorFree(void);
void preAlps_OperatorPrint(int rank);
int preAlps_OperatorGetSizes(int* M, int* m);
int preAlps_BlockOperator(CPLM_Mat_Dense_t* X, CPLM_Mat_Dense_t* AX);
int preAlps_OperatorGetA(CPLM_Mat_CSR_t* A);
int preAlps_OperatorGetRowPosPtr(int** rowPos, int* sizeRowPos);
int preAlps_OperatorGetColPosPtr(int** colPos, int* sizeColPos);
int preAlps_OperatorGetDepPtr(int** dep, int* sizeDep);

/* ---- block-Jacobi preconditioner (block_jacobi.h:45-65) ---------------- */
int preAlps_BlockJacobiCreate(CPLM_Mat_CSR_t* A, int* rowPos, int sizeRowPos,
                              int* colPos, int sizeColPos);
int preAlps_BlockJacobiApply(CPLM_Mat_Dense_t* A_in, CPLM_Mat_Dense_t* B_out);
void preAlps_BlockJacobiFree(void);

/* ---- generic preconditioner handle of the ECG drivers --------------------
 * (src/preconditioners/preAlps_preconditioner_struct.h:13-33, preAlps_preconditioner.h:18-25,
 * preAlps_preconditioner.c:20-76).  PREALPS_NOPREC copies the panel, PREALPS_BLOCKJACOBI
 * calls preAlps_BlockJacobiApply; LORASC and PRESC are outside this library and abort like
 * the reference's "Unknown preconditioner". */
typedef enum { LEFT_PREC, SPLIT_PREC } Prec_Side_t;
typedef enum { PREALPS_NOPREC, PREALPS_BLOCKJACOBI, PREALPS_LORASC, PREALPS_PRESC } Prec_Type_t;
typedef struct {
  Prec_Side_t side;
  Prec_Type_t type;
  void* data;
} PreAlps_preconditioner_t;
int preAlps_PreconditionerCreate(PreAlps_preconditioner_t** precond, Prec_Type_t precond_type, void* data);
int preAlps_PreconditionerDestroy(PreAlps_preconditioner_t** precond);
int preAlps_PreconditionerMatApply(PreAlps_preconditioner_t* precond, CPLM_Mat_Dense_t* A_in,
                                   CPLM_Mat_Dense_t* B_out);

#ifdef __cplusplus
}
#endif
#endif
