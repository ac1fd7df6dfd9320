"""ctypes binding of libprealps_hip.so (include/preAlps_abi.h, include/preAlps_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprealps_hip.so")

# enums of src/solvers/ecg.h:23-37
ORTHOMIN, ORTHODIR, ORTHODIR_FUSED = 0, 1, 2
ADAPT_BS, NO_BS_RED = 0, 1
ROW_MAJOR, COL_MAJOR = 0, 1


class PreAlpsError(RuntimeError):
    pass


class CPLM_Info_Dense_t(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("m", C.c_int), ("n", C.c_int), ("lda", C.c_int),
                ("nval", C.c_int), ("stor_type", C.c_int)]


class CPLM_Mat_Dense_t(C.Structure):
    _fields_ = [("val", C.c_void_p), ("info", CPLM_Info_Dense_t)]


class CPLM_Info_t(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("nnz", C.c_int), ("m", C.c_int), ("n", C.c_int),
                ("lnnz", C.c_int), ("blockSize", C.c_int), ("format", C.c_int), ("structure", C.c_int)]


class CPLM_Mat_CSR_t(C.Structure):
    _fields_ = [("info", CPLM_Info_t), ("rowPtr", C.POINTER(C.c_int)), ("colInd", C.POINTER(C.c_int)),
                ("val", C.POINTER(C.c_double))]


_PD = C.POINTER(CPLM_Mat_Dense_t)


class preAlps_ECG_t(C.Structure):
    _fields_ = [("b", C.c_void_p),
                ("X", _PD), ("R", _PD), ("V", _PD), ("AV", _PD), ("Z", _PD), ("alpha", _PD), ("beta", _PD),
                ("P", _PD), ("AP", _PD),
                ("R_p", C.c_void_p), ("P_p", C.c_void_p), ("AP_p", C.c_void_p), ("Z_p", C.c_void_p),
                ("work", C.c_void_p), ("iwork", C.c_void_p),
                ("normb", C.c_double), ("res", C.c_double),
                ("iter", C.c_int), ("bs", C.c_int), ("kbs", C.c_int),
                ("globPbSize", C.c_int), ("locPbSize", C.c_int), ("maxIter", C.c_int), ("enlFac", C.c_int),
                ("tol", C.c_double), ("ortho_alg", C.c_int), ("bs_red", C.c_int), ("comm", C.c_int),
                ("tot_t", C.c_double), ("comm_t", C.c_double), ("trsm_t", C.c_double), ("gemm_t", C.c_double),
                ("potrf_t", C.c_double), ("pstrf_t", C.c_double), ("lapmt_t", C.c_double),
                ("gesvd_t", C.c_double), ("geqrf_t", C.c_double), ("ormqr_t", C.c_double), ("copy_t", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int)

# every symbol include/preAlps_abi.h and include/preAlps_hip.h declare
EXPORTS = [
    "CPLM_MatDenseSetInfo",
    "preAlps_ECGInitialize", "preAlps_ECGIterate", "preAlps_ECGStoppingCriterion", "preAlps_ECGFinalize",
    "preAlps_ECGPrint", "_preAlps_ECGMalloc", "_preAlps_ECGReset", "_preAlps_ECGWrapUp", "_preAlps_ECGFree",
    "_preAlps_ECGSplit", "_preAlps_ECGIterateOmin", "_preAlps_ECGIterateOdir", "_preAlps_ECGIterateOdirFused",
    "preAlps_OperatorBuild", "preAlps_OperatorFree", "preAlps_OperatorPrint", "preAlps_OperatorGetSizes",
    "preAlps_BlockOperator", "preAlps_OperatorGetA", "preAlps_OperatorGetRowPosPtr",
    "preAlps_OperatorGetColPosPtr", "preAlps_OperatorGetDepPtr",
    "preAlps_BlockJacobiCreate", "preAlps_BlockJacobiApply", "preAlps_BlockJacobiFree",
    "preAlps_PreconditionerCreate", "preAlps_PreconditionerDestroy", "preAlps_PreconditionerMatApply",
    "preAlps_hip_init", "preAlps_hip_shutdown", "preAlps_hip_set_stream", "preAlps_hip_get_stream",
    "preAlps_hip_sync", "preAlps_hip_set_abort_mode", "preAlps_hip_last_error", "preAlps_hip_panel_stride",
    "preAlps_hip_set_world", "preAlps_hip_set_comm", "preAlps_hip_rccl_unique_id", "preAlps_hip_rccl_init", "preAlps_hip_comm_selftest", "preAlps_OperatorBuildFromCSR",
    "preAlps_OperatorGetPermPtr", "preAlps_hip_plan_only", "preAlps_hip_prepare_operator", "preAlps_OperatorGetHaloPlan", "preAlps_hip_pack_map", "preAlps_hip_nparts", "preAlps_hip_reference_rhs", "preAlps_ECGSolve", "preAlps_ECGAdvance", "preAlps_hip_hbm_probe",
    "preAlps_hip_panel_alloc", "preAlps_hip_panel_free", "preAlps_hip_panel_to_host",
    "preAlps_hip_panel_from_host", "preAlps_hip_get_stat", "preAlps_hip_timing",
    "preAlps_hip_timer_start", "preAlps_hip_timer_stop",
    "preAlps_hip_timing_reset", "preAlps_hip_get_time",
    "preAlps_hip_partition_kway", "preAlps_hip_rccl_available",
    "preAlps_hip_panel_gram", "preAlps_hip_panel_update", "preAlps_hip_panel_trsm_update",
    "preAlps_hip_panel_permute_solve",
    "preAlps_hip_nd_selfcheck", "preAlps_hip_loopback", "preAlps_hip_graphs",
]

_lib = None


def load():
    """Load libprealps_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PreAlpsError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C prealps_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    pe, pi, pd = C.POINTER(preAlps_ECG_t), C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.preAlps_hip_last_error.restype = C.c_char_p
    L.preAlps_hip_get_stream.restype = C.c_void_p
    L.preAlps_hip_set_stream.argtypes = [C.c_void_p]
    L.preAlps_hip_set_comm.argtypes = [ALLREDUCE_FN, EXCHANGE_FN, C.c_void_p]
    L.preAlps_hip_rccl_unique_id.argtypes = [C.c_char_p]
    L.preAlps_hip_rccl_init.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.preAlps_ECGInitialize.argtypes = [pe, pd, pi]
    L.preAlps_ECGIterate.argtypes = [pe, pi]
    L.preAlps_ECGStoppingCriterion.argtypes = [pe, pi]
    L.preAlps_ECGFinalize.argtypes = [pe, pd]
    L.preAlps_ECGPrint.argtypes = [pe, C.c_int]
    L.preAlps_ECGPrint.restype = None
    L._preAlps_ECGWrapUp.argtypes = [pe, pd]
    L._preAlps_ECGFree.argtypes = [pe]
    L._preAlps_ECGFree.restype = None
    L.preAlps_ECGSolve.argtypes = [pe, pd, pd, pd, pi, C.c_int, pi]
    L.preAlps_ECGAdvance.argtypes = [pe, pd, pi, C.c_int, pi, pi, pd]
    L._preAlps_ECGReset.argtypes = [pe, pd, pi]
    L.preAlps_BlockOperator.argtypes = [_PD, _PD]
    L.preAlps_BlockJacobiApply.argtypes = [_PD, _PD]
    L.preAlps_PreconditionerCreate.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    L.preAlps_PreconditionerDestroy.argtypes = [C.POINTER(C.c_void_p)]
    L.preAlps_PreconditionerMatApply.argtypes = [C.c_void_p, _PD, _PD]
    L.preAlps_BlockJacobiCreate.argtypes = [C.POINTER(CPLM_Mat_CSR_t), pi, C.c_int, pi, C.c_int]
    L.preAlps_OperatorBuild.argtypes = [C.c_char_p, C.c_int]
    L.preAlps_OperatorBuildFromCSR.argtypes = [C.c_int, pi, pi, pd, C.c_int, pi, C.c_int]
    L.preAlps_OperatorGetA.argtypes = [C.POINTER(CPLM_Mat_CSR_t)]
    L.preAlps_OperatorGetSizes.argtypes = [pi, pi]
    L.preAlps_OperatorGetRowPosPtr.argtypes = [C.POINTER(pi), pi]
    L.preAlps_OperatorGetColPosPtr.argtypes = [C.POINTER(pi), pi]
    L.preAlps_OperatorGetDepPtr.argtypes = [C.POINTER(pi), pi]
    L.preAlps_OperatorGetPermPtr.argtypes = [C.POINTER(pi), pi]
    L.preAlps_OperatorGetHaloPlan.argtypes = [pi, C.POINTER(pi), C.POINTER(pi), C.POINTER(pi), C.POINTER(pi), pi, C.POINTER(pi), pi]
    L.preAlps_hip_plan_only.restype = None
    L.preAlps_OperatorFree.restype = None
    L.preAlps_BlockJacobiFree.restype = None
    L.preAlps_hip_reference_rhs.argtypes = [pd]
    L.preAlps_hip_panel_alloc.argtypes = [_PD, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.preAlps_hip_panel_free.argtypes = [_PD]
    L.preAlps_hip_panel_free.restype = None
    L.preAlps_hip_panel_to_host.argtypes = [_PD, C.c_int, pd, C.c_int]
    L.preAlps_hip_panel_from_host.argtypes = [_PD, C.c_int, pd, C.c_int]
    L.preAlps_hip_get_stat.argtypes = [C.c_char_p, pd]
    L.preAlps_hip_get_time.argtypes = [C.c_char_p, pd]
    L.preAlps_hip_timer_stop.argtypes = [pd]
    L.preAlps_hip_hbm_probe.argtypes = [C.c_size_t, C.c_int, pd, pd]
    L.preAlps_hip_panel_gram.argtypes = [_PD, _PD, _PD, pd, C.c_int]
    L.preAlps_hip_panel_update.argtypes = [_PD, _PD, _PD, pd, C.c_int]
    L.preAlps_hip_panel_trsm_update.argtypes = [_PD, _PD, _PD, _PD, pd, pd, pd]
    L.preAlps_hip_panel_permute_solve.argtypes = [_PD, _PD, C.POINTER(C.c_int), C.c_int, pd, C.c_int]
    L.preAlps_hip_nd_selfcheck.argtypes = [C.c_int, pi, pi, pd, C.c_int, pd]
    L.preAlps_hip_partition_kway.argtypes = [C.c_int, pi, pi, C.c_int, pi]
    L.preAlps_hip_timing.restype = None
    L.preAlps_hip_graphs.restype = None
    L.preAlps_hip_timing_reset.restype = None
    L.preAlps_hip_shutdown.restype = None
    L.preAlps_hip_set_abort_mode.restype = None
    # Python callers want exceptions, not abort()
    L.preAlps_hip_set_abort_mode(0)
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = load().preAlps_hip_last_error()
        raise PreAlpsError("%s failed: %s" % (what or "call", msg.decode() if msg else "?"))
