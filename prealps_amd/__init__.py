"""prealps_amd -- MI355X-native Enlarged Conjugate Gradient hot path behind the
preAlps reverse-communication interface.

The compute lives in ``libprealps_hip.so`` (hand-written HIP for gfx950 + C
host code, built by ``__graft_entry__.build()`` or ``make -C prealps_amd/csrc``).
This package is the Python binding of that C ABI: ctypes mirrors of the
reference's structs (``preAlps_ECG_t``, ``CPLM_Mat_Dense_t``, ``CPLM_Mat_CSR_t``)
and thin wrappers with the reference's function names.  There is no CPU
fallback: importing works anywhere, computing needs an MI355X.
"""
from .lib import (ADAPT_BS, NO_BS_RED, ORTHODIR, ORTHODIR_FUSED, ORTHOMIN, CPLM_Mat_CSR_t,
                  CPLM_Mat_Dense_t, PreAlpsError, load, preAlps_ECG_t)
from .solver import EcgProblem, EcgResult

__all__ = ["load", "PreAlpsError", "preAlps_ECG_t", "CPLM_Mat_Dense_t", "CPLM_Mat_CSR_t",
           "ORTHOMIN", "ORTHODIR", "ORTHODIR_FUSED", "ADAPT_BS", "NO_BS_RED", "EcgProblem",
           "EcgResult"]
