"""CPU-side checks of the product library: it loads without a GPU, exports the
whole C ABI, and fails loudly (no CPU fallback) when asked to compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import prealps_amd
from prealps_amd import lib as pl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b((?:_?preAlps|CPLM)_\w+)\s*\(", txt)
    return {n for n in names if not n.endswith("_fn") and n not in ("CPLM_MatDenseNULL", "CPLM_MatCSRNULL",
                                                                   "CPLM_TIC", "CPLM_TAC", "CPLM_SetEnv",
                                                                   "CPLM_printTimer", "CPLM_resetTimer")}


def test_library_exports_every_declared_symbol():
    L = prealps_amd.load()
    declared = _declared("preAlps_abi.h") | _declared("preAlps_hip.h")
    assert declared, "header parsing found nothing"
    missing = sorted(s for s in declared if not hasattr(L, s))
    assert not missing, missing
    assert declared <= set(pl.EXPORTS) | {"CPLM_MatDenseSetInfo"}


def test_struct_layout_matches_the_reference_abi():
    # src/solvers/ecg.h:45-100 on LP64: 16 pointers, 2 doubles, 7 ints, double, 3 ints(+pad), 11 doubles
    assert C.sizeof(pl.CPLM_Mat_Dense_t) == 40
    assert C.sizeof(pl.CPLM_Mat_CSR_t) == 64
    assert pl.preAlps_ECG_t.normb.offset == 128
    assert pl.preAlps_ECG_t.tol.offset == 176
    assert pl.preAlps_ECG_t.tot_t.offset == 200
    assert C.sizeof(pl.preAlps_ECG_t) == 288


def test_set_info_semantics():
    L = prealps_amd.load()
    d = pl.CPLM_Mat_Dense_t()
    L.CPLM_MatDenseSetInfo(C.byref(d), 10, 4, 5, 4, pl.COL_MAJOR)
    assert (d.info.lda, d.info.nval, d.info.stor_type) == (5, 20, pl.COL_MAJOR)
    L.CPLM_MatDenseSetInfo(C.byref(d), 10, 4, 5, 4, pl.ROW_MAJOR)
    assert d.info.lda == 4
    assert L.preAlps_hip_panel_stride(1) == 2 and L.preAlps_hip_panel_stride(4) == 4
    assert L.preAlps_hip_panel_stride(12) == 16


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = prealps_amd.load()
    assert L.preAlps_hip_init(0) != 0
    assert b"no CPU path" in L.preAlps_hip_last_error() or b"HIP" in L.preAlps_hip_last_error()
    rp = np.array([0, 1, 2], dtype=np.int32)
    ci = np.array([0, 1], dtype=np.int32)
    v = np.array([1.0, 1.0])
    with pytest.raises(prealps_amd.PreAlpsError):
        prealps_amd.EcgProblem(rp, ci, v, 2)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "prealps_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)


REF_DRIVER = "/root/reference/examples/test_ecg_prealps_op.c"


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="the reference tree is not on this machine")
def test_reference_driver_links_unchanged(tmp_path):
    """The drop-in claim: the reference's own driver compiles against include/ and links
    against libprealps_hip.so without a single edit (it is read where it lies, never copied)."""
    import subprocess
    prealps_amd.load()
    exe = str(tmp_path / "ref_driver")
    subprocess.check_call(["gcc", "-std=gnu99", "-w", "-I" + os.path.join(ROOT, "include", "compat"),
                           "-I" + os.path.join(ROOT, "include"), REF_DRIVER,
                           "-L" + os.path.join(ROOT, "prealps_amd"), "-lprealps_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "prealps_amd"), "-lm", "-o", exe])
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "-m", os.path.join(ROOT, "tests", "golden", "LFAT5.mtx"), "-e", "2"],
                           capture_output=True, text=True, env=dict(os.environ, PREALPS_NPARTS="2"))
        assert r.returncode != 0 and "ABORTING from" in r.stderr      # CPLM_Abort-style banner, no fallback


def test_own_c_driver_compiles(tmp_path):
    import subprocess
    prealps_amd.load()
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ecg_driver.c"), "-L" + os.path.join(ROOT, "prealps_amd"),
                           "-lprealps_hip", "-Wl,-rpath," + os.path.join(ROOT, "prealps_amd"), "-lm",
                           "-o", str(tmp_path / "ecg_driver")])
