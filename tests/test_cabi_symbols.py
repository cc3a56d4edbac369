"""The C-ABI shared library loads on a CPU-only box and exports every symbol that
include/nss_krylov.h declares (no compute calls here)."""

import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    from hipla.hip_engine import LIB_PATH, load_library
    if not os.path.exists(LIB_PATH):
        entry.build()
    return load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nss_krylov.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"NSS_API\s+[\w\s\*]+?\b(nss_\w+)\s*\(", text)


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for must in ("nss_csr_spmv_f64", "nss_bjac_apply_f64", "nss_dot_f64", "nss_lincomb_f64",
                 "nss_diag_apply_f64", "nss_stream_triad_f64", "nss_last_error"):
        assert must in names
    assert len(names) == len(set(names))


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.nss_abi_version() == 1
    assert isinstance(lib.nss_last_error(), bytes)


def test_ctypes_signatures_cover_the_header(lib):
    from hipla.hip_engine import _signatures
    assert set(_signatures()) == set(declared_symbols())


def test_argument_errors_are_reported_not_thrown(lib):
    out = ctypes.c_void_p()
    rowptr = (ctypes.c_int32 * 2)(0, 5)        # claims 5 entries but nnz = 1
    col = (ctypes.c_int32 * 1)(0)
    val = (ctypes.c_double * 1)(1.0)
    rc = lib.nss_csr_create(1, 1, 1, ctypes.addressof(rowptr), ctypes.addressof(col), ctypes.addressof(val),
                            ctypes.byref(out))
    assert rc != 0 and b"rowptr" in lib.nss_last_error()
    # NULL handles / pointers of the set-up entry points are rejected before anything is launched
    width = ctypes.c_int32()
    assert lib.nss_csr_index_width(None, ctypes.byref(width)) != 0 and b"NULL" in lib.nss_last_error()
    assert lib.nss_csr_spgemm(None, None, 0, ctypes.byref(out), None) != 0 and b"NULL" in lib.nss_last_error()
    nagg = ctypes.c_int64()
    assert lib.nss_amg_aggregate(None, 0.0, None, None, ctypes.byref(nagg), None) != 0
    assert lib.nss_amg_prolongator(None, None, 1, 0.5, ctypes.byref(out), None) != 0
    assert lib.nss_csr_download(None, None, None, None) != 0
    assert lib.nss_csr_transpose(None, ctypes.byref(out)) != 0


def test_product_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import hipla
    prev = hipla.set_engine(None)
    try:
        with pytest.raises(hipla.EngineUnavailable):
            hipla.get_engine()
        with pytest.raises(hipla.EngineUnavailable):
            hipla.Vector(4)
    finally:
        hipla.set_engine(prev)
