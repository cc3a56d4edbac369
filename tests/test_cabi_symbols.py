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


def test_host_routine_tridiagonal_extremes_against_lapack(lib):
    """`nss_tridiag_extremes` is host code (the convergence check of the device-resident Lanczos, hipla/eigen.py): the
    smallest / largest eigenvalue of random tridiagonals, of the tridiagonals of a Lanczos run with converged copies
    of its extreme Ritz values (clusters: Laguerre's iteration turns linear there), with and without starting hints,
    with wrong hints (inside the spectrum: rejected by the Sturm count), sizes 1 ... 5000 (rescaling of the recurrences)."""
    import numpy as np
    from scipy.linalg import eigvalsh_tridiagonal

    def extremes(d, e, hint_lo=float("nan"), hint_hi=float("nan")):
        d, e = np.ascontiguousarray(d, dtype=np.float64), np.ascontiguousarray(e, dtype=np.float64)
        lo, hi = ctypes.c_double(hint_lo), ctypes.c_double(hint_hi)
        rc = lib.nss_tridiag_extremes(d.ctypes.data, e.ctypes.data if len(e) else None, len(d), ctypes.byref(lo), ctypes.byref(hi))
        assert rc == 0, lib.nss_last_error()
        return lo.value, hi.value

    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 5, 17, 100, 290, 1000, 5000):
        for trial in range(4):
            d = rng.random(n) * 2 - (trial % 2)
            e = rng.standard_normal(max(n - 1, 0)) * 10.0 ** rng.integers(-8, 1)
            ref = eigvalsh_tridiagonal(d, e) if n > 1 else d
            lo, hi = extremes(d, e)
            scale = max(abs(ref).max(), 1e-300)
            assert abs(lo - ref[0]) <= 1e-13 * scale and abs(hi - ref[-1]) <= 1e-13 * scale, (n, trial)
            mid = 0.5 * (ref[0] + ref[-1])                    # hints inside the spectrum change nothing
            assert extremes(d, e, mid, mid) == (lo, hi) or n == 1
    lam = np.linspace(0.003, 1.9, 2000)
    v = rng.standard_normal(len(lam))
    v /= np.linalg.norm(v)
    v_old, beta, ds, es = np.zeros_like(v), 0.0, [], []
    for _ in range(400):                                       # plain Lanczos, no re-orthogonalisation
        w = lam * v - beta * v_old
        a = w @ v
        w -= a * v
        beta = np.linalg.norm(w)
        ds.append(a)
        es.append(beta)
        v_old, v = v, w / beta
    ds, es = np.array(ds), np.array(es)
    prev = None
    for j in range(5, 400, 5):
        ref = eigvalsh_tridiagonal(ds[:j], es[:j - 1])
        lo, hi = extremes(ds[:j], es[:j - 1])
        assert abs(lo - ref[0]) <= 1e-13 * ref[-1] and abs(hi - ref[-1]) <= 1e-13 * ref[-1], j
        if prev is not None:                                   # from just outside the previous check's values
            lo2, hi2 = extremes(ds[:j], es[:j - 1], prev[0] - 2 * (prev[0] - lo) - 1e-14, prev[1] + 2 * (hi - prev[1]) + 1e-14)
            assert abs(lo2 - lo) <= 1e-13 * ref[-1] and abs(hi2 - hi) <= 1e-13 * ref[-1], j
        prev = (lo, hi)
    assert lib.nss_tridiag_extremes(None, None, 3, None, None) != 0 and b"tridiag_extremes" in lib.nss_last_error()
