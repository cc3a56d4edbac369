"""Single-kernel parity on the GPU, through the C ABI (ctypes): every HIP kernel against
numpy/scipy on the same seeded inputs.  Tolerance (SURVEY.md section 8c iv): <= 1e-13
relative for SpMV / dot / block-Jacobi / element-wise kernels (fp64; the only
differences are FMA contraction and the reduction tree)."""

import numpy as np
import pytest
import scipy.sparse as sp

from staggered_grid import diffusion_2d, mac_stokes

pytestmark = pytest.mark.gpu

RTOL = 1e-13


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_engine_is_the_hip_library(hip_engine):
    info = hip_engine.device_info()
    assert hip_engine.name == "hip-gfx950"
    assert info["arch"].startswith("gfx950"), info
    assert info["wavefront"] == 64 and info["cu_count"] >= 200


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097, 1 << 20, (1 << 20) + 3])
def test_blas1(hip_engine, n):
    import hipla
    rng = np.random.default_rng(n)
    x, y, z = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    X, Y, Z = (hipla.Vector.from_numpy(v) for v in (x, y, z))
    W = hipla.Vector(n)
    if n == 0:
        assert hipla.InnerProduct(X, Y) == 0.0
        return
    W[:] = 3.5
    np.testing.assert_array_equal(W.numpy(), np.full(n, 3.5))
    W.data = X
    np.testing.assert_array_equal(W.numpy(), x)
    W *= -0.25
    np.testing.assert_allclose(W.numpy(), -0.25 * x, rtol=1e-15)
    W.data = 2.0 * X - 0.5 * Y + 3.0 * Z
    assert relerr(W.numpy(), 2.0 * x - 0.5 * y + 3.0 * z) < RTOL
    W.data = X + Y + Z - W                        # 4 terms, destination aliased as last operand
    assert relerr(W.numpy(), x + y + z - (2.0 * x - 0.5 * y + 3.0 * z)) < RTOL
    X.data += 0.75 * Y
    assert relerr(X.numpy(), x + 0.75 * y) < RTOL
    d = hipla.InnerProduct(Y, Z)
    assert abs(d - np.dot(y, z)) <= RTOL * np.linalg.norm(y) * np.linalg.norm(z)
    assert abs(hipla.Norm(Y) - np.linalg.norm(y)) <= RTOL * np.linalg.norm(y)


def test_unaligned_views(hip_engine):
    import hipla
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(1001), rng.standard_normal(1001)
    X, Y = hipla.Vector.from_numpy(x), hipla.Vector.from_numpy(y)
    xs, ys = X[1:1000], Y[1:1000]                 # 8-byte aligned only
    assert abs(hipla.InnerProduct(xs, ys) - np.dot(x[1:1000], y[1:1000])) < 1e-11
    xs.data += 2.0 * ys
    x[1:1000] += 2.0 * y[1:1000]
    np.testing.assert_allclose(X.numpy(), x, rtol=1e-15)
    X[3:7] = 9.0
    x[3:7] = 9.0
    np.testing.assert_array_equal(X.numpy(), x)


def test_dot_is_deterministic_and_block(hip_engine):
    import hipla
    rng = np.random.default_rng(2)
    a = [rng.standard_normal(m) for m in (7000, 3000)]
    b = [rng.standard_normal(m) for m in (7000, 3000)]
    A = hipla.BlockVector([hipla.Vector.from_numpy(v) for v in a])
    B = hipla.BlockVector([hipla.Vector.from_numpy(v) for v in b])
    vals = {hipla.InnerProduct(A, B) for _ in range(5)}
    assert len(vals) == 1                          # fixed grid + fixed tree: bit-reproducible
    assert abs(vals.pop() - (np.dot(a[0], b[0]) + np.dot(a[1], b[1]))) < 1e-10


def _spmv_check(eng, mat, seed=0, alpha=1.0, beta=0.0):
    import hipla
    mat = sp.csr_matrix(mat)
    mat.sort_indices()
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(mat.shape[1])
    y0 = rng.standard_normal(mat.shape[0])
    M = hipla.SparseMatrix.from_scipy(mat)
    X, Y = hipla.Vector.from_numpy(x), hipla.Vector.from_numpy(y0)
    eng.csr_spmv(M.handle, alpha, X.buf, beta, Y.buf)
    ref = alpha * (mat @ x) + (beta * y0 if beta != 0.0 else 0.0)
    scale = np.abs(mat) @ np.abs(x) + np.abs(y0) * abs(beta) + 1e-300
    assert np.max(np.abs(Y.numpy() - ref) / scale) < RTOL
    return M


@pytest.mark.parametrize("dim,n", [(2, 12), (2, 61), (3, 6), (3, 17)])
def test_spmv_stokes_blocks(hip_engine, dim, n):
    s = mac_stokes(dim, n)
    for mat in (s.A, s.B, s.B.T.tocsr()):
        M = _spmv_check(hip_engine, mat)
        assert M.handle.info()["lanes_per_row"] == 1
        _spmv_check(hip_engine, mat, seed=1, alpha=-0.5, beta=2.0)
    # transpose path of the protocol: B.T is an explicit CSR built once
    import hipla
    B = hipla.SparseMatrix.from_scipy(s.B)
    assert B.T is B.CreateTranspose() and B.T.T is B
    p = np.random.default_rng(3).standard_normal(s.n_p)
    out = hipla.Vector(s.n_u)
    out.data = B.T * hipla.Vector.from_numpy(p)
    assert relerr(out.numpy(), s.B.T @ p) < RTOL


def hipla_info(eng, mat):
    return _spmv_check(eng, mat, seed=21).handle.info()


def test_spmv_operand_forms(hip_engine):
    """How the kernel reaches x (csrc/csr_stream.h): grid operators and their block-inflated forms take the staged
    form (runs of consecutive columns copied to LDS ahead of the matrix stream); a staged matrix may hold a few row
    blocks that do not fit and gather; a matrix without column runs keeps a gather form.  Every form against scipy."""
    import scipy.sparse as sp
    s = mac_stokes(3, 12)
    for mat in (s.A, s.B, s.B.T.tocsr(), s.inflate(5).A, s.inflate(12).A, s.inflate(12).B):
        M = _spmv_check(hip_engine, mat, seed=11)
        info = M.handle.info()
        assert info["operand_form"] == "staged" and info["index_bytes"] == 2, info
        _spmv_check(hip_engine, mat, seed=12, alpha=-2.0, beta=0.5)
    # a banded operator with ONE dense coupling row (a mean-value constraint): its row block scatters over ~1500
    # runs, and a staged matrix has no per-block fallback -> the whole matrix gathers
    n = 40000
    band = sp.diags([1.0, -2.0, 1.0, 0.3, 0.3], [-1, 0, 1, -200, 200], shape=(n, n), format="lil")
    assert hipla_info(hip_engine, band.tocsr())["operand_form"] == "staged"
    rng = np.random.default_rng(4)
    band[n // 2, rng.choice(n, 1500, replace=False)] = 1.0
    assert hipla_info(hip_engine, band.tocsr())["operand_form"] in ("gather16", "gather32")
    # columns without runs: every row block would need thousands of segments -> not staged
    rnd = sp.random(20000, 20000, density=4e-4, random_state=5, format="csr")
    M = _spmv_check(hip_engine, rnd, seed=14)
    assert M.handle.info()["operand_form"] in ("gather16", "gather32")
    # 13 runs per row block is the capacity of the run table, 14 is one too many
    n = 60000
    # (pairs of adjacent diagonals: the rows of a block share most of a run, so the copy is half the entries)
    def banded(runs):
        offs = [o for j in range(runs) for o in ((j - 6) * 3000, (j - 6) * 3000 + 1)]
        return sp.diags([1.0 + 0.01 * j for j in range(len(offs))], offs, shape=(n, n), format="csr")
    assert hipla_info(hip_engine, banded(13))["operand_form"] == "staged"
    assert hipla_info(hip_engine, banded(14))["operand_form"] in ("gather16", "gather32")
    # one row longer than a chunk inside a staged matrix: its row block is reduced from the 4-byte columns
    long_row = sp.diags([1.0, -2.0, 1.0], [-1, 0, 1], shape=(20000, 20000), format="lil")
    long_row[777, 5000:11000] = 0.5
    info = hipla_info(hip_engine, long_row.tocsr())
    assert info["operand_form"] == "staged", info
    _spmv_check(hip_engine, long_row.tocsr(), seed=16, alpha=0.5, beta=-1.0)
    # odd and even run lengths / starts, operand vectors that are only 8-byte aligned (sub-views of a buffer)
    import hipla
    import torch
    odd = sp.diags([1.0, 2.0, 3.0, 4.0], [-301, 0, 2, 77], shape=(5001, 5001), format="csr")
    M = hipla.SparseMatrix.from_scipy(odd)
    assert M.handle.info()["operand_form"] == "staged"
    buf = torch.arange(5001 + 3, dtype=torch.float64, device="cuda") * 0.25 + 1.0
    out = torch.zeros(5001 + 3, dtype=torch.float64, device="cuda")
    for shift in (0, 1, 2, 3):
        x, y = buf[shift:shift + 5001], out[shift:shift + 5001]
        hip_engine.csr_spmv(M.handle, 1.0, x, 0.0, y)
        ref = odd @ x.cpu().numpy()
        assert np.max(np.abs(y.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref)), shift


def test_spmv_row_per_lane_kernel(hip_engine):
    """Matrices with at most two entries per row (B^T of the staggered grid) above a size threshold are multiplied
    by the row-per-lane kernel from a fixed-width copy: same bits as the stream kernel, empty and one-entry rows
    included; the threshold is a run-time knob so that small cases reach it."""
    import hipla
    lib = hip_engine.lib
    s = mac_stokes(3, 14)
    bt = s.B.T.tocsr()
    rng = np.random.default_rng(8)
    ragged = sp.random(3000, 900, density=1.2e-3, random_state=6, format="csr")
    keep = ragged.getnnz(axis=1) <= 2
    ragged = sp.csr_matrix(sp.diags(keep.astype(float)) @ ragged)               # rows with 0, 1 or 2 entries
    ragged.eliminate_zeros()
    assert set(np.unique(ragged.getnnz(axis=1))) >= {0, 1, 2}
    for mat in (bt, ragged):
        x = rng.standard_normal(mat.shape[1])
        y0 = rng.standard_normal(mat.shape[0])
        out = {}
        for name, rows in (("stream", -1), ("rows", 0)):
            assert lib.nss_csr_direct_rows_threshold(rows) == 0
            try:
                M = hipla.SparseMatrix.from_scipy(mat)
            finally:
                lib.nss_csr_direct_rows_threshold(-1)
            assert (M.handle.info()["operand_form"] == "rows") == (name == "rows"), M.handle.info()
            X, Y = hipla.Vector.from_numpy(x), hipla.Vector.from_numpy(y0)
            hip_engine.csr_spmv(M.handle, -1.5, X.buf, 0.25, Y.buf)
            out[name] = Y.numpy()
        ref = -1.5 * (mat @ x) + 0.25 * y0
        assert np.max(np.abs(out["rows"] - ref)) <= 1e-13 * (np.max(np.abs(ref)) + 1.0)
        np.testing.assert_array_equal(out["rows"], out["stream"])


def _plan_lanes(mean):
    """The plan rule of csrc/spmv.hip: 2048 products per row block (8 per lane); lanes per row = the largest power
    of two for which one reduce pass still covers a full chunk."""
    lanes = 1
    while lanes < 64 and 2 * lanes * 8 <= mean:
        lanes *= 2
    return lanes


def test_spmv_row_length_regimes(hip_engine):
    """Every lanes-per-row instantiation of the CSR-stream kernel (1 ... 64), picked from the mean row length by the
    plan rule."""
    s = mac_stokes(3, 6)
    seen = set()
    for bs in (1, 3, 6, 12, 22, 44, 90):            # ~6 ... ~530 non-zeros per row: 1, 2, 4 ... 64 lanes per row
        infl = s.inflate(bs) if bs > 1 else s
        M = _spmv_check(hip_engine, infl.A)
        mean = infl.A.nnz / infl.A.shape[0]
        info = M.handle.info()
        assert info["lanes_per_row"] == _plan_lanes(mean)
        if info["index_bytes"] == 2:                 # block-structured operator: one 16-bit index per run of columns
            expect = {1: 1, 3: 3, 6: 6, 12: 12, 22: 11, 44: 11, 90: 15}[bs]
            assert info["index_group"] == expect, (bs, info)
            per_block = 128 if info["operand_form"] == "staged" else 64    # segment descriptor / window bases
            assert info["algorithmic_bytes"] == (8 * infl.A.nnz + 2 * (infl.A.nnz // expect)
                                                 + per_block * info["row_blocks"]
                                                 + 4 * (infl.A.shape[0] + 1) + 16 * infl.A.shape[0])
        seen.add(_plan_lanes(mean))
        if bs <= 12:
            _spmv_check(hip_engine, infl.B, seed=5, alpha=2.0, beta=-1.0)
    rng = np.random.default_rng(3)
    for m, n in ((24, 1100), (24, 3500)):      # dense rows: 64 lanes per row
        mat = sp.csr_matrix(rng.standard_normal((m, n)))
        M = _spmv_check(hip_engine, mat, seed=m + n)
        assert M.handle.info()["lanes_per_row"] == _plan_lanes(float(n))
        seen.add(_plan_lanes(float(n)))
    # long-row matrix with rows beyond the 2048-product chunk (whole-workgroup row reduction)
    lens = np.full(400, 60)
    lens[7], lens[399] = 4500, 9000
    rows = np.repeat(np.arange(400), lens)
    cols = np.concatenate([rng.choice(12000, size=k, replace=False) for k in lens])
    mat = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(400, 12000))
    _spmv_check(hip_engine, mat, seed=8, alpha=-0.5, beta=2.0)
    assert seen == {1, 2, 4, 8, 16, 32, 64}


def test_spmv_ragged_empty_and_long_rows(hip_engine):
    rng = np.random.default_rng(11)
    m, n = 3000, 5000
    lens = rng.integers(0, 9, size=m)
    lens[::7] = 0                                   # empty rows
    lens[100] = 2500                                # longer than one LDS chunk (2048)
    lens[2999] = 4099
    rows = np.repeat(np.arange(m), lens)
    cols = np.concatenate([rng.choice(n, size=k, replace=False) for k in lens])
    vals = rng.standard_normal(rows.size)
    mat = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    _spmv_check(hip_engine, mat)
    _spmv_check(hip_engine, mat, seed=2, alpha=0.3, beta=1.0)
    # all-empty matrix and single-row matrix
    _spmv_check(hip_engine, sp.csr_matrix((50, 40)))
    _spmv_check(hip_engine, sp.csr_matrix(rng.standard_normal((1, 300))))
    # many very short rows (block row cap)
    _spmv_check(hip_engine, sp.identity(20000, format="csr"))


def test_spmv_16_bit_column_offsets(hip_engine):
    """Operators whose row blocks touch at most 16 windows of 4096 columns stream 2-byte indices
    (4-bit window + 12-bit offset, nss_csr_index_width), scattered ones the 4-byte indices; the
    products are the same either way."""
    import hipla
    s = mac_stokes(3, 32)                              # n_u = 95 232 > 2^16: the bases matter
    for mat in (s.A, s.B, s.B.T.tocsr()):
        M = _spmv_check(hip_engine, mat, seed=3)
        assert M.handle.info()["index_bytes"] == 2
        _spmv_check(hip_engine, mat, seed=4, alpha=-1.5, beta=0.5)
    assert hipla.SparseMatrix.from_scipy(s.B).CreateTranspose().handle.info()["index_bytes"] == 2   # device-built
    rng = np.random.default_rng(7)
    m, n = 4000, 300000
    lens = rng.integers(1, 12, size=m)
    rows = np.repeat(np.arange(m), lens)
    cols = np.concatenate([rng.choice(n, size=k, replace=False) for k in lens])
    wide = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    M = _spmv_check(hip_engine, wide)
    assert M.handle.info()["index_bytes"] == 4
    # exactly at the limit: 16 column windows of 4096 per row block fit, 17 do not
    for windows, width in ((16, 2), (17, 4)):
        cols = np.arange(windows) * 4096 + 5
        edge = sp.csr_matrix((np.arange(1.0, windows + 1), (np.zeros(windows, dtype=int), cols)), shape=(1, 80000))
        M = _spmv_check(hip_engine, edge)
        assert M.handle.info()["index_bytes"] == width
    # a row longer than one LDS chunk does not count against the windows (it reads 4-byte indices)
    lens = np.full(600, 5)
    lens[300] = 3000
    rows = np.repeat(np.arange(600), lens)
    cols = np.concatenate([np.sort(rng.choice(200000, size=3000, replace=False)) if k == 3000
                           else 1000 + r + np.arange(5) for r, k in enumerate(lens)])
    mixed = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(600, 200000))
    M = _spmv_check(hip_engine, mixed)
    assert M.handle.info()["index_bytes"] == 2


def test_spmv_cfg1_heat_matrix(hip_engine):
    M = diffusion_2d(64)
    assert M.shape == (4096, 4096) and M.nnz == 20224
    _spmv_check(hip_engine, M)


def test_spmv_rejects_aliasing_and_bad_shapes(hip_engine):
    import hipla
    from hipla.hip_engine import NssError
    s = mac_stokes(2, 8)
    A = hipla.SparseMatrix.from_scipy(s.A)
    x = hipla.Vector(s.n_u)
    with pytest.raises(NssError):
        hip_engine.csr_spmv(A.handle, 1.0, x.buf, 0.0, x.buf)
    with pytest.raises(ValueError):
        hip_engine.csr_spmv(A.handle, 1.0, hipla.Vector(3).buf, 0.0, x.buf)
    y = hipla.Vector(s.n_u)
    x.set_from(np.arange(s.n_u, dtype=float))
    y.data = x
    y.data = A * y                                   # protocol resolves the aliasing through a temp
    assert relerr(y.numpy(), s.A @ np.arange(s.n_u, dtype=float)) < RTOL


@pytest.mark.parametrize("bs", [1, 2, 3, 5, 8, 12, 16])
def test_block_jacobi(hip_engine, bs):
    import hipla
    from oracle import krylov_ref as kr
    s = mac_stokes(2, 20)
    idx = s.line_blocks(bs)
    A = hipla.SparseMatrix.from_scipy(s.A)
    J = hipla.BlockJacobi(A, idx)
    rng = np.random.default_rng(bs)
    x = rng.standard_normal(s.n_u)
    ref = kr.block_jacobi(s.A, idx)(x)
    y = hipla.Vector(s.n_u)
    y.data = J * hipla.Vector.from_numpy(x)
    assert relerr(y.numpy(), ref) < 1e-12
    y0 = rng.standard_normal(s.n_u)
    y.set_from(y0)
    y.data += (-0.5) * J * hipla.Vector.from_numpy(x)
    assert relerr(y.numpy(), y0 - 0.5 * ref) < 1e-12
    # a non-symmetric matrix keeps the full inverse blocks (the symmetric ones are stored packed)
    skew = (s.A + 0.3 * sp.triu(s.A, 1)).tocsr()
    J2 = hipla.BlockJacobi(hipla.SparseMatrix.from_scipy(skew), idx)
    y.data = J2 * hipla.Vector.from_numpy(x)
    assert relerr(y.numpy(), kr.block_jacobi(skew, idx)(x)) < 1e-12


def test_block_jacobi_lists_uncovered_dofs_and_errors(hip_engine):
    import hipla
    from hipla.hip_engine import NssError
    s = mac_stokes(3, 5).inflate(4)
    A = hipla.SparseMatrix.from_scipy(s.A)
    blocks = [list(range(0, 4)), [10, 4, 7], [20], list(range(30, 42))]     # ragged, unordered, 12-block
    J = hipla.BlockJacobi(A, blocks)
    x = np.random.default_rng(0).standard_normal(s.n_u)
    y = hipla.Vector(s.n_u)
    y[:] = 7.0
    y.data = J * hipla.Vector.from_numpy(x)
    ref = np.zeros(s.n_u)
    dense = s.A.toarray()
    for b in blocks:
        ref[b] = np.linalg.solve(dense[np.ix_(b, b)], x[b])
    assert relerr(y.numpy(), ref) < 1e-12          # dofs in no block map to zero
    with pytest.raises((NssError, ValueError)):
        hipla.BlockJacobi(A, [[0, 1], [1, 2]])       # overlapping
    with pytest.raises((NssError, ValueError)):
        hipla.BlockJacobi(A, [[0, s.n_u]])           # out of range
    facet = hipla.BlockJacobi(A, s.facet_blocks())
    assert facet.bs == 12
    from oracle import krylov_ref as kr
    y.data = facet * hipla.Vector.from_numpy(x)
    assert relerr(y.numpy(), kr.block_jacobi(s.A, s.facet_blocks())(x)) < 1e-12


def test_diag_and_jacobi(hip_engine):
    import hipla
    s = mac_stokes(2, 15)
    A = hipla.SparseMatrix.from_scipy(s.A)
    Jp = hipla.JacobiPreconditioner(A)
    x = np.random.default_rng(4).standard_normal(s.n_u)
    y = hipla.Vector(s.n_u)
    y.data = Jp * hipla.Vector.from_numpy(x)
    assert relerr(y.numpy(), x / s.A.diagonal()) < RTOL
    M = hipla.DiagonalMatrix(1.0 / s.mass)
    p = np.random.default_rng(5).standard_normal(s.n_p)
    q = hipla.Vector.from_numpy(p)
    w = hipla.Vector.from_numpy(np.ones(s.n_p))
    w.data = w + (-0.3) * M * q                      # the statement of bramblepasciak_new.py:233
    assert relerr(w.numpy(), 1.0 - 0.3 * p / s.mass) < RTOL
    d = hipla.Vector(s.n_u)
    hip_engine._check(hip_engine.lib.nss_csr_diagonal(A.handle.ptr, d.buf.data_ptr(), hip_engine.stream))
    np.testing.assert_array_equal(d.numpy(), s.A.diagonal())


def test_stream_triad(hip_engine):
    import hipla
    n = (1 << 22) + 1
    rng = np.random.default_rng(6)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    X, Y, Z = hipla.Vector.from_numpy(x), hipla.Vector.from_numpy(y), hipla.Vector(n)
    hip_engine.stream_triad(1.5, X.buf, Y.buf, Z.buf)
    assert relerr(Z.numpy(), x + 1.5 * y) < RTOL


def test_lanczos_scale_factor_matches_oracle(hip_engine):
    import hipla
    from oracle import krylov_ref as kr
    s = mac_stokes(3, 10)
    A = hipla.SparseMatrix.from_scipy(s.A)
    J = hipla.BlockJacobi(A, s.line_blocks(3))
    lams = hipla.la.EigenValues_Preconditioner(mat=A, pre=J, tol=1e-3)
    ref = kr.lanczos_ritz(s.A, kr.block_jacobi(s.A, s.line_blocks(3)), tol=1e-3)
    assert len(lams) == len(ref)
    assert abs(min(lams) - ref.min()) <= 1e-9 * ref.min()
    d = np.load(__import__("conftest").golden_path("stokes3d_n10_bjac_bpcg2"))
    assert abs(1.0 / min(lams) + 1e-3 - float(d["k"])) <= 1e-8 * float(d["k"])


@pytest.mark.parametrize("case", ["line3_2d", "facet_3d_inflated", "ragged_lists"])
def test_multicolour_block_gauss_seidel_sweeps(hip_engine, case):
    """nss_bjac_smooth_f64 (one launch per colour) against the sequential block Gauss-Seidel of
    the oracle over the same colour-major block order; symmetric pair as an operator."""
    import hipla
    from oracle import krylov_ref as kr
    if case == "line3_2d":
        s = mac_stokes(2, 24)
        blocks = s.line_blocks(3)
    elif case == "facet_3d_inflated":
        s = mac_stokes(3, 5).inflate(4)
        blocks = s.facet_blocks()
    else:
        s = mac_stokes(2, 9)
        rng = np.random.default_rng(3)
        perm = rng.permutation(s.n_u)
        cuts = np.sort(rng.choice(np.arange(1, s.n_u), size=s.n_u // 3, replace=False))
        blocks = [list(b) for b in np.split(perm, cuts) if 0 < len(b) <= 16]
    A = hipla.SparseMatrix.from_scipy(s.A)
    G = hipla.BlockGaussSeidel(A, blocks)
    assert G.ncolors >= 2
    rng = np.random.default_rng(1)
    x, y0 = rng.standard_normal(s.n_u), rng.standard_normal(s.n_u)
    X, Y = hipla.Vector.from_numpy(x), hipla.Vector.from_numpy(y0)
    G.Smooth(Y, X)
    ref = kr.block_gauss_seidel_sweep(s.A, G.idx_host, x, y0)
    assert relerr(Y.numpy(), ref) < 1e-12
    G.SmoothBack(Y, X)
    ref = kr.block_gauss_seidel_sweep(s.A, G.idx_host, x, ref, backward=True)
    assert relerr(Y.numpy(), ref) < 1e-12
    out = hipla.Vector(s.n_u)
    out[:] = 5.0
    out.data = G * X
    assert relerr(out.numpy(), kr.symmetric_block_gauss_seidel(s.A, G.idx_host)(x)) < 1e-12
    z = rng.standard_normal(s.n_u)
    gz = hipla.Vector(s.n_u)
    gz.data = G * hipla.Vector.from_numpy(z)
    assert abs(np.dot(out.numpy(), z) - np.dot(x, gz.numpy())) < 1e-10 * np.linalg.norm(x) * np.linalg.norm(gz.numpy())
    out.data = 2.5 * G * X                       # scaled operator (preA = k * preA_unscaled)
    assert relerr(out.numpy(), 2.5 * kr.symmetric_block_gauss_seidel(s.A, G.idx_host)(x)) < 1e-12
    # the default is the colour-major layout inside the sweep (P A P^T, x / y gathered once, one launch per colour with
    # the block solve in the epilogue) over first-fit colours; round 1's form (rows of A permuted only, two launches
    # per colour) with the same colours gives the same bits, and so does every other proper colouring against ITS
    # sequential order
    assert G.layout == "colour-major" and G.coloring_method == "greedy"
    R = hipla.BlockGaussSeidel(A, blocks, colors=G.colors, layout="rows")
    assert R.layout == "rows" and np.array_equal(R.idx_host, G.idx_host)
    for op in ("Smooth", "SmoothBack", "Mult"):
        ya, yb = hipla.Vector.from_numpy(y0), hipla.Vector.from_numpy(y0)
        if op == "Mult":
            G.Mult(X, ya), R.Mult(X, yb)
        else:
            getattr(G, op)(ya, X), getattr(R, op)(yb, X)
        np.testing.assert_array_equal(ya.numpy(), yb.numpy())
    L = hipla.BlockGaussSeidel(A, blocks, coloring_method="luby")
    assert L.coloring_method == "luby" and L.ncolors >= G.ncolors
    yl = hipla.Vector.from_numpy(y0)
    L.Smooth(yl, X)
    assert relerr(yl.numpy(), kr.block_gauss_seidel_sweep(s.A, L.idx_host, x, y0)) < 1e-12
    if case == "line3_2d":
        assert G.ncolors <= 3 and L.ncolors >= 4     # parity colouring of the grid-like block graph vs Luby's MIS tail


def test_csr_transpose_native(hip_engine):
    """nss_csr_transpose (explicit B^T, solvers/bramblepasciak_new.py:198) against scipy, including a
    ragged matrix with empty rows / columns; the cached transpose round-trips."""
    import hipla
    rng = np.random.default_rng(9)
    mats = [mac_stokes(3, 7).B, mac_stokes(2, 11).A,
            sp.random(300, 170, density=0.02, random_state=3, format="csr"), sp.csr_matrix((5, 9))]
    for m in mats:
        m = sp.csr_matrix(m)
        m.sort_indices()
        M = hipla.SparseMatrix.from_scipy(m)
        T = M.CreateTranspose()
        assert (T.height, T.width, T.nnz) == (m.shape[1], m.shape[0], m.nnz) and T.CreateTranspose() is M
        y = rng.standard_normal(m.shape[0])
        out = hipla.Vector(m.shape[1])
        out.data = T * hipla.Vector.from_numpy(y)
        ref = m.T @ y
        assert np.max(np.abs(out.numpy() - ref)) <= RTOL * (np.abs(m.T) @ np.abs(y) + 1e-300).max()
        t_host = T.to_scipy()
        assert abs(t_host - m.T).max() == 0 if m.nnz else True
