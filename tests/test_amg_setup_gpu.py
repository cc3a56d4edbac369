"""Device-side AMG set-up (csrc/amg_setup.hip) against its CPU restatement
(oracle/krylov_ref.py::sa_*): integer results identical, floating-point results bit-identical
(the device sums every product in the same order, each operation rounded once)."""

import numpy as np
import pytest
import scipy.sparse as sp

from staggered_grid import mac_stokes

pytestmark = pytest.mark.gpu


def _random_csr(rng, m, n, density, empty_every=0):
    mat = sp.random(m, n, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                    data_rvs=rng.standard_normal, format="lil")
    if empty_every:
        for r in range(0, m, empty_every):
            mat.rows[r], mat.data[r] = [], []
    mat = mat.tocsr()
    mat.sort_indices()
    return mat


def _same_csr(dev, ref):
    rowptr, col, val = dev.host_csr()
    ref = sp.csr_matrix(ref)
    ref.sort_indices()
    assert (dev.height, dev.width, dev.nnz) == (ref.shape[0], ref.shape[1], ref.nnz)
    np.testing.assert_array_equal(rowptr, ref.indptr)
    np.testing.assert_array_equal(col, ref.indices)
    np.testing.assert_array_equal(val, ref.data)          # bit-identical


def test_device_transpose(hip_engine):
    import hipla
    rng = np.random.default_rng(0)
    for m, n, dens, skip in ((1, 1, 1.0, 0), (37, 91, 0.2, 5), (500, 300, 0.03, 0), (40, 2000, 0.01, 3)):
        mat = _random_csr(rng, m, n, dens, skip)
        M = hipla.SparseMatrix.from_scipy(mat)
        _same_csr(M.CreateTranspose(), mat.T.tocsr())
        x = rng.standard_normal(m)
        y = hipla.Vector(n)
        y.data = M.T * hipla.Vector.from_numpy(x)
        assert np.max(np.abs(y.numpy() - mat.T @ x)) <= 1e-13 * (np.abs(mat.T) @ np.abs(x)).max()
    s = mac_stokes(3, 9)
    _same_csr(hipla.SparseMatrix.from_scipy(s.B).CreateTranspose(), s.B.T.tocsr())
    empty = hipla.SparseMatrix.from_scipy(sp.csr_matrix((5, 7)))
    assert empty.CreateTranspose().nnz == 0 and empty.CreateTranspose().height == 7


def test_device_spgemm_is_bitwise_the_rowwise_product(hip_engine):
    import hipla
    from oracle import krylov_ref as kr
    rng = np.random.default_rng(1)
    cases = [(_random_csr(rng, 300, 200, 0.05, 7), _random_csr(rng, 200, 250, 0.08, 5)),
             (_random_csr(rng, 64, 64, 0.5), _random_csr(rng, 64, 64, 0.5)),        # many collisions per entry
             (_random_csr(rng, 10, 30, 0.3), sp.csr_matrix((30, 12)))]              # empty result
    s = mac_stokes(3, 8)
    cases.append((s.A, s.A))
    cases.append((s.B, s.B.T.tocsr()))
    for X, Y in cases:
        ref = kr.sa_spgemm(X, Y)
        Xd, Yd = hipla.SparseMatrix.from_scipy(X), hipla.SparseMatrix.from_scipy(Y)
        for cap in (0, 257):                                                         # one pass / many row chunks
            C = hipla.SparseMatrix.from_handle(hip_engine.csr_spgemm(Xd.handle, Yd.handle, cap))
            _same_csr(C, ref)
    with pytest.raises(ValueError):
        hip_engine.csr_spgemm(hipla.SparseMatrix.from_scipy(s.A).handle, hipla.SparseMatrix.from_scipy(s.B).handle)


@pytest.mark.parametrize("dim,n", [(2, 24), (3, 12)])
def test_device_aggregation_equals_restatement(hip_engine, dim, n):
    import hipla
    from oracle import krylov_ref as kr
    s = mac_stokes(dim, n, 0.01)
    A = s.A.tocsr()
    A.sort_indices()
    for level, theta in enumerate((0.0, 0.04, 0.04)):
        pri = np.random.default_rng(level).permutation(A.shape[0]).astype(np.int64) + 1
        ref_agg, ref_n = kr.sa_aggregate(A, theta, pri)
        Ad = hipla.SparseMatrix.from_scipy(A)
        agg, nagg = hip_engine.amg_aggregate(Ad.handle, theta, pri)
        assert nagg == ref_n
        np.testing.assert_array_equal(hip_engine.index_to_host(agg), ref_agg)
        # every node aggregated, ids dense
        assert ref_agg.min() == 0 and np.unique(ref_agg).size == ref_n
        P_ref = kr.sa_prolongator(A, ref_agg, ref_n, 2.0 / 3.0)
        P = hipla.SparseMatrix.from_handle(hip_engine.amg_prolongator(Ad.handle, agg, nagg, 2.0 / 3.0))
        _same_csr(P, P_ref)
        A = kr.sa_spgemm(P_ref.T.tocsr(), kr.sa_spgemm(A, P_ref))                    # next level
        if A.shape[0] < 50:
            break


def test_isolated_and_weakly_coupled_nodes(hip_engine):
    """Diagonal rows (no neighbours) become singleton roots; a threshold above every coupling
    leaves only singletons."""
    import hipla
    from oracle import krylov_ref as kr
    A = sp.diags([np.full(9, -1.0), np.full(10, 4.0), np.full(9, -1.0)], [-1, 0, 1]).tolil()
    A[4, 3] = A[4, 5] = A[3, 4] = A[5, 4] = 0.0
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    pri = np.arange(10, 0, -1).astype(np.int64)
    for theta in (0.0, 0.2, 0.9):
        ref, nref = kr.sa_aggregate(A, theta, pri)
        agg, nagg = hip_engine.amg_aggregate(hipla.SparseMatrix.from_scipy(A).handle, theta, pri)
        np.testing.assert_array_equal(hip_engine.index_to_host(agg), ref)
        assert nagg == nref
    assert nref == 10                                                                 # theta 0.9 > 1/4


def test_hierarchy_built_on_device_equals_cpu_hierarchy(hip_engine, numpy_engine):
    import hipla
    from hipla.amg import build_hierarchy
    s = mac_stokes(3, 14, 0.01)
    dev = build_hierarchy(hipla.SparseMatrix.from_scipy(s.A, engine=hip_engine), coarse_size=60)
    cpu = build_hierarchy(hipla.SparseMatrix.from_scipy(s.A, engine=numpy_engine), coarse_size=60)
    assert [lv["n"] for lv in dev] == [lv["n"] for lv in cpu] and len(dev) >= 3
    for d, c in zip(dev, cpu):
        _same_csr(d["A"], c["A"].to_scipy())
        np.testing.assert_array_equal(hip_engine.to_host(d["dinv"]), c["dinv"])
        if "P" in d:
            _same_csr(d["P"], c["P"].to_scipy())
            _same_csr(d["R"], c["R"].to_scipy())
        else:
            _same_csr(d["inv"], c["inv"].to_scipy())


def test_device_colouring_and_row_selection(hip_engine):
    """Multicolour ordering on the device (nss_csr_ones_like + two sparse products + nss_graph_color)
    gives exactly the colours of the host algorithm (hipla/coloring.py) with the same priorities,
    and nss_csr_select_rows the row-permuted matrix scipy builds."""
    import hipla
    from hipla import coloring
    for dim, n, bs in ((2, 24, 3), (3, 10, 3), (3, 8, 1)):
        s = mac_stokes(dim, n, 0.01)
        A = s.A.tocsr()
        A.sort_indices()
        idx = s.line_blocks(bs)
        Ad = hipla.SparseMatrix.from_scipy(A)
        dev = hipla.BlockGaussSeidel._device_colors(Ad, idx, 0, "luby")
        graph = coloring.block_graph(A, idx)
        host = coloring.color_blocks(graph, 0)
        np.testing.assert_array_equal(dev, host)
        assert coloring.check_coloring(graph, dev)
        # the default: first fit in block order on the host (nss_graph_color_greedy) == its Python twin; on these
        # grid-like block graphs the parity colouring, fewer colours than the maximal independent sets
        greedy = hipla.BlockGaussSeidel._device_colors(Ad, idx, 0, "greedy")
        np.testing.assert_array_equal(greedy, coloring.color_blocks_greedy(graph))
        assert coloring.check_coloring(graph, greedy) and greedy.max() < dev.max() and greedy.max() <= 3
        rng = np.random.default_rng(dim)
        rows = rng.permutation(A.shape[0])[: A.shape[0] * 2 // 3].astype(np.int32)
        cuts = np.array([0, rows.size // 3, rows.size], dtype=np.int32)
        sel = hipla.SparseMatrix.from_handle(hip_engine.csr_select_rows(Ad.handle, rows, cuts))
        _same_csr(sel, A[rows])
        rb = sel.handle.row_blocks()
        assert rows.size // 3 in rb                      # the launch plan does not span a colour boundary
    # structurally non-symmetric matrix: neighbours through the transposed graph as well
    M = sp.csr_matrix(np.array([[2.0, 1, 0, 0], [0, 2, 1, 0], [0, 0, 2, 1], [0, 0, 0, 2]]))
    g = hipla.SparseMatrix.from_scipy(M)
    colors, ncol = hip_engine.graph_color(g.handle, g.CreateTranspose().handle, np.array([4, 3, 2, 1], dtype=np.int64))
    assert ncol == 2 and all(colors[i] != colors[i + 1] for i in range(3))
    pat = hipla.SparseMatrix.from_handle(hip_engine.csr_ones_like(g.handle))
    np.testing.assert_array_equal(pat.to_scipy().toarray(), (M.toarray() != 0).astype(float))
