"""Driver / harness shape on CPU (numpy checker engine): discretisation factories keep the
reference's `(builder, order)` interface, run.py writes the reference's CSV columns, the
NavierStokes class hands the reference's operands to BramblePasciakCG, the sweep writes
`iterations` / `time`."""

import contextlib
import io
import os

import numpy as np
import pytest


def test_discretization_factories_interface(numpy_engine):
    import discretizations as dz
    mesh = dz.SyntheticMesh(0.1)
    assert mesh.n == 10 and mesh.ne == 100
    for fac, order in [(dz.taylor_hood(2), 2), (dz.taylor_hood(3), 3), (dz.mini(), 1),
                       (dz.P1_nonconforming_velocity_constant_pressure(), 1), (dz.P2_velocity_constant_pressure(), 2),
                       (dz.P2_velocity_linear_pressure(), 2), (dz.P2_velocity_with_cubic_bubbles_linear_pressure(), 2),
                       (dz.bdm_hybrid(2, 10), 2), (dz.rt_hybrid(1, 10), 1)]:
        builder, o = fac
        assert o == order
        V, Q = builder(mesh, velocity_dirichlet="wall|inlet|cyl")
        a, b, mp, f, g, s = dz.assemble(V, Q)
        assert a.mat.height == a.mat.width == V.ndof == f.vec.size
        assert b.mat.height == Q.ndof == g.vec.size and b.mat.width == V.ndof
        assert mp.mat.height == Q.ndof and not a.condense
    V, S, Q = dz.hcurldiv(2)[0](mesh, velocity_dirichlet="wall", velocity_neumann="outlet")
    assert S.role == "stress"
    # hybrid H(div), order 2, 2-D: ~5 dofs per facet -> ~25 non-zeros per row
    V, Q = dz.bdm_hybrid(2, 10)[0](mesh, velocity_dirichlet="wall")
    s = dz.system_of(V)
    assert 20 <= s.A.nnz / s.A.shape[0] <= 25 and s.block_size == 5
    mesh3 = dz.SyntheticMesh(0.25, dim=3)
    V3, _ = dz.bdm_hybrid(2, 10)[0](mesh3, velocity_dirichlet="wall")
    assert dz.system_of(V3).block_size == 12


def test_run_harness_writes_reference_csv_schema(numpy_engine, tmp_path):
    import run as harness
    from discretizations import bdm_hybrid, taylor_hood
    methods = {"mixed": {"solve": harness.solve, "discretizations": {"taylor hood 2": taylor_hood(2)}},
               "hybrid_dg": {"solve": harness.solve_hybrid, "discretizations": {"HDG BDM 1": bdm_hybrid(1, 10)}}}
    out = tmp_path / "errors.csv"
    with contextlib.redirect_stdout(io.StringIO()):
        data = harness.run([0.125], methods, harness.solver_factories, str(out), False)
    assert list(data.columns) == ['mesh_size', 'discretization', 'order', 'solver', 'iteration', 'error',
                                  'solver_time', 'nvertices', 'nedges', 'nfaces', 'nfacets', 'nelements', 'ndofs',
                                  'method']
    assert set(data.solver) == {"bramble pasciak cg", "minres"} and os.path.exists(out)
    for _, grp in data.groupby(["discretization", "solver"]):
        assert grp.error.iloc[0] == 1.0 and grp.error.iloc[-1] < 1e-6 and (grp.solver_time > 0).all()
        assert list(grp.iteration) == list(range(len(grp)))
    assert harness.data_file(["-p", "x.csv"]) == "x.csv" and harness.profiling_enabled(["-p"])
    ops = __import__("discretizations").assemble(*bdm_hybrid(1, 10)[0](harness.create_mesh(0.25), "w"))[:3]
    with pytest.raises(NotImplementedError):             # sparse direct solvers are not part of the path
        harness.create_iterative_solver_factory(harness.solve_with_min_res, "direct", "local", 1e-7, 10)(None, *ops)
    # 'bddc' (the reference's default for A) resolves to the algebraic V-cycle
    harness.create_iterative_solver_factory(harness.solve_with_min_res, "bddc", "local", 1e-7, 10)(None, *ops)


def test_navier_stokes_class_and_sweep(numpy_engine, tmp_path):
    from templates.NavierStokesSIMPLE_iterative import NavierStokes, SyntheticMesh
    from templates.run_navier_stokes_parameter_sweep import sweep
    ns = NavierStokes(SyntheticMesh(0.2, dim=2), nu=0.001, inflow="inlet", outflow="outlet", wall="wall|cyl", uin=None,
                      timestep=0.001, order=2)
    with contextlib.redirect_stdout(io.StringIO()):
        ns.SolveInitial(iterative=True, GS=False, tol=1e-8)
    its_jacobi = ns.stokes_bpcg_iterations
    assert its_jacobi > 5 and ns.stokes_bpcg_time > 0
    ns.gfu[:] = 0.0
    ns.gfup[:] = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        ns.SolveInitial(iterative=True, tol=1e-8)            # reference default GS=True
    assert 3 < ns.stokes_bpcg_iterations <= its_jacobi       # the multiplicative form is at least as strong
    s = ns.system
    f, g = s.rhs(0)
    x = np.concatenate([ns.velocity.numpy(), ns.gfup.numpy()])
    res = np.linalg.norm(np.concatenate([f, g]) - s.saddle_matrix() @ x) / np.linalg.norm(f)
    assert res < 1e-5
    np.testing.assert_array_equal(ns.pressure.numpy(), -ns.gfup.numpy())
    with contextlib.redirect_stdout(io.StringIO()):
        data = sweep([0.25, 0.2], [2, 1], (True, False), out=str(tmp_path / "data.csv"), tol=1e-6)
    assert list(data.columns) == ['mesh_size', 'order', 'iterations', 'time', 'gauss_seidel_enabled']
    from templates.run_navier_stokes_parameter_sweep import sweep_reynolds
    with contextlib.redirect_stdout(io.StringIO()):
        re_data = sweep_reynolds((100, 1000), mesh_size=0.25, dim=3, out=str(tmp_path / "re.csv"), tol=1e-6)
    assert list(re_data.reynolds) == [100, 1000] and (re_data.iterations > 0).all()
    assert len(data) == 8 and (data.iterations > 0).all() and data.gauss_seidel_enabled.sum() == 4
    for _, grp in data.groupby(["mesh_size", "order"]):
        gs = grp[grp.gauss_seidel_enabled].iterations.iloc[0]
        assert gs <= grp[~grp.gauss_seidel_enabled].iterations.iloc[0]    # with the auxiliary term both need few


def test_stokes_hcurldiv_driver(numpy_engine):
    from stokes_hcurldiv import solve_stokes
    with contextlib.redirect_stdout(io.StringIO()):
        sol, errors, (a, b, f, g) = solve_stokes(maxh=0.2, tolerance=1e-8, max_steps=10000)
    assert errors[0] == 1.0 and errors[-1] < 1e-8 and len(sol) == a.mat.height + b.mat.height


def test_multicolour_block_gauss_seidel_against_sequential_oracle(numpy_engine):
    """Scope row N1: colouring is proper, the colour-parallel sweep equals the sequential sweep
    over the same (colour-major) block order, the symmetric pair is a symmetric positive operator
    and BPCG with it matches the oracle loop driven by the sequential sweep."""
    import hipla
    from hipla import coloring
    from oracle import krylov_ref as kr
    from solvers.bramblepasciak_new import BramblePasciakCG
    from staggered_grid import mac_stokes
    for s, blocks in ((mac_stokes(2, 10, 0.01), "line"), (mac_stokes(3, 5, 0.01).inflate(2), "facet")):
        idx = s.line_blocks(3) if blocks == "line" else s.facet_blocks()
        A = hipla.SparseMatrix.from_scipy(s.A)
        g = coloring.block_graph(s.A, idx)
        luby = coloring.color_blocks(g)
        colors = coloring.color_blocks_greedy(g)          # the default: first fit in block order
        assert coloring.check_coloring(g, luby) and luby.min() == 0
        assert coloring.check_coloring(g, colors) and colors.min() == 0 and colors.max() <= luby.max()
        assert hipla.BlockGaussSeidel(A, idx, coloring_method="luby").ncolors == luby.max() + 1
        G = hipla.BlockGaussSeidel(A, idx)
        assert G.ncolors == colors.max() + 1 and sorted(G.idx_host[G.idx_host >= 0]) == list(range(s.n_u))
        rng = np.random.default_rng(0)
        x, y0 = rng.standard_normal(s.n_u), rng.standard_normal(s.n_u)
        y = hipla.Vector.from_numpy(y0)
        G.Smooth(y, hipla.Vector.from_numpy(x))
        ref = kr.block_gauss_seidel_sweep(s.A, G.idx_host, x, y0)
        np.testing.assert_allclose(y.numpy(), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        G.SmoothBack(y, hipla.Vector.from_numpy(x))
        ref = kr.block_gauss_seidel_sweep(s.A, G.idx_host, x, ref, backward=True)
        np.testing.assert_allclose(y.numpy(), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        out, z = hipla.Vector(s.n_u), rng.standard_normal(s.n_u)
        out.data = G * hipla.Vector.from_numpy(x)
        gz = hipla.Vector(s.n_u)
        gz.data = G * hipla.Vector.from_numpy(z)
        assert abs(np.dot(out.numpy(), z) - np.dot(x, gz.numpy())) < 1e-10 * np.linalg.norm(x) * np.linalg.norm(gz.numpy())
        assert np.dot(out.numpy(), x) > 0
        # J.Smooth on a BlockJacobi goes through the same companion handle
        J = hipla.BlockJacobi(A, idx)
        y2 = hipla.Vector(s.n_u)
        J.Smooth(y2, hipla.Vector.from_numpy(x))
        np.testing.assert_allclose(y2.numpy(), kr.block_gauss_seidel_sweep(s.A, J.gauss_seidel().idx_host, x,
                                                                           np.zeros(s.n_u)), atol=1e-12)
    # BPCG v2 with the symmetric sweep as preA: protocol loop vs the oracle loop
    s = mac_stokes(2, 10, 0.01)
    f, gg = s.rhs(0)
    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    G = hipla.BlockGaussSeidel(A, s.line_blocks(3))

    class Form:
        def __init__(self, mat):
            self.mat, self.condense = mat, False

    sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        it, _ = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(gg), G,
                                 hipla.DiagonalMatrix(1.0 / s.mass), sol, tol=1e-8, maxsteps=500)
    pa = kr.symmetric_block_gauss_seidel(s.A, G.idx_host)
    k = kr.scale_factor(kr.lanczos_ritz(s.A, pa, tol=1e-3))
    it_ref, u, p, hist, err0 = kr.bpcg_v2(s.A, s.B, pa, kr.diag_inverse(s.mass), f, gg, k, tol=1e-8, maxsteps=500)
    assert abs(it - it_ref) <= 2
    assert np.linalg.norm(sol.numpy() - np.concatenate([u, p])) < 1e-6 * np.linalg.norm(u)


def test_static_condensation_path(numpy_engine):
    """Scope row N2 through the protocol: (I - E^T)(S + A_ii)(I - E) = A on the synthetic
    partition, and BramblePasciakCG with ``blfA.condense = True`` (harmonic_extension's condensed
    branch + the composite A operator, solvers/bramblepasciak_new.py:11-18,84-109) reproduces the
    uncondensed solve."""
    import hipla
    from discretizations import AssembledForm, CondensedForm
    from solvers.bramblepasciak_new import BramblePasciakCG
    from staggered_grid import mac_stokes
    s = mac_stokes(2, 10, 0.01)
    parts = s.condense()
    n = s.n_u
    import scipy.sparse as sp
    eye = sp.identity(n)
    recon = (eye - parts["harmonic_extension_trans"]) @ (parts["mat"] + parts["inner_matrix"]) @ (eye - parts["harmonic_extension"])
    assert abs(recon - s.A).max() < 1e-14 * abs(s.A).max()
    assert 0.1 * n < parts["interior"].sum() < 0.6 * n
    blfA = CondensedForm(s)
    blfB = AssembledForm(hipla.SparseMatrix.from_scipy(s.B))
    f, g = s.rhs(0)
    preM = hipla.DiagonalMatrix(1.0 / s.mass)
    sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    with contextlib.redirect_stdout(io.StringIO()):
        it, _ = BramblePasciakCG(blfA, blfB, None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                                 blfA.jacobi(), preM, sol, tol=1e-9, maxsteps=3000)
    A = hipla.SparseMatrix.from_scipy(s.A)
    ref = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    with contextlib.redirect_stdout(io.StringIO()):
        it_ref, _ = BramblePasciakCG(AssembledForm(A), blfB, None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                                     hipla.JacobiPreconditioner(A), preM, ref, tol=1e-9, maxsteps=3000)
    x, xr = sol.numpy(), ref.numpy()
    b = np.concatenate([f, g])
    assert np.linalg.norm(b - s.saddle_matrix() @ x) < 1e-5 * np.linalg.norm(b)
    assert np.linalg.norm(x[:n] - xr[:n]) < 1e-5 * np.linalg.norm(xr[:n])
    assert 5 < it < 3000


def test_heat_config1_driver(numpy_engine):
    """BASELINE config 1 through the package's heat.py: CG history and Galerkin spectrum against the
    golden produced with the reference's orthonormalization.py."""
    import heat
    d = np.load(__import__("conftest").golden_path("cfg1_heat_plumbing"))
    x, hist, gal = heat.solve(n=int(d["n"]), seed=int(d["seed"]))
    assert len(hist) - 1 == int(d["cg_iterations"])
    np.testing.assert_allclose(hist, d["cg_history"], rtol=1e-8)
    assert abs(np.linalg.norm(x.numpy()) - float(d["cg_x_norm"])) < 1e-9 * float(d["cg_x_norm"])
    np.testing.assert_allclose(np.linalg.eigvalsh(gal), np.linalg.eigvalsh(d["galerkin"]), rtol=1e-7)


def test_mypre_a_with_auxiliary_space_term(numpy_engine):
    """Scope row N3: MypreA as the reference builds it (templates/NavierStokesSIMPLE_iterative.py:208-391):
    `transform`, per-component `Preconditioner(aH1_c, 'h1amg')` stacked with `Embedding`, additive (:383)
    and multiplicative (:376-381) composition.  The one-handle operator (`AuxiliarySpaceAMG`) equals the
    protocol composition `transform @ preAh1 @ transform.T` the reference writes; both forms of MypreA
    are symmetric positive; the auxiliary term makes the iteration count (nearly) mesh independent."""
    import hipla
    from templates.NavierStokesSIMPLE_iterative import (MypreA, NavierStokes, SyntheticMesh,
                                                       auxiliary_space_preconditioner)
    ns = NavierStokes(SyntheticMesh(0.125, dim=2), nu=0.001, inflow="inlet", outflow="outlet", wall="wall|cyl", uin=None,
                      timestep=0.001, order=1)
    blocks = ns.system.facet_blocks()
    rng = np.random.default_rng(0)
    x, z = rng.standard_normal(ns.V.ndof), rng.standard_normal(ns.V.ndof)
    transform, preAh1, aux = auxiliary_space_preconditioner(ns.system)
    composed = transform @ preAh1 @ transform.T
    ya, yb = hipla.Vector(ns.V.ndof), hipla.Vector(ns.V.ndof)
    ya.data = aux * hipla.Vector.from_numpy(x)
    yb.data = composed * hipla.Vector.from_numpy(x)
    assert np.linalg.norm(ya.numpy() - yb.numpy()) <= 1e-12 * np.linalg.norm(yb.numpy())
    # against scipy: T (sum_c V_c) T^T with exact component solves is what the V-cycles approximate
    space = ns.system.auxiliary_space()
    assert space["transform"].shape == (ns.V.ndof, sum(len(r) for r in space["ranges"]))
    for gs in (False, True):
        P = MypreA(ns.V, ns.a, blocks, GS=gs, aux=aux)
        px, pz = hipla.Vector(ns.V.ndof), hipla.Vector(ns.V.ndof)
        px.data = P * hipla.Vector.from_numpy(x)
        pz.data = P * hipla.Vector.from_numpy(z)
        assert abs(px.numpy() @ z - x @ pz.numpy()) < 1e-9 * abs(px.numpy() @ z) + 1e-12
        assert px.numpy() @ x > 0
    counts = {}
    for n in (8, 16):
        for use_aux in (False, True):
            for gs in (False, True):
                drv = NavierStokes(SyntheticMesh(1.0 / n, dim=2), nu=0.001, inflow="inlet", outflow="outlet",
                                   wall="wall|cyl", uin=None, timestep=0.001, order=1)
                with contextlib.redirect_stdout(io.StringIO()):
                    drv.SolveInitial(iterative=True, GS=gs, aux=use_aux, tol=1e-8)
                counts[(n, use_aux, gs)] = drv.stokes_bpcg_iterations
                s = drv.system
                f, g = s.rhs(0)
                xsol = np.concatenate([drv.gfu.numpy(), drv.gfup.numpy()])
                assert np.linalg.norm(np.concatenate([f, g]) - s.saddle_matrix() @ xsol) < 1e-5 * np.linalg.norm(f)
    for n in (8, 16):
        assert counts[(n, True, False)] < counts[(n, False, False)] and counts[(n, True, True)] < counts[(n, False, True)]
    # halving h: the smoother alone needs ~2x the iterations, with the auxiliary term far less
    assert counts[(16, False, False)] > 1.6 * counts[(8, False, False)]
    assert counts[(16, True, False)] < 1.4 * counts[(8, True, False)]


def test_convection_term_of_the_imex_step(numpy_engine):
    """Scope row N4: `conv_operator * gfu` (templates/NavierStokesSIMPLE_iterative.py:106-113,429) as four
    SpMVs around the donor-cell flux kernel against direct loops over the grid; DoTimeStep uses it."""
    import hipla
    from staggered_grid import mac_stokes
    from templates.NavierStokesSIMPLE_iterative import ConvectionOperator, NavierStokes, SyntheticMesh
    for dim, n in ((2, 7), (3, 4)):
        s = mac_stokes(dim, n, 0.01)
        conv = ConvectionOperator(s)
        u = np.random.default_rng(dim).standard_normal(s.n_u)
        y = hipla.Vector(s.n_u)
        y.data = conv * hipla.Vector.from_numpy(u)
        ref = s.convection_reference(u)
        assert np.abs(y.numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
        y.data = conv * hipla.Vector.from_numpy(-u)              # quadratic: conv(-u) has the avg part unchanged
        assert np.abs(y.numpy() - s.convection_reference(-u)).max() <= 1e-13 * np.abs(ref).max()
    ns = NavierStokes(SyntheticMesh(0.125, dim=2), nu=0.01, inflow="inlet", outflow="outlet", wall="wall|cyl", uin=None,
                      timestep=1e-3, order=1)
    ns.gfu.set_from(0.1 * np.random.default_rng(5).standard_normal(ns.V.ndof))
    with contextlib.redirect_stdout(io.StringIO()):
        ns.Project(ns.gfu)
        u0 = ns.gfu.numpy().copy()
        ns.DoTimeStep()
        with_conv = ns.gfu.numpy().copy()
        ns.gfu.set_from(u0)
        ns.conv_operator = 0.0 * hipla.IdentityMatrix(ns.V.ndof)
        ns.DoTimeStep()
        without = ns.gfu.numpy().copy()
    assert np.all(np.isfinite(with_conv)) and np.linalg.norm(with_conv - without) > 1e-8 * np.linalg.norm(without)
    div = ns.system.B @ with_conv
    assert np.linalg.norm(div) < 1e-5 * np.linalg.norm(ns.system.B @ (u0 + 1.0))   # projected: discretely solenoidal


def test_smoothed_aggregation_restatement_properties(numpy_engine):
    """oracle/krylov_ref.py::sa_* (the CPU restatement csrc/amg_setup.hip is checked against):
    aggregate roots are more than two hops apart, every node belongs to exactly one aggregate
    the strength filter removes weak
    couplings, the Galerkin hierarchy stays symmetric positive definite and the stall guard stops
    on a matrix without strong couplings."""
    import scipy.sparse as sp
    import hipla
    from hipla.amg import build_hierarchy
    from oracle import krylov_ref as kr
    from staggered_grid import mac_stokes
    s = mac_stokes(2, 20, 0.01)
    A = s.A.tocsr()
    A.sort_indices()
    pri = np.random.default_rng(0).permutation(A.shape[0]).astype(np.int64) + 1
    g = kr.sa_strength_graph(A, 0.0)
    assert g.nnz == A.nnz - A.shape[0]
    roots = kr.sa_mis2(g, pri)
    g2 = ((g @ g) + g).tocsr()
    g2.setdiag(0)
    g2.eliminate_zeros()
    r = np.nonzero(roots)[0]
    assert g2[r][:, r].nnz == 0                                   # independent at distance 2
    covered = np.asarray((g2[:, r] != 0).sum(axis=1)).ravel() + roots
    assert covered.min() >= 1                                     # maximal
    agg, nagg = kr.sa_aggregate(A, 0.0, pri)
    assert nagg == np.unique(agg).size and agg.min() == 0 and agg.max() == nagg - 1
    assert np.array_equal(agg[r], np.arange(r.size))              # roots numbered in index order
    # a threshold above the off-diagonal / diagonal ratio leaves no edge -> singletons
    assert kr.sa_strength_graph(A, 0.6).nnz == 0
    assert kr.sa_aggregate(A, 0.6, pri)[1] == A.shape[0]
    P = kr.sa_prolongator(A, agg, nagg, 2.0 / 3.0)
    Ac = kr.sa_spgemm(P.T.tocsr(), kr.sa_spgemm(A, P))
    assert abs(Ac - Ac.T).max() < 1e-12 * abs(Ac).max()
    assert np.linalg.eigvalsh(Ac.toarray()).min() > 0
    # hierarchy through the protocol layer: sizes shrink, stall guard on a diagonal matrix
    levels = build_hierarchy(hipla.SparseMatrix.from_scipy(A), coarse_size=30)
    sizes = [lv["n"] for lv in levels]
    assert sizes == sorted(sizes, reverse=True) and len(sizes) >= 3 and sizes[-1] <= 30
    lone = build_hierarchy(hipla.SparseMatrix.from_scipy(sp.identity(500, format="csr") * 2.0), coarse_size=30)
    assert len(lone) == 1 and "inv" in lone[0]


def test_preconditioner_factory_kinds(numpy_engine):
    """`Preconditioner(form, kind)` for the kinds the reference's drivers ask for."""
    import hipla
    from staggered_grid import mac_stokes
    s = mac_stokes(2, 12, 0.01)
    A = hipla.SparseMatrix.from_scipy(s.A)
    assert isinstance(hipla.Preconditioner(A, "local"), hipla.JacobiPreconditioner)
    assert isinstance(hipla.Preconditioner(A, "local", blocks=s.line_blocks(3)), hipla.BlockJacobi)
    for kind in ("h1amg", "multigrid", "bddc"):
        P = hipla.Preconditioner(A, kind)
        assert isinstance(P, hipla.SmoothedAggregationAMG)
        x = np.random.default_rng(0).standard_normal(s.n_u)
        y = hipla.Vector(s.n_u)
        y.data = P * hipla.Vector.from_numpy(x)
        assert y.numpy() @ x > 0
    with pytest.raises(NotImplementedError):
        hipla.Preconditioner(A, "direct")


def test_time_stepping_orchestration(numpy_engine):
    """Scope row N4: CGSolver as an operator, pressure projection, IMEX step and the pseudo time
    stepping branch of SolveInitial, checked against dense host algebra."""
    import hipla
    from templates.NavierStokesSIMPLE_iterative import NavierStokes, SyntheticMesh
    ns = NavierStokes(SyntheticMesh(0.125, dim=2), nu=0.01, inflow="inlet", outflow="outlet", wall="wall|cyl", uin=None,
                      timestep=0.05, order=1)
    s = ns.system
    ops = ns._time_stepping_operators()
    rng = np.random.default_rng(2)
    # CGSolver = inverse of M_u + tau A to its precision
    r = rng.standard_normal(s.n_u)
    y = hipla.Vector(s.n_u)
    y.data = ops["invmstar"] * hipla.Vector.from_numpy(r)
    mstar = ops["mstar"].to_scipy().toarray()
    exact = np.linalg.solve(mstar, r)
    assert np.linalg.norm(y.numpy() - exact) < 2e-3 * np.linalg.norm(exact) and 1 < ops["invmstar"].iterations < 500
    # projection: divergence-free afterwards, idempotent, pressure = potential
    v0 = rng.standard_normal(s.n_u)
    vel = hipla.Vector.from_numpy(v0)
    ns.Project(vel)
    assert np.linalg.norm(s.B @ vel.numpy()) < 1e-6 * np.linalg.norm(s.B @ v0)
    once = vel.numpy().copy()
    ns.Project(vel)
    assert np.linalg.norm(vel.numpy() - once) < 1e-6 * np.linalg.norm(once)
    # one IMEX step against the dense computation: temp = conv(u) + f - A u (:429-431)
    u0 = once
    ns.gfu.set_from(u0)
    with contextlib.redirect_stdout(io.StringIO()):
        ns.DoTimeStep()
    f = ns.f.vec.numpy()
    cops = s.convection_operators()
    adv, avg, dif = cops["adv"] @ u0, cops["avg"] @ u0, cops["diff"] @ u0
    conv = -(cops["div"] @ (adv * avg - 0.5 * np.abs(adv) * dif))
    assert np.linalg.norm(conv) > 0
    t2 = np.linalg.solve(mstar, conv + f - s.A @ u0)
    m_u = np.full(s.n_u, s.h ** s.dim)
    L = s.B @ np.diag(1.0 / m_u) @ s.B.T
    phi = np.linalg.lstsq(np.asarray(L), s.B @ t2, rcond=None)[0]
    t2p = t2 - (s.B.T @ phi) / m_u
    assert np.linalg.norm(ns.gfu.numpy() - (u0 + ns.timestep * t2p)) < 5e-3 * np.linalg.norm(ns.timestep * t2p)
    assert np.linalg.norm(s.B @ ns.gfu.numpy()) < 1e-5 * np.linalg.norm(ns.gfu.numpy()) * abs(s.B).max()
    # pseudo time stepping (SolveInitial(timesteps=N), :406-417) decays towards the Stokes state of f = 0 forcing
    ns.f.vec[:] = 0.0
    ns.gfu.set_from(once)
    e0 = float(once @ (s.A @ once))
    with contextlib.redirect_stdout(io.StringIO()):
        ns.SolveInitial(timesteps=5)
    u5 = ns.gfu.numpy()
    assert float(u5 @ (s.A @ u5)) < 0.9 * e0 and np.linalg.norm(s.B @ u5) < 1e-5 * np.linalg.norm(u5) * abs(s.B).max()
