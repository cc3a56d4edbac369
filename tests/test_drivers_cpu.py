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
    with pytest.raises(NotImplementedError):
        harness.create_iterative_solver_factory(harness.solve_with_min_res, "bddc", "local", 1e-7, 10)(
            None, *__import__("discretizations").assemble(*bdm_hybrid(1, 10)[0](harness.create_mesh(0.25), "w"))[:3])


def test_navier_stokes_class_and_sweep(numpy_engine, tmp_path):
    from templates.NavierStokesSIMPLE_iterative import NavierStokes, SyntheticMesh
    from templates.run_navier_stokes_parameter_sweep import sweep
    ns = NavierStokes(SyntheticMesh(0.2, dim=2), nu=0.001, inflow="inlet", outflow="outlet", wall="wall|cyl", uin=None,
                      timestep=0.001, order=2)
    with pytest.raises(NotImplementedError):
        ns.SolveInitial(iterative=True)                      # reference default GS=True: scope row N1
    with pytest.raises(NotImplementedError):
        ns.DoTimeStep()
    with contextlib.redirect_stdout(io.StringIO()):
        ns.SolveInitial(iterative=True, GS=False, tol=1e-8)
    assert ns.stokes_bpcg_iterations > 5 and ns.stokes_bpcg_time > 0
    s = ns.system
    f, g = s.rhs(0)
    x = np.concatenate([ns.velocity.numpy(), ns.gfup.numpy()])
    res = np.linalg.norm(np.concatenate([f, g]) - s.saddle_matrix() @ x) / np.linalg.norm(f)
    assert res < 1e-5
    np.testing.assert_array_equal(ns.pressure.numpy(), -ns.gfup.numpy())
    with contextlib.redirect_stdout(io.StringIO()):
        data = sweep([0.25, 0.2], [2, 1], (True, False), out=str(tmp_path / "data.csv"), tol=1e-6)
    assert list(data.columns) == ['mesh_size', 'order', 'iterations', 'time', 'gauss_seidel_enabled']
    assert len(data) == 4 and (data.iterations > 0).all() and not data.gauss_seidel_enabled.any()


def test_stokes_hcurldiv_driver(numpy_engine):
    from stokes_hcurldiv import solve_stokes
    with contextlib.redirect_stdout(io.StringIO()):
        sol, errors, (a, b, f, g) = solve_stokes(maxh=0.2, tolerance=1e-8, max_steps=10000)
    assert errors[0] == 1.0 and errors[-1] < 1e-8 and len(sol) == a.mat.height + b.mat.height
