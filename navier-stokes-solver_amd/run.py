"""Benchmark harness with the shape of the reference's ``run.py`` (run.py:22-303): mesh sizes x
methods x discretisations x solver factories -> per-iteration relative errors in a CSV with the
reference's columns (run.py:244-259), solver timers as in :32-56.

Differences, all forced by scope: assembly is the synthetic staggered-grid
generator behind `discretizations.py`; preconditioners are the hot-path ones ('local' = point
Jacobi, 'blockjacobi' = additive facet-block Jacobi, 'h1amg' / 'bddc' = the algebraic V-cycle of
`hipla.amg` -- NGSolve's BDDC itself is not available);
no GUI (`Draw`, `input`); importing this module does not start a run (use `main()`)."""

import sys

import pandas as pd

import hipla
from hipla import BlockMatrix, BlockVector
from hipla.ngstd import TaskManager, Timer
from bramble_pasciak_cg import bramble_pasciak_cg
from discretizations import SyntheticMesh, assemble, bdm_hybrid
from minres import MinRes


def create_mesh(mesh_size, dim=2):
    return SyntheticMesh(mesh_size, dim=dim)


def solve_with_bramble_pasciak_cg(a_matrix, b_matrix, pre_a, pre_schur_complement, gfu, gfp, f, g, tolerance, max_steps):
    sol = BlockVector([gfu, gfp])
    timer = Timer("BramblePasciakCG")
    timer.Start()
    (solution, errors) = bramble_pasciak_cg(a_matrix, b_matrix, None, pre_a, pre_schur_complement, f, g, sol,
                                            tolerance=tolerance, max_steps=max_steps)
    timer.Stop()
    print("Bramble Pasciak CG took", timer.time, "seconds")
    return (solution, errors, timer.time)


def solve_with_min_res(a, b, preA, preS, gfu, gfp, f, g, tolerance, max_steps):
    K = BlockMatrix([[a, b.T], [b, None]])
    C = BlockMatrix([[preA, None], [None, preS]])
    rhs = BlockVector([f, g])
    sol = BlockVector([gfu, gfp])
    timer = Timer("MinRes")
    timer.Start()
    (solution, errors) = MinRes(mat=K, pre=C, rhs=rhs, sol=sol, initialize=False, tol=tolerance, maxsteps=max_steps)
    timer.Stop()
    print("MinRes took", timer.time, "seconds")
    return (solution, errors, timer.time)


def _preconditioner(form, kind, system=None):
    if kind == "local":
        return hipla.Preconditioner(form, "local")
    if kind == "blockjacobi":
        return hipla.Preconditioner(form, "blockjacobi", blocks=system.facet_blocks())
    return hipla.Preconditioner(form, kind)          # 'h1amg' / 'multigrid' / 'bddc' -> algebraic V-cycle


def create_iterative_solver_factory(solver, a_pre, schur_complement_pre, tolerance, max_steps):
    def create_iterative_solver(space, a, b, m, system=None):
        pre_a = _preconditioner(a, a_pre, system)
        pre_schur_complement = _preconditioner(m, schur_complement_pre, system)

        def solve(a_matrix, b_matrix, gfu, gfp, f, g):
            return solver(a_matrix, b_matrix, pre_a, pre_schur_complement, gfu, gfp, f, g, tolerance, max_steps)
        return solve
    return create_iterative_solver


def solve_hybrid(mesh, discretization, solver_factory):
    """Counterpart of run.py:114-172 (and of `solve`, :71-111): spaces -> assembled forms ->
    solver(a.mat, b.mat, gfu.vec, gfp.vec, f.vec, g.vec)."""
    V, Q = discretization(mesh, velocity_dirichlet='wall|inlet|cyl')
    a, b, mp, f, g, system = assemble(V, Q)
    solver = solver_factory((V, Q), a, b, mp, system)
    velocity = hipla.Vector(V.ndof)
    pressure = hipla.Vector(Q.ndof)
    solution, errors, time = solver(a.mat, b.mat, velocity, pressure, f.vec, g.vec)
    return (velocity, pressure, errors, time, V.ndof + Q.ndof)


solve = solve_hybrid


def profiling_enabled(argv=None):
    return '-p' in (sys.argv[1:] if argv is None else argv)


def data_file(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    return next((argument for argument in argv if not argument.startswith('-')), "errors.csv")


def run(mesh_sizes, methods, solver_factories, data_file, profiling_enabled, profiling_file_size=100 * 1000 * 1000):
    frames = []
    for mesh_size in mesh_sizes:
        mesh = create_mesh(mesh_size=mesh_size)
        for method_name, method_map in methods.items():
            solve_method = method_map['solve']
            for discretization_name, (discretization, order) in method_map['discretizations'].items():
                for solver_name, solver in solver_factories.items():
                    print("solving with", ", ".join([discretization_name, solver_name, "h=" + str(mesh_size)]))
                    with TaskManager(pajetrace=profiling_file_size) if profiling_enabled else TaskManager():
                        solution, _, errors, solver_time, ndofs = solve_method(mesh, discretization, solver)
                    print("\n")
                    frames.append(pd.DataFrame({
                        'mesh_size': mesh_size, 'discretization': discretization_name, 'order': order,
                        'solver': solver_name, 'iteration': range(len(errors)), 'error': errors,
                        'solver_time': solver_time, 'nvertices': mesh.nv, 'nedges': mesh.nedge,
                        'nfaces': mesh.nface, 'nfacets': mesh.nfacet, 'nelements': mesh.ne, 'ndofs': ndofs,
                        'method': method_name}))
    data = pd.concat(frames, ignore_index=True)
    data.to_csv(data_file)
    return data


mesh_sizes = [0.1]
methods = {'hybrid_dg': {'solve': solve_hybrid, 'discretizations': {"HDG BDM 2": bdm_hybrid(2, 10)}}}
solver_factories = {
    "bramble pasciak cg": create_iterative_solver_factory(solve_with_bramble_pasciak_cg, a_pre='blockjacobi',
                                                          schur_complement_pre='local', tolerance=1e-7,
                                                          max_steps=10000),
    "minres": create_iterative_solver_factory(solve_with_min_res, a_pre='blockjacobi',
                                              schur_complement_pre='local', tolerance=1e-7, max_steps=10000),
}


def main(argv=None):
    print("profiling_enabled:", profiling_enabled(argv))
    print("data file:", data_file(argv))
    return run(mesh_sizes, methods, solver_factories, data_file(argv), profiling_enabled(argv))


if __name__ == "__main__":
    main()
