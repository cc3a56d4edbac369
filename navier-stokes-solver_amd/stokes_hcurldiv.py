"""Headless counterpart of the reference's ``stokes_hcurldiv.py`` (stokes_hcurldiv.py:1-80):
order-2 hybrid H(div) Stokes problem -> `bramble_pasciak_cg(a.mat, b.mat, None, preA, preM, f.vec,
g.vec, solution=BlockVector([...]), max_steps=10000)` (:75-77).

The reference meshes a channel with a cylinder and assembles an MCS form with NGSolve, GUI calls
included; here the operands come from the synthetic facet-block system of
`discretizations.hcurldiv` (BASELINE.json config 2: ~1e5 DoF) and `preA` is the facet-block
Jacobi of the hot path instead of NGSolve's BDDC (out of scope)."""

import hipla
from hipla import BlockVector
from bramble_pasciak_cg import bramble_pasciak_cg
from discretizations import SyntheticMesh, assemble, hcurldiv


def solve_stokes(maxh=0.06, order=2, dim=2, tolerance=1e-12, max_steps=10000, print_rates=False):
    mesh = SyntheticMesh(maxh, dim=dim)
    mesh.Curve(max(order, 1))
    V, _sigma, Q = hcurldiv(order)[0](mesh, velocity_dirichlet="wall|inlet|cyl", velocity_neumann="outlet")
    a, b, m, f, g, system = assemble(V, Q)
    preA = hipla.Preconditioner(a, "blockjacobi", blocks=system.facet_blocks())
    preM = hipla.Preconditioner(m, "local")
    solution = BlockVector([hipla.Vector(V.ndof), hipla.Vector(Q.ndof)])
    solution, errors = bramble_pasciak_cg(a.mat, b.mat, None, preA, preM, f.vec, g.vec, solution=solution,
                                          tolerance=tolerance, max_steps=max_steps, print_rates=print_rates)
    return solution, errors, (a, b, f, g)


if __name__ == "__main__":
    sol, errs, _ = solve_stokes()
    print("iterations:", len(errs) - 1, "final relative error:", errs[-1])
