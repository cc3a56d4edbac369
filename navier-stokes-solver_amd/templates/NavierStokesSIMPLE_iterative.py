"""``NavierStokes`` with the reference's constructor / ``SolveInitial`` surface
(templates/NavierStokesSIMPLE_iterative.py:13-157,168-399), reduced to what the test and sweep
drivers exercise: the *iterative Stokes initial solve*, i.e. the hand-off

    BramblePasciakCG(blfA, blfB, None, f.vec, g.vec, preA, preM, sol,
                     initialize=False, tol=1e-10, maxsteps=100000, rel_err=True)          (:397)

with ``preA`` = facet-block Jacobi (``MypreA`` with ``GS=False`` minus the AMG term, :364-391) and
``preM`` = ``Preconditioner(mass, 'local')`` (:197-200).

``GS=True`` (the reference's default) selects the symmetric multiplicative block Gauss-Seidel
sweep (:376-381) over a multicolour block ordering, ``GS=False`` the additive block Jacobi.

Out of scope (SURVEY.md sections 2 and 8f): the MCS/HDG assembly, the auxiliary-space AMG
correction (N3), static condensation (N2), the IMEX time stepping ``DoTimeStep`` / ``Project``
(N4) and the sparse direct branch ``iterative=False``.  Those raise ``NotImplementedError``
naming the row."""

import hipla
from hipla import BlockVector
from discretizations import AssembledForm, SyntheticMesh, assemble, bdm_hybrid
from solvers.bramblepasciak_new import BramblePasciakCG

__all__ = ["NavierStokes", "SyntheticMesh", "MypreA"]


class _MultiplicativePreA(hipla.BaseMatrix):
    """``y = 0; J.Smooth(y, x); r = x - A y; y += AMG r; J.SmoothBack(y, x)`` -- the
    ``GS=True`` branch of the reference's MypreA.Mult
    (templates/NavierStokesSIMPLE_iterative.py:376-381) with the build's AMG V-cycle in the place
    of ``transform @ preAh1 @ transform.T``.  Runs through the protocol (sweeps, SpMV, V-cycle are
    device kernels; the composition is host-driven)."""

    def __init__(self, space, a, jacblocks):
        super().__init__()
        self.space, self.mat, self.GS = space, a.mat, True
        self.jacobi = hipla.BlockGaussSeidel(a.mat, jacblocks)
        self.amg = hipla.SmoothedAggregationAMG(a.mat)
        self.temp = a.mat.CreateColVector()

    def Mult(self, x, y):
        y[:] = 0
        self.jacobi.Smooth(y, x)
        self.temp.data = x - self.mat * y
        y.data += self.amg * self.temp
        self.jacobi.SmoothBack(y, x)

    def Height(self):
        return self.mat.height

    def Width(self):
        return self.mat.width


def MypreA(space, a, jacblocks, GS, amg=False):
    """``MypreA(space, a, jacblocks, GS)`` of the reference
    (templates/NavierStokesSIMPLE_iterative.py:364-391).  Its auxiliary-space term
    ``transform @ preAh1 @ transform.T`` (:380,383) is NGSolve FE machinery; ``amg=True`` puts the
    build's smoothed-aggregation V-cycle on ``a.mat`` in its place (scope row N3):

    * ``GS=False`` -> additive ``y = (AMG + J) x`` with ``J = a.mat.CreateBlockSmoother(jacblocks)``
      (:383); without ``amg`` just ``J``.  Native: applied inside the fused loops.
    * ``GS=True``  -> ``y = 0; J.Smooth(y, x); [r = x - A y; y += AMG r;] J.SmoothBack(y, x)``
      (:376-381) over a multicolour block ordering (scope row N1).  Without ``amg`` native (fused
      loops); with ``amg`` the multiplicative composition runs through the protocol."""
    if GS and amg:
        return _MultiplicativePreA(space, a, jacblocks)
    op = hipla.BlockGaussSeidel(a.mat, jacblocks) if GS else hipla.BlockJacobi(a.mat, jacblocks)
    op.space, op.GS = space, GS
    if amg:
        return hipla.SmoothedAggregationAMG(a.mat) + op
    return op


class NavierStokes:
    def __init__(self, mesh, nu, inflow, outflow, wall, uin, timestep, order=2, volumeforce=None):
        self.mesh, self.nu, self.timestep, self.order = mesh, nu, timestep, order
        self.inflow, self.outflow, self.wall, self.uin = inflow, outflow, wall, uin
        self.V, self.Q = bdm_hybrid(order, 10)[0](mesh, velocity_dirichlet=inflow + "|" + wall)
        self.a, self.b, self.mp, self.f, self.g, self.system = assemble(self.V, self.Q, nu=nu)
        self.gfu = hipla.Vector(self.V.ndof)          # velocity dofs (zero start, inflow data not modelled)
        self.gfup = hipla.Vector(self.Q.ndof)
        self.stokes_bpcg_iterations = None
        self.stokes_bpcg_time = None

    @property
    def velocity(self):
        return self.gfu

    @property
    def pressure(self):
        out = self.gfup.CreateVector()
        out.data = -self.gfup                          # reference: pressure = -gfup (:163-165)
        return out

    def SolveInitial(self, timesteps=None, iterative=True, GS=True, tol=1e-10, maxsteps=100000, printrates=False,
                     amg=False):
        if timesteps:
            raise NotImplementedError("projection time stepping: SURVEY.md section 8f row N4")
        if not iterative:
            raise NotImplementedError("sparse direct initial solve is not on the Krylov path")
        blfA = AssembledForm(self.a.mat)
        blfB = AssembledForm(self.b.mat)
        preM = hipla.Preconditioner(self.mp, "local")
        preA = MypreA(self.V, blfA, self.system.facet_blocks(), GS=GS, amg=amg)
        sol = BlockVector([self.gfu, self.gfup])       # aliases the grid-function storage (:206)
        out = BramblePasciakCG(blfA, blfB, None, self.f.vec, self.g.vec, preA, preM, sol, initialize=False,
                               tol=tol, maxsteps=maxsteps, rel_err=True, printrates=printrates)
        if isinstance(out, tuple):
            self.stokes_bpcg_iterations, self.stokes_bpcg_time = out
        else:                                          # zero initial residual: bare vector (:191-192)
            self.stokes_bpcg_iterations, self.stokes_bpcg_time = 0, 0.0

    def DoTimeStep(self):
        raise NotImplementedError("IMEX time stepping: SURVEY.md section 8f row N4")

    def Project(self, vel):
        raise NotImplementedError("pressure projection: SURVEY.md section 8f row N4")
