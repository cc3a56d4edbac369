"""``NavierStokes`` with the reference's constructor / ``SolveInitial`` surface
(templates/NavierStokesSIMPLE_iterative.py:13-157,168-399), reduced to what the test and sweep
drivers exercise: the *iterative Stokes initial solve*, i.e. the hand-off

    BramblePasciakCG(blfA, blfB, None, f.vec, g.vec, preA, preM, sol,
                     initialize=False, tol=1e-10, maxsteps=100000, rel_err=True)          (:397)

with ``preA`` = facet-block Jacobi (``MypreA`` with ``GS=False`` minus the AMG term, :364-391) and
``preM`` = ``Preconditioner(mass, 'local')`` (:197-200).

``GS=True`` (the reference's default) selects the symmetric multiplicative block Gauss-Seidel
sweep (:376-381) over a multicolour block ordering, ``GS=False`` the additive block Jacobi.

``DoTimeStep`` / ``Project`` / ``SolveInitial(timesteps=N)`` (:400-443, scope row N4) are the
reference's orchestration restated on the staggered-grid operators: ``invmstar`` = CG on
``M_u + timestep * A`` (:85-96), ``Project`` = pressure projection through CG on ``B M_u^-1 B^T``
(:115-144,440-443), explicit Euler update ``u += timestep * temp2`` (:438).  The convection term
(:106-113, JIT-compiled nonlinear form) is not modelled: ``conv_operator`` is ``None`` unless the
caller installs an operator.

Out of scope: the MCS/HDG assembly itself and the sparse direct branch
``iterative=False`` (raises ``NotImplementedError``)."""

import numpy as np
import scipy.sparse as sp

import hipla
from hipla import BlockVector, CGSolver
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
        self.conv_operator = None
        self._stepping = None

    def _time_stepping_operators(self):
        """mstar = M_u + timestep*A with its CG inverse (:85-96) and the projection operators
        (:115-144): pressure operator B M_u^-1 B^T, CG inverse, velocity correction M_u^-1 B^T."""
        if self._stepping is None:
            s = self.system
            mass_u = s.h ** s.dim * self.V.dofs_per_site ** 0       # lumped velocity mass: cell volume
            m_u = np.full(s.n_u, mass_u)
            mstar = hipla.SparseMatrix.from_scipy((sp.diags(m_u) + self.timestep * s.A).tocsr())
            invmstar = CGSolver(mstar, pre=hipla.JacobiPreconditioner(mstar), precision=1e-4, maxsteps=500)
            bt_scaled = (sp.diags(1.0 / m_u) @ s.B.T).tocsr()
            lap_p = (s.B @ bt_scaled).tocsr()
            lap_p.sort_indices()
            Lp = hipla.SparseMatrix.from_scipy(lap_p)
            invproj = CGSolver(Lp, pre=hipla.JacobiPreconditioner(Lp), precision=1e-8, maxsteps=5000)
            self._stepping = dict(invmstar=invmstar, invproj=invproj, correct=hipla.SparseMatrix.from_scipy(bt_scaled),
                                  mstar=mstar, Lp=Lp)
        return self._stepping

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
        if timesteps:                                     # pseudo time stepping to the Stokes state (:406-417)
            ops = self._time_stepping_operators()
            self.Project(self.gfu)
            for it in range(timesteps):
                print("it =", it)
                temp = self.a.mat.CreateColVector()
                temp2 = self.a.mat.CreateColVector()
                temp.data = -self.a.mat * self.gfu
                temp2.data = ops["invmstar"] * temp
                self.Project(temp2)
                self.gfu.data += self.timestep * temp2
                self.Project(self.gfu)
            return
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

    def AddForce(self, force):
        """`force`: host array of nodal forces on the velocity dofs, added to f (:419-422)."""
        self.f.vec.data += hipla.Vector.from_numpy(np.asarray(force, dtype=np.float64))

    def DoTimeStep(self):
        """One IMEX step (:424-438): temp = conv(u) + f - A u;  temp2 = invmstar temp;
        Project(temp2);  u += timestep * temp2."""
        ops = self._time_stepping_operators()
        temp = self.a.mat.CreateColVector()
        temp2 = self.a.mat.CreateColVector()
        if self.conv_operator is not None:
            temp.data = self.conv_operator * self.gfu
        else:
            temp[:] = 0.0
        temp.data += self.f.vec
        temp.data += -self.a.mat * self.gfu
        temp2.data = ops["invmstar"] * temp
        self.Project(temp2)
        self.gfu.data += self.timestep * temp2

    def Project(self, vel):
        """Make `vel` discretely divergence-free (:440-443): phi = (B M_u^-1 B^T)^-1 B vel;
        pressure <- phi;  vel -= M_u^-1 B^T phi."""
        ops = self._time_stepping_operators()
        rhs = self.b.mat.CreateColVector()
        rhs.data = self.b.mat * vel
        self.gfup.data = ops["invproj"] * rhs
        vel.data -= ops["correct"] * self.gfup
