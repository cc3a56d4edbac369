"""``NavierStokes`` with the reference's constructor / ``SolveInitial`` surface
(templates/NavierStokesSIMPLE_iterative.py:13-157,168-399), reduced to what the test and sweep
drivers exercise: the *iterative Stokes initial solve*, i.e. the hand-off

    BramblePasciakCG(blfA, blfB, None, f.vec, g.vec, preA, preM, sol,
                     initialize=False, tol=1e-10, maxsteps=100000, rel_err=True)          (:397)

with ``preA = MypreA(X2, blfA, blocks, GS)`` (:364-391) and ``preM = Preconditioner(mass, 'local')``
(:197-200).  ``MypreA`` is built the way the reference builds it:

* ``blocks`` = the free dofs of every mesh facet (:360-362) -> ``a.mat.CreateBlockSmoother(blocks)``;
* the auxiliary space (:150-157): one P1-like nodal space per velocity component, its ``nu``-scaled
  Laplacian ``aH1_c`` with ``Preconditioner(aH1_c, 'h1amg')`` (:320-351), stacked by ``Embedding``
  (:334-337,353-357) to ``preAh1``, and the ``transform`` from nodal fields to the facet unknowns
  (:208-291) -- FE machinery in the reference, here their grid restatement
  (``StokesSystem.auxiliary_space``);
* ``GS=False``: ``y = ((transform @ preAh1 @ transform.T) + jacobi) * x`` (:383);
  ``GS=True`` (the reference's default): ``y = 0; jacobi.Smooth(y, x); temp = x - mat*y;
  y += (transform @ preAh1 @ transform.T) * temp; jacobi.SmoothBack(y, x)`` (:376-381), the sweeps over a
  multicolour block ordering (scope row N1).
  Both forms are applied natively inside the fused BPCG loop (one ``nss_amg_create_auxiliary`` handle
  for the auxiliary term); ``aux=False`` drops the auxiliary term (block smoother only).

``DoTimeStep`` / ``Project`` / ``SolveInitial(timesteps=N)`` (:400-443, scope row N4) are the
reference's orchestration restated on the staggered-grid operators: ``invmstar`` = CG on
``M_u + timestep * A`` (:85-96), ``Project`` = pressure projection through CG on ``B M_u^-1 B^T``
(:115-144,440-443), explicit Euler update ``u += timestep * temp2`` (:438), and the explicit
convection term ``conv_operator * gfu`` (:106-113,427-431): conservative upwind fluxes
(``ConvectionOperator``: three SpMVs, the donor-cell flux kernel, one SpMV).

Out of scope: the MCS/HDG assembly itself and the sparse direct branch
``iterative=False`` (raises ``NotImplementedError``)."""

import numpy as np
import scipy.sparse as sp

import hipla
from hipla import BlockVector, CGSolver
from discretizations import AssembledForm, SyntheticMesh, assemble, bdm_hybrid
from solvers.bramblepasciak_new import BramblePasciakCG

__all__ = ["NavierStokes", "SyntheticMesh", "MypreA"]


class ConvectionOperator(hipla.BaseMatrix):
    """``conv_operator`` of the reference (templates/NavierStokesSIMPLE_iterative.py:106-113): the
    *nonlinear* map u -> conv(u), weak form of -div(u (x) u) with upwind fluxes, used as
    ``temp.data = conv_operator * gfu.vec`` (:429).  On the staggered grid: donor-cell fluxes
    ``F = adv*avg - |adv|*diff/2`` with ``adv = I_adv u``, ``avg = Avg u``, ``diff = Diff u`` (three
    SpMVs), then ``conv = -D F`` (one SpMV); all on the device."""

    def __init__(self, system):
        super().__init__()
        ops = system.convection_operators()
        self.n = system.n_u
        self.adv, self.avg, self.diff, self.div = (hipla.SparseMatrix.from_scipy(ops[k])
                                                   for k in ("adv", "avg", "diff", "div"))
        self._work = [self.adv.CreateColVector() for _ in range(4)]

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def Mult(self, x, y):
        adv, avg, diff, flux = self._work
        adv.data = self.adv * x
        avg.data = self.avg * x
        diff.data = self.diff * x
        flux.engine.upwind_flux(adv.buf, avg.buf, diff.buf, flux.buf)
        y.data = -self.div * flux


def auxiliary_space_preconditioner(system, space=None):
    """``transform`` and ``preAh1`` of the reference (:208-357) on the grid restatement of the auxiliary
    space: returns (transform, preAh1, aux) with ``preAh1 = sum_c emb_c @ Preconditioner(aH1_c, 'h1amg') @
    emb_c.T`` as the protocol composition the reference writes (:336-337,357) and ``aux`` = the same
    operator ``transform @ preAh1 @ transform.T`` as one native handle (`hipla.AuxiliarySpaceAMG`)."""
    if space is None:                  # (`space`: an assembled `system.auxiliary_space()`, so that callers can time the
        space = system.auxiliary_space()   #  host assembly of the auxiliary operators apart from the preconditioner set-up)
    transform = hipla.SparseMatrix.from_scipy(space["transform"])
    ndof = transform.width
    comps, preAh1 = [], None
    built = {}
    for lap, rng in zip(space["laplacians"], space["ranges"]):
        if id(lap) not in built:       # components with the same boundary conditions share one matrix: one hierarchy
            aH1 = AssembledForm(hipla.SparseMatrix.from_scipy(lap))
            built[id(lap)] = hipla.Preconditioner(aH1, "h1amg")          # :326-329,340-349
        pre_c = built[id(lap)]
        emb = hipla.Embedding(ndof, rng)                                 # :334-335,353-355
        term = emb @ pre_c @ emb.T
        preAh1 = term if preAh1 is None else preAh1 + term               # :337,357
        comps.append(pre_c)
    import os
    if os.environ.get("NSS_AUX_FORM", "components") == "stacked":
        # measurement variant: ONE V-cycle on the slab-major stacked block-diagonal Laplacian (what the row-partitioned
        # form applies): a third of the launches on the launch-bound coarse levels, no shared hierarchy
        st = system.auxiliary_space_stacked()
        stacked = hipla.Preconditioner(AssembledForm(hipla.SparseMatrix.from_scipy(st["laplacian"])), "h1amg")
        return transform, preAh1, hipla.AuxiliarySpaceAMG(hipla.SparseMatrix.from_scipy(st["transform"]), [stacked])
    return transform, preAh1, hipla.AuxiliarySpaceAMG(transform, comps)


def MypreA(space, a, jacblocks, GS, aux=None):
    """``MypreA(space, a, jacblocks, GS)`` of the reference
    (templates/NavierStokesSIMPLE_iterative.py:364-391); `aux` is the auxiliary-space term
    ``transform @ preAh1 @ transform.T`` (an `hipla.AuxiliarySpaceAMG`, or any operator), ``None`` = block
    smoother only.

    * ``GS=False`` -> additive ``y = (aux + J) x`` with ``J = a.mat.CreateBlockSmoother(jacblocks)`` (:383);
    * ``GS=True``  -> ``y = 0; J.Smooth(y, x); r = x - A y; y += aux r; J.SmoothBack(y, x)`` (:376-381)
      over a multicolour block ordering (scope row N1).
    Both are native operands of the fused BPCG loop when `aux` is an `AuxiliarySpaceAMG` (or a
    `SmoothedAggregationAMG`); other operators run through the protocol."""
    if GS:
        op = hipla.BlockGaussSeidel(a.mat, jacblocks, middle=aux)
        op.space, op.GS = space, True
        return op
    op = hipla.BlockJacobi(a.mat, jacblocks)
    op.space, op.GS = space, False
    return op if aux is None else aux + op


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
        self._conv_operator = None        # explicit convection term of the IMEX step (:106-113), built on first use
        self._stepping = None

    @property
    def conv_operator(self):
        if self._conv_operator is None:
            self._conv_operator = ConvectionOperator(self.system)
        return self._conv_operator

    @conv_operator.setter
    def conv_operator(self, op):
        self._conv_operator = op

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
                     aux=True, amg=False):
        """`aux`: build the auxiliary-space term of MypreA (:208-357) -- the reference always does; False
        keeps the block smoother alone.  `amg=True` (kept from round 1) puts a smoothed-aggregation
        V-cycle on a.mat itself in the place of the auxiliary term."""
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
        middle = None
        if amg:
            middle = hipla.SmoothedAggregationAMG(blfA.mat)
        elif aux:
            self.transform, self.preAh1, middle = auxiliary_space_preconditioner(self.system)
        preA = MypreA(self.V, blfA, self.system.facet_blocks(), GS=GS, aux=middle)
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
        temp.data = self.conv_operator * self.gfu        # :429
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
