"""Discretisation factories with the reference's interface (discretizations.py:6-88):
each factory returns ``(builder, order)`` and ``builder(mesh, velocity_dirichlet[,
velocity_neumann])`` returns the velocity / pressure "spaces" the drivers assemble on.

NGSolve's FE spaces are outside the hot-path scope (SURVEY.md section 2, row "FE-space
factories": *interface kept, synthetic generators behind it*).  Here a *space* is a light
handle on a staggered-grid Stokes system (`staggered_grid.mac_stokes`) whose sparsity is
inflated to mimic the family: H1-conforming pairs keep the 5-/7-point rows, the hybrid H(div)
families get the facet-block structure (about 5 dofs per facet in 2-D and 12 in 3-D at
order 2 -> ~25 / ~84 non-zeros per row).  The drivers then call
``assemble(V, Q)`` to obtain ``a``, ``b``, ``mp`` (objects with ``.mat``), ``f``, ``g``."""

import numpy as np

import hipla
from staggered_grid import mac_stokes

__all__ = ["taylor_hood", "P1_nonconforming_velocity_constant_pressure", "P2_velocity_constant_pressure",
           "P2_velocity_linear_pressure", "P2_velocity_with_cubic_bubbles_linear_pressure", "mini",
           "bdm_hybrid", "rt_hybrid", "hcurldiv", "SyntheticMesh", "assemble", "AssembledForm", "CondensedForm"]


class SyntheticMesh:
    """Stand-in for a Netgen mesh: the unit square / cube with n = round(1/maxh) cells per
    direction and the counts `run.py:252-256` records."""

    def __init__(self, maxh, dim=2, nu=1.0):
        self.dim = int(dim)
        self.n = max(2, int(round(1.0 / float(maxh))))
        self.maxh = float(maxh)
        n, d = self.n, self.dim
        self.nv = (n + 1) ** d
        self.ne = n ** d
        self.nfacet = d * n ** (d - 1) * (n + 1)
        self.nedge = self.nfacet if d == 2 else 3 * n * (n + 1) ** 2
        self.nface = self.ne if d == 2 else self.nfacet

    def Curve(self, order):
        return self


class _Space:
    def __init__(self, mesh, role, family, order, dofs_per_site, dirichlet=None):
        self.mesh, self.role, self.family, self.order = mesh, role, family, order
        self.dofs_per_site = int(dofs_per_site)
        self.dirichlet = dirichlet
        self._system = None

    @property
    def ndof(self):
        s = system_of(self)
        return s.n_u if self.role == "velocity" else s.n_p


def system_of(space, nu=1.0):
    """The (cached) Stokes system behind a velocity/pressure space pair of one mesh + family."""
    key = (space.family, space.order, space.dofs_per_site, float(nu))
    cache = space.mesh.__dict__.setdefault("_systems", {})
    if key not in cache:
        s = mac_stokes(space.mesh.dim, space.mesh.n, nu)
        if space.dofs_per_site > 1:
            s = s.inflate(space.dofs_per_site)
        cache[key] = s
    return cache[key]


class AssembledForm:
    """BilinearForm-like result of `assemble`: ``.mat`` plus the flags
    ``solvers/bramblepasciak_new.py:105-109`` reads."""

    def __init__(self, mat, space=None):
        self.mat = mat
        self.space = space
        self.condense = False

    def Assemble(self):
        return self


class CondensedForm:
    """Statically condensed BilinearForm-like operand (``condense=True, store_inner=True`` of
    templates/NavierStokesSIMPLE_iterative.py:188): ``.mat`` is the Schur complement on the
    coupling dofs, plus the harmonic-extension / inner-solve operators
    solvers/bramblepasciak_new.py:11-17,88 applies.  All operators are CSR matrices in HBM."""

    def __init__(self, system, seed=0):
        parts = system.condense(seed)
        self.condense = True
        self.interior = parts["interior"]
        self.mat = hipla.SparseMatrix.from_scipy(parts["mat"])
        self.inner_matrix = hipla.SparseMatrix.from_scipy(parts["inner_matrix"])
        self.inner_solve = hipla.SparseMatrix.from_scipy(parts["inner_solve"])
        self.harmonic_extension = hipla.SparseMatrix.from_scipy(parts["harmonic_extension"])
        self.harmonic_extension_trans = hipla.SparseMatrix.from_scipy(parts["harmonic_extension_trans"])
        self.schur_host = parts["mat"]

    def jacobi(self):
        """Point Jacobi of the Schur complement on the coupling dofs (zero on the interior ones) --
        a preconditioner for the condensed system, as `Preconditioner(blfA, ...)` is in the
        reference when the form is condensed."""
        d = self.schur_host.diagonal()
        inv = np.where(self.interior, 0.0, 1.0 / np.where(d != 0.0, d, 1.0))
        return hipla.DiagonalMatrix(inv)


class AssembledVector:
    def __init__(self, vec):
        self.vec = vec

    def Assemble(self):
        return self


def assemble(V, Q, nu=1.0, seed=0):
    """-> (a, b, mp, f, g, system): `a.mat` (n_u x n_u), `b.mat` (n_p x n_u), `mp.mat` (lumped
    pressure mass), `f.vec`, `g.vec` in HBM, and the host-side `StokesSystem` (for blocks)."""
    s = system_of(V, nu)
    a = AssembledForm(hipla.SparseMatrix.from_scipy(s.A), V)
    b = AssembledForm(hipla.SparseMatrix.from_scipy(s.B), Q)
    import scipy.sparse as sp
    mp = AssembledForm(hipla.SparseMatrix.from_scipy(sp.diags(s.mass).tocsr()), Q)
    fh, gh = s.rhs(seed)
    f = AssembledVector(hipla.Vector.from_numpy(fh))
    g = AssembledVector(hipla.Vector.from_numpy(gh))
    return a, b, mp, f, g, s


def _pair(mesh, family, order, dofs_per_site, velocity_dirichlet):
    return (_Space(mesh, "velocity", family, order, dofs_per_site, velocity_dirichlet),
            _Space(mesh, "pressure", family, order, dofs_per_site))


def _h1_family(name, order, dofs_per_site=1):
    def discretization(mesh, velocity_dirichlet):
        return _pair(mesh, name, order, dofs_per_site, velocity_dirichlet)
    return (discretization, order)


def taylor_hood(order):
    return _h1_family("taylor_hood", order, dofs_per_site=max(1, order - 1))


def P1_nonconforming_velocity_constant_pressure():
    return _h1_family("P1nc-P0", 1)


def P2_velocity_constant_pressure():
    return _h1_family("P2-P0", 2)


def P2_velocity_linear_pressure():
    return _h1_family("P2-P1dc", 2)


def P2_velocity_with_cubic_bubbles_linear_pressure():
    return _h1_family("P2+-P1dc", 2)


def mini():
    return _h1_family("mini", 1)


def _facet_dofs(dim, order):
    """dofs per mesh facet of the hybrid H(div) velocity space (normal BDM + tangential facet
    dofs): about 5 in 2-D and 12 in 3-D at order 2."""
    return (2 * order + 1) if dim == 2 else 3 * order * order


def bdm_hybrid(order, penalty, hodivfree=False):
    def discretization(mesh, velocity_dirichlet):
        return _pair(mesh, "hdg-bdm", order, _facet_dofs(mesh.dim, order), velocity_dirichlet)
    return (discretization, order)


def rt_hybrid(order, penalty, hodivfree=False):
    def discretization(mesh, velocity_dirichlet):
        return _pair(mesh, "hdg-rt", order, _facet_dofs(mesh.dim, order + 1), velocity_dirichlet)
    return (discretization, order)


def hcurldiv(order, raviart_thomas=True):
    def discretization(mesh, velocity_dirichlet, velocity_neumann):
        V, Q = _pair(mesh, "mcs", order, _facet_dofs(mesh.dim, max(1, order)), velocity_dirichlet)
        sigma = _Space(mesh, "stress", "mcs", order, 1, velocity_neumann)
        return (V, sigma, Q)
    return (discretization, order)
