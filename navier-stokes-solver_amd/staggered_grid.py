"""Deterministic synthetic assembler: MAC (staggered-grid) Stokes systems.

Netgen meshing / NGSolve FE assembly are outside the hot-path scope (SURVEY.md
section 2, rows "FE-space factories", "MCS Stokes script"); what the Krylov path
needs from them is the triple the drivers hand over --
``(a.mat, b.mat, mass) + (f.vec, g.vec)`` (stokes_hcurldiv.py:75-77,
run.py:105-106,166-167, templates/NavierStokesSIMPLE_iterative.py:397).  This module
produces that triple for the unit square / cube with no-slip walls (SURVEY.md
section 8d): ``A`` = nu * vector Laplacian (5-/7-point, Dirichlet in the normal and
ghost-cell reflection in the tangential directions), ``B`` = discrete divergence,
``M_p`` = lumped pressure mass.  Rows are scaled by the cell volume h^d (the
integrated, FE-like form): A ~ nu h^(d-2), B ~ +-h^(d-1), M_p = h^d.

Unknown ordering is *slab-major*: the slowest grid axis (y in 2-D, z in 3-D) is the
outermost loop, inside a slab the velocity components follow each other, x is the
fastest index.  A contiguous row range is therefore a spatial slab, which is what
the row partition across GPUs cuts.

Sizes: n_u = d n^(d-1) (n-1), n_p = n^d  (n=136, d=3: 7 490 880 + 2 515 456).
Assembly runs on the host with numpy/scipy (set-up, not the solve path).
"""

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp


def _component_ids(dim, n):
    """Global velocity ids per component, arrays indexed [slab, ..., x]."""
    m = n - 1
    if dim == 2:
        per_slab = m + n
        S = np.arange(n, dtype=np.int64) * per_slab
        gu = S[:, None] + np.arange(m, dtype=np.int64)[None, :]                       # (n, n-1)
        gv = S[: n - 1, None] + m + np.arange(n, dtype=np.int64)[None, :]             # (n-1, n)
        n_u = int(n * m + m * n)
        return [gu, gv], S, n_u
    if dim == 3:
        nu_, nv_, nw_ = n * m, m * n, n * n
        per_slab = nu_ + nv_ + nw_
        S = np.arange(n, dtype=np.int64) * per_slab
        j_m = np.arange(n, dtype=np.int64)
        gu = S[:, None, None] + (j_m[None, :, None] * m + np.arange(m, dtype=np.int64)[None, None, :])
        gv = S[:, None, None] + nu_ + (np.arange(m, dtype=np.int64)[None, :, None] * n + j_m[None, None, :])
        gw = S[: n - 1, None, None] + nu_ + nv_ + (j_m[None, :, None] * n + j_m[None, None, :])
        n_u = int(3 * n * n * m)
        return [gu, gv, gw], S, n_u
    raise ValueError("dim must be 2 or 3")


def _axis_of_component(dim, c):
    """Array axis (in [slab, ..., x] order) along which component c is face-centred."""
    return dim - 1 - c       # c=0 (x-normal) -> last axis


def _shift_slices(ndim, axis):
    lo = [slice(None)] * ndim
    hi = [slice(None)] * ndim
    lo[axis] = slice(0, -1)
    hi[axis] = slice(1, None)
    return tuple(lo), tuple(hi)


@dataclass
class StokesSystem:
    dim: int
    n: int
    nu: float
    h: float
    A: sp.csr_matrix
    B: sp.csr_matrix
    mass: np.ndarray
    velocity_slab_offsets: np.ndarray      # length n+1, row offsets of the slabs in A
    pressure_slab_offsets: np.ndarray      # length n+1, row offsets of the slabs in B
    component_ids: list = field(repr=False, default_factory=list)
    block_size: int = 1

    @property
    def n_u(self):
        return self.A.shape[0]

    @property
    def n_p(self):
        return self.B.shape[0]

    @property
    def ndof(self):
        return self.n_u + self.n_p

    def rhs(self, seed=0):
        """f ~ N(0,1) from default_rng(seed), g = 0."""
        rng = np.random.default_rng(seed)
        return rng.standard_normal(self.n_u), np.zeros(self.n_p)

    def saddle_matrix(self):
        return sp.bmat([[self.A, self.B.T], [self.B, None]], format="csr")

    def facet_blocks(self):
        """One block per grid cell: its 'plus' faces (u_{i+1/2}, v_{j+1/2}[, w_{k+1/2}]) --
        the facet-like block-Jacobi variant (bs = dim) of SURVEY.md section 8a row A7.
        Returns an int32 array (bs, nblocks), -1 = padding; blocks are disjoint and
        cover every velocity dof.  For an inflated system the block holds all
        `block_size` copies of each face dof."""
        dim, n = self.dim, self.n
        cells = -np.ones((dim,) + (n,) * dim, dtype=np.int64)
        for c, g in enumerate(self.component_ids):
            ax = _axis_of_component(dim, c)
            sl = [slice(None)] * dim
            sl[ax] = slice(0, n - 1)
            cells[(c,) + tuple(sl)] = g
        idx = cells.reshape(dim, -1)
        keep = (idx >= 0).any(axis=0)
        idx = idx[:, keep]
        if self.block_size > 1:
            b = self.block_size
            rep = np.where(idx[:, None, :] >= 0, idx[:, None, :] * b + np.arange(b)[None, :, None], -1)
            idx = rep.reshape(dim * b, -1)
        return np.ascontiguousarray(idx, dtype=np.int32)

    def line_blocks(self, bs=3):
        """Blocks of `bs` consecutive dofs of one velocity component along x (the
        fastest axis): genuinely coupled tridiagonal A_bb, contiguous in memory.
        Returns int32 (bs * block_size, nblocks), -1 = padding (ragged line ends)."""
        cols = []
        for g in self.component_ids:
            L = g.shape[-1]
            pad = (-L) % bs
            gp = np.concatenate([g, -np.ones(g.shape[:-1] + (pad,), dtype=np.int64)], axis=-1)
            cols.append(gp.reshape(-1, bs).T)
        idx = np.concatenate(cols, axis=1)
        if self.block_size > 1:
            b = self.block_size
            rep = np.where(idx[:, None, :] >= 0, idx[:, None, :] * b + np.arange(b)[None, :, None], -1)
            idx = rep.reshape(bs * b, -1)
        return np.ascontiguousarray(idx, dtype=np.int32)

    def partition(self, nranks):
        """Slab-aligned row ranges for `nranks` GPUs: (velocity offsets, pressure offsets)."""
        cuts = np.round(np.arange(nranks + 1) * self.n / nranks).astype(np.int64)
        return self.velocity_slab_offsets[cuts].copy(), self.pressure_slab_offsets[cuts].copy()

    def permuted(self, perm):
        """The same system with the velocity dofs re-ordered: new dof k = old dof perm[k]
        (A' = P A P^T, B' = B P^T).  Slab offsets / component ids are dropped (the order is no
        longer slab-major)."""
        perm = np.asarray(perm, dtype=np.int64)
        A = self.A[perm][:, perm].tocsr()
        B = self.B[:, perm].tocsr()
        A.sort_indices()
        B.sort_indices()
        out = StokesSystem(self.dim, self.n, self.nu, self.h, A, B, self.mass.copy(),
                           np.zeros(0, dtype=np.int64), self.pressure_slab_offsets.copy(), [], self.block_size)
        out.velocity_permutation = perm
        return out

    def convection_operators(self):
        """Sparse operators of the explicit convection term of the IMEX step
        (templates/NavierStokesSIMPLE_iterative.py:106-113,403,427-431: ``conv_operator * gfu`` = the weak
        form of -div(u (x) u) with upwind numerical fluxes) restated on the staggered grid: conservative
        donor-cell (first-order upwind) fluxes of every velocity component c through the faces of its
        control volume in every direction d,

            F = adv * avg - 1/2 |adv| * diff,      conv(u) = - D F,

        with adv = (I_adv u) the advecting velocity at the flux point, avg = (Avg u), diff = (Diff u) the
        mean and the jump of the transported component across it, and D the (cell-volume-integrated)
        divergence of the fluxes; wall fluxes vanish (no-slip).  Returns dict(adv, avg, diff, div) of CSR
        matrices (flux points x n_u, n_u x flux points).  In an inflated system every one of the
        `block_size` copies of the field is advected by itself (operators (x) I)."""
        dim, n = self.dim, self.n
        hface = self.h ** (dim - 1)
        rows_a, cols_a, vals_a = [], [], []        # I_adv
        rows_m, cols_m, vals_m = [], [], []        # Avg
        rows_j, cols_j, vals_j = [], [], []        # Diff
        rows_d, cols_d, vals_d = [], [], []        # D
        nflux = 0

        def sl(ax, a, b):
            out = [slice(None)] * dim
            out[ax] = slice(a, b)
            return tuple(out)

        for c, g in enumerate(self.component_ids):
            axc = _axis_of_component(dim, c)
            for axd in range(dim):
                if axd == axc:
                    # flux points = the n cell centres along the normal axis; faces -1 and n-1 are walls
                    shape = list(g.shape)
                    shape[axc] = n
                    fid = nflux + np.arange(int(np.prod(shape)), dtype=np.int64).reshape(shape)
                    nflux += fid.size
                    lo_f, lo_u = fid[sl(axc, 1, n)], g                        # cell i >= 1: lower face i - 1
                    hi_f, hi_u = fid[sl(axc, 0, n - 1)], g                    # cell i <= n-2: upper face i
                    for f_, u_, w_avg, w_diff in ((lo_f, lo_u, 0.5, -1.0), (hi_f, hi_u, 0.5, 1.0)):
                        rows_a.append(f_.ravel()); cols_a.append(u_.ravel()); vals_a.append(np.full(u_.size, 0.5))
                        rows_m.append(f_.ravel()); cols_m.append(u_.ravel()); vals_m.append(np.full(u_.size, w_avg))
                        rows_j.append(f_.ravel()); cols_j.append(u_.ravel()); vals_j.append(np.full(u_.size, w_diff))
                    # face m gets (F_{m+1} - F_m) h^(d-1)
                    rows_d += [g.ravel(), g.ravel()]
                    cols_d += [fid[sl(axc, 1, n)].ravel(), fid[sl(axc, 0, n - 1)].ravel()]
                    vals_d += [np.full(g.size, hface), np.full(g.size, -hface)]
                else:
                    # flux points between cells j-1 | j along axd, j = 1 .. n-1 (walls carry no flux)
                    shape = list(g.shape)
                    shape[axd] = n - 1
                    fid = nflux + np.arange(int(np.prod(shape)), dtype=np.int64).reshape(shape)
                    nflux += fid.size
                    u_lo, u_hi = g[sl(axd, 0, n - 1)], g[sl(axd, 1, n)]
                    for u_, w_avg, w_diff in ((u_lo, 0.5, -1.0), (u_hi, 0.5, 1.0)):
                        rows_m.append(fid.ravel()); cols_m.append(u_.ravel()); vals_m.append(np.full(u_.size, w_avg))
                        rows_j.append(fid.ravel()); cols_j.append(u_.ravel()); vals_j.append(np.full(u_.size, w_diff))
                    # advecting component e (normal to axd) at face j-1, mean over the cells m, m+1 along axc
                    e = next(k for k in range(dim) if _axis_of_component(dim, k) == axd)
                    ge = self.component_ids[e]
                    for part in (ge[sl(axc, 0, n - 1)], ge[sl(axc, 1, n)]):
                        rows_a.append(fid.ravel()); cols_a.append(part.ravel()); vals_a.append(np.full(part.size, 0.5))
                    # dof (m, j) gets (F_{j+1} - F_j) h^(d-1); F_0 = F_n = 0
                    rows_d += [g[sl(axd, 0, n - 1)].ravel(), g[sl(axd, 1, n)].ravel()]
                    cols_d += [fid.ravel(), fid.ravel()]
                    vals_d += [np.full(fid.size, hface), np.full(fid.size, -hface)]

        def csr(rows, cols, vals, shape):
            m = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=shape).tocsr()
            m.sort_indices()
            return m

        n0 = self.n_u // self.block_size
        ops = {"adv": csr(rows_a, cols_a, vals_a, (nflux, n0)), "avg": csr(rows_m, cols_m, vals_m, (nflux, n0)),
               "diff": csr(rows_j, cols_j, vals_j, (nflux, n0)), "div": csr(rows_d, cols_d, vals_d, (n0, nflux))}
        if self.block_size > 1:
            eye = sp.identity(self.block_size, format="csr")
            for key in ops:
                ops[key] = sp.kron(ops[key], eye, format="csr")
                ops[key].sort_indices()
        return ops

    def convection_reference(self, u):
        """conv(u) by direct loops over the grid (small plain cases; the check of `convection_operators`)."""
        if self.block_size != 1:
            raise ValueError("convection_reference: plain (not inflated) systems only")
        dim, n = self.dim, self.n
        hface = self.h ** (dim - 1)
        out = np.zeros(self.n_u)
        comps = self.component_ids
        axes = [_axis_of_component(dim, c) for c in range(dim)]

        def val(c, idx):                                   # u_c at face multi-index idx, 0 outside (walls)
            g = comps[c]
            if any(i < 0 or i >= g.shape[a] for a, i in enumerate(idx)):
                return 0.0
            return u[g[tuple(idx)]]

        for c, g in enumerate(comps):
            axc = axes[c]
            for idx in np.ndindex(*g.shape):
                total = 0.0
                for axd in range(dim):
                    for side in (1, -1):                   # upper / lower face of the control volume along axd
                        nb = list(idx)
                        nb[axd] += side
                        lo, hi = (idx, nb) if side == 1 else (nb, idx)
                        u_lo, u_hi = val(c, lo), val(c, hi)
                        if axd == axc:
                            adv = 0.5 * (u_lo + u_hi)
                        else:
                            e = axes.index(axd)
                            j = idx[axd] + (1 if side == 1 else 0)        # flux point between cells j-1 | j
                            if j <= 0 or j >= n:
                                continue                   # wall: no flux
                            ie = [0] * dim
                            for a in range(dim):
                                ie[a] = idx[a]
                            ie[axd] = j - 1
                            i2 = list(ie)
                            ie[axc], i2[axc] = idx[axc], idx[axc] + 1
                            adv = 0.5 * (val(e, ie) + val(e, i2))
                        flux = adv * 0.5 * (u_lo + u_hi) - 0.5 * abs(adv) * (u_hi - u_lo)
                        total += side * flux
                out[g[idx]] = -hface * total
        return out

    def auxiliary_space(self):
        """The auxiliary space of the reference's ``MypreA`` restated on the grid
        (templates/NavierStokesSIMPLE_iterative.py:150-157,208-357): one P1-like *nodal* scalar space
        per velocity component on the (n+1)^d grid vertices with homogeneous Dirichlet data on the walls
        (``fesh1_c``), the component's ``nu``-scaled 5-/7-point Laplacian on it (``aH1_c``, :322-351) and
        the ``transform`` that carries nodal vector fields to the face unknowns (:291; there a facet-wise
        L2 projection ``einv @ amixed``, here the average of the component over the 2^(d-1) vertices of
        the face).  For an inflated system every nodal unknown carries the `block_size` copies of its
        site (operator ``L_c (x) S``, transform ``T (x) I``).

        Returns dict(transform = CSR n_u x sum_c n_c, laplacians = [CSR per component],
        ranges = [range per component in the stacked auxiliary vector])."""
        dim, n, b = self.dim, self.n, self.block_size
        m = n - 1                                           # interior vertices per direction
        nodes = np.arange(m ** dim, dtype=np.int64).reshape((m,) * dim)
        ca = self.nu * self.h ** (dim - 2)
        one = sp.identity(m, format="csr")
        lap1 = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m), format="csr")
        lap = None
        for ax in range(dim):
            term = None
            for k in range(dim):
                f = lap1 if k == ax else one
                term = f if term is None else sp.kron(term, f, format="csr")
            lap = term if lap is None else lap + term
        lap = (ca * lap).tocsr()
        blocks, lapl, ranges = [], [], []
        off = 0
        n_nodes = m ** dim
        for c, g in enumerate(self.component_ids):
            normal = _axis_of_component(dim, c)
            rows, cols = [], []
            tang = [ax for ax in range(dim) if ax != normal]
            for corner in np.ndindex(*(2,) * len(tang)):     # the 2^(d-1) vertices of a face
                sel_face = [slice(None)] * dim
                sel_node = [slice(None)] * dim
                ok = True
                for ax, hi in zip(tang, corner):
                    # cell j touches vertices j and j + 1, i.e. interior nodes j - 1 and j
                    if hi:
                        sel_face[ax], sel_node[ax] = slice(0, n - 1), slice(0, m)
                    else:
                        sel_face[ax], sel_node[ax] = slice(1, n), slice(0, m)
                if ok:
                    rows.append(g[tuple(sel_face)].ravel())
                    cols.append(nodes[tuple(sel_node)].ravel())
            r, cc = np.concatenate(rows), np.concatenate(cols)
            w = np.full(r.size, 1.0 / 2 ** (dim - 1))
            blocks.append(sp.csr_matrix((w, (r, cc)), shape=(self.n_u // b, n_nodes)))
            lapl.append(lap)
            ranges.append(range(off * b, (off + n_nodes) * b))
            off += n_nodes
        T = sp.hstack(blocks, format="csr")
        if b > 1:
            eye_b = sp.identity(b, format="csr")
            T = sp.kron(T, eye_b, format="csr")
            S = sp.csr_matrix(self.inflation_block)
            lapl = [sp.kron(L, S, format="csr") for L in lapl]
        T.sort_indices()
        for L in lapl:
            L.sort_indices()
        return {"transform": T, "laplacians": lapl, "ranges": ranges}

    def auxiliary_space_stacked(self):
        """The auxiliary space of `auxiliary_space()` as ONE operator in slab-major order -- what a row partition
        needs: the stacked nodal spaces of all components re-ordered so that the vertex plane is the slowest index
        ([plane k | component c | in-plane node]), the block-diagonal Laplacian ``L = diag(aH1_1, .., aH1_d)`` and the
        transform in that numbering.  ``transform @ V(L) @ transform.T`` with one V-cycle on L is the term
        ``transform @ preAh1 @ transform.T`` of the reference's MypreA (templates/NavierStokesSIMPLE_iterative.py:336-337,
        357,380,383) with the component V-cycles run as one (the components are not coupled: the hierarchy of L is the
        union of theirs).  `node_slab_offsets[k]`: first stacked nodal dof of vertex plane k (plane k lies between the
        cell slabs k and k + 1 and goes with slab k); length n + 1 like `velocity_slab_offsets`."""
        space = self.auxiliary_space()
        dim, b = self.dim, self.block_size
        m = self.n - 1
        ncomp, per_comp, per_plane = len(space["laplacians"]), m ** dim, m ** (dim - 1)
        old = np.arange(ncomp * per_comp, dtype=np.int64)
        c, rest = old // per_comp, old % per_comp
        k, inplane = rest // per_plane, rest % per_plane
        new_of_old = k * (ncomp * per_plane) + c * per_plane + inplane
        if b > 1:                                            # every nodal unknown carries the b copies of its site
            new_of_old = (new_of_old[:, None] * b + np.arange(b)[None, :]).ravel()
        ntot = new_of_old.size
        Q = sp.csr_matrix((np.ones(ntot), (np.arange(ntot), new_of_old)), shape=(ntot, ntot))   # column old -> new
        L = (Q.T @ sp.block_diag(space["laplacians"], format="csr") @ Q).tocsr()
        T = (space["transform"] @ Q).tocsr()
        L.sort_indices()
        T.sort_indices()
        planes = np.minimum(np.arange(self.n + 1), m)       # slab n - 1 owns no vertex plane
        return {"transform": T, "laplacian": L, "node_slab_offsets": planes * (ncomp * per_plane * b)}

    def condense(self, seed=0):
        """Static condensation of A: split the velocity dofs into an
        *interior* set I (a maximal independent set of A's graph, so A_ii is diagonal -- the role
        the element-interior dofs play in the reference) and the coupling set C, and return the
        operators the reference reads off a condensed BilinearForm
        (solvers/bramblepasciak_new.py:11-17,88): ``mat`` = Schur complement S = A_cc - A_ci A_ii^-1
        A_ic, ``inner_matrix`` = A_ii, ``inner_solve`` = A_ii^-1, ``harmonic_extension`` = E with
        E_ic = -A_ii^-1 A_ic, ``harmonic_extension_trans`` = E^T -- all embedded in n_u x n_u, so
        that (I - E^T)(S + A_ii)(I - E) = A."""
        from hipla import coloring
        n = self.n_u
        singletons = np.arange(n, dtype=np.int32)[None, :]
        colors = coloring.color_blocks(coloring.block_graph(self.A, singletons), seed)
        interior = colors == 0
        I = np.nonzero(interior)[0]
        Cc = np.nonzero(~interior)[0]
        A = self.A.tocsr()
        a_ii = A[I][:, I]
        assert abs(a_ii - sp.diags(a_ii.diagonal())).max() == 0.0      # independent set: diagonal block
        dii = a_ii.diagonal()
        a_ic, a_ci, a_cc = A[I][:, Cc], A[Cc][:, I], A[Cc][:, Cc]
        schur = (a_cc - a_ci @ sp.diags(1.0 / dii) @ a_ic).tocoo()
        e_ic = (-(sp.diags(1.0 / dii) @ a_ic)).tocoo()

        def embed(m, rows, cols):
            return sp.csr_matrix((m.data, (rows[m.row], cols[m.col])), shape=(n, n))

        S = embed(schur, Cc, Cc)
        E = embed(e_ic, I, Cc)
        inner = sp.csr_matrix((dii, (I, I)), shape=(n, n))
        inner_solve = sp.csr_matrix((1.0 / dii, (I, I)), shape=(n, n))
        for m in (S, E, inner, inner_solve):
            m.sort_indices()
        return {"mat": S, "inner_matrix": inner, "inner_solve": inner_solve, "harmonic_extension": E,
                "harmonic_extension_trans": E.T.tocsr(), "interior": interior}

    def inflate(self, bs, seed=1):
        """'HDG-like' stress variant: Kronecker-inflate A with a seeded SPD bs x bs
        block (about 7*bs non-zeros per row, cf. the reference's ~84 at order 2 in 3-D)
        and B with a seeded 1 x bs row."""
        rng = np.random.default_rng(seed)
        q = rng.standard_normal((bs, bs))
        s_blk = np.eye(bs) + 0.25 * (q @ q.T) / bs
        r_blk = (1.0 + 0.5 * rng.random((1, bs))) / bs
        A = sp.kron(self.A, sp.csr_matrix(s_blk), format="csr")
        B = sp.kron(self.B, sp.csr_matrix(r_blk), format="csr")
        A.sort_indices()
        B.sort_indices()
        out = StokesSystem(self.dim, self.n, self.nu, self.h, A, B, self.mass.copy(),
                           self.velocity_slab_offsets * bs, self.pressure_slab_offsets.copy(),
                           self.component_ids, block_size=self.block_size * bs)
        out.inflation_block = s_blk if self.block_size == 1 else np.kron(self.inflation_block, s_blk)
        return out


def mac_stokes(dim, n, nu=0.01):
    """Assemble the MAC Stokes system on the unit square (dim=2) / cube (dim=3)."""
    if n < 2:
        raise ValueError("need n >= 2")
    h = 1.0 / n
    comps, S, n_u = _component_ids(dim, n)
    n_p = n ** dim
    ca = nu * h ** (dim - 2)
    cb = h ** (dim - 1)

    rows, cols, vals = [], [], []
    for c, g in enumerate(comps):
        normal_ax = _axis_of_component(dim, c)
        diag = np.full(g.shape, 2.0 * dim)
        for ax in range(dim):
            lo, hi = _shift_slices(dim, ax)
            rows += [g[lo].ravel(), g[hi].ravel()]
            cols += [g[hi].ravel(), g[lo].ravel()]
            off = np.full(g[lo].size, -ca)
            vals += [off, off]
            if ax != normal_ax:      # tangential wall: ghost reflection u_ghost = -u
                first = [slice(None)] * dim
                last = [slice(None)] * dim
                first[ax] = 0
                last[ax] = -1
                diag[tuple(first)] += 1.0
                diag[tuple(last)] += 1.0
        rows.append(g.ravel())
        cols.append(g.ravel())
        vals.append(ca * diag.ravel())
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows).astype(np.int32),
                                               np.concatenate(cols).astype(np.int32))),
                      shape=(n_u, n_u)).tocsr()
    A.sort_indices()

    pid = np.arange(n_p, dtype=np.int64).reshape((n,) * dim)
    rows, cols, vals = [], [], []
    for c, g in enumerate(comps):
        ax = _axis_of_component(dim, c)
        lo, hi = _shift_slices(dim, ax)
        rows += [pid[lo].ravel(), pid[hi].ravel()]          # face m: + for cell m, - for cell m+1
        cols += [g.ravel(), g.ravel()]
        vals += [np.full(g.size, cb), np.full(g.size, -cb)]
    B = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows).astype(np.int32),
                                               np.concatenate(cols).astype(np.int32))),
                      shape=(n_p, n_u)).tocsr()
    B.sort_indices()

    vel_off = np.concatenate([S, [n_u]]).astype(np.int64)
    prs_off = (np.arange(n + 1, dtype=np.int64) * n ** (dim - 1))
    return StokesSystem(dim, n, float(nu), h, A, B, np.full(n_p, h ** dim), vel_off, prs_off, comps)


def diffusion_2d(n=64, dt=1e-3):
    """cfg1 plumbing matrix: 5-point ``M + dt*K`` on an n x n interior grid with
    homogeneous Dirichlet walls (4 096 rows, 20 224 nnz at n=64;
    stands in for the heat.py operator, heat.py:58-61)."""
    h = 1.0 / (n + 1)
    t = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    k = (sp.kron(sp.identity(n), t) + sp.kron(t, sp.identity(n))) / (h * h)
    m = sp.identity(n * n) + dt * k
    m = m.tocsr()
    m.sort_indices()
    return m
