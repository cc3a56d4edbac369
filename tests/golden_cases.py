"""Shared by tests/golden/make_golden.py and the parity tests: turns the *inputs* recorded in a
golden fixture (generator parameters + seed) back into the assembled host-side system, and into
operands for the CPU oracle.  Host-only numpy/scipy; no engine is touched here.

Fixture parameters: ``dim, n, nu, seed`` (MAC Stokes system), ``inflate`` (Kronecker inflation
to the reference's ~25 / ~84 non-zeros per row), ``condense`` (statically
condensed form: solvers/bramblepasciak_new.py:8-21,84-109) and ``pre``:

* ``jacobi`` -- point Jacobi (`Preconditioner(.., 'local')`, run.py:62);
* ``bjac``   -- block Jacobi over blocks of 3 consecutive dofs of one component;
* ``facet``  -- block Jacobi over the dofs that sit on one mesh facet: all `inflate` copies of
  one face unknown (templates/NavierStokesSIMPLE_iterative.py:360-362; bs = 5 / 12 at order 2).

With ``condense`` the preconditioner acts on the coupling dofs only (it is a preconditioner of
the Schur complement `blfA.mat`) and is zero on the interior ones."""

import numpy as np

from staggered_grid import mac_stokes


def _get(d, key, default):
    return d[key] if key in d else default


class Case:
    def __init__(self, d):
        self.dim, self.n, self.nu = int(d["dim"]), int(d["n"]), float(d["nu"])
        self.seed, self.pre = int(d["seed"]), str(d["pre"])
        self.inflate = int(_get(d, "inflate", 1))
        self.condense = bool(int(_get(d, "condense", 0)))
        s = mac_stokes(self.dim, self.n, self.nu)
        if self.inflate > 1:
            s = s.inflate(self.inflate)
        self.system = s
        self.f, self.g = s.rhs(self.seed)
        self.parts = s.condense() if self.condense else None
        self.pre_matrix = self.parts["mat"] if self.condense else s.A      # what preA approximates
        if self.pre == "jacobi":
            self.blocks = None
        elif self.pre == "bjac":
            self.blocks = s.line_blocks(3)
        elif self.pre == "facet":
            self.blocks = s.line_blocks(1)
        else:
            raise ValueError(self.pre)
        if self.condense and self.blocks is not None:       # blocks of coupling dofs only (S is zero elsewhere)
            idx = self.blocks.copy()
            idx[(idx >= 0) & self.parts["interior"][np.maximum(idx, 0)]] = -1
            idx = -np.sort(-idx, axis=0)                    # padding last
            self.blocks = np.ascontiguousarray(idx[:, (idx >= 0).any(axis=0)])

    @property
    def rhs(self):
        return np.concatenate([self.f, self.g])

    def jacobi_diagonal(self):
        """Inverse diagonal of the point-Jacobi preconditioner (zero on interior dofs when condensed)."""
        d = self.pre_matrix.diagonal()
        if self.condense:
            return np.where(self.parts["interior"], 0.0, 1.0 / np.where(d != 0.0, d, 1.0))
        return 1.0 / d

    def condensed_operators(self):
        p = self.parts
        return {k: p[k] for k in ("harmonic_extension", "harmonic_extension_trans", "inner_solve", "inner_matrix")}

    # ---- operands of oracle/krylov_ref.py ----------------------------------------------------
    def oracle_operands(self, kr):
        """(A, B, pre_a, pre_s, condensed-or-None) for kr.bpcg_v1 / bpcg_v2 / minres."""
        if self.blocks is None:
            dinv = self.jacobi_diagonal()
            pa = lambda x: dinv * x
        else:
            pa = kr.block_jacobi(self.pre_matrix, self.blocks)
        cond = self.condensed_operators() if self.condense else None
        return self.pre_matrix, self.system.B, pa, kr.diag_inverse(self.system.mass), cond
