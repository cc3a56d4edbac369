"""Bramble-Pasciak CG, recurrence-optimised form (3 SpMV per iteration) -- drop-in for
the reference's ``solvers.bramblepasciak_new.BramblePasciakCG``
(solvers/bramblepasciak_new.py:24-253) and ``harmonic_extension`` (:8-21).  This is the
solver the SIMPLE / Stokes drivers call (templates/NavierStokesSIMPLE_iterative.py:397).

Execution paths (both on the GPU): the fused device-resident loop behind the C ABI
(``nss_bpcg2_*``: the three SpMVs carry the vector recurrences and dot partials in their
epilogues, alpha / beta / stop test stay on the device) when every operand is native;
otherwise the operator protocol, statement for statement.

Scope: the plain branch of ``harmonic_extension`` (:20) is the hot path.  The
static-condensation branch (``blfA.condense``: :11-18, ``myAmatrix`` :84-103) is kept
behind the same interface through the protocol but needs element-block operators from FE
assembly."""

from math import sqrt

from hipla import BaseMatrix, BlockVector, IdentityMatrix, InnerProduct
from hipla.fused import Bpcg2Loop
from hipla.la import EigenValues_Preconditioner
from hipla.ngstd import Timer

__all__ = ["harmonic_extension", "BramblePasciakCG", "BpcgSession"]


def harmonic_extension(f, blfA, inverse, result=None):
    """``result = inverse * f`` -- or, for a statically condensed form,
    ``(I+H) inverse (I+H^T) f + A_ii^-1 (I+H^T) f`` (solvers/bramblepasciak_new.py:11-18)."""
    if result is None:
        result = inverse.CreateColVector()
    if blfA.condense:
        lifted = f.Copy()
        lifted.data += blfA.harmonic_extension_trans * lifted
        result.data = inverse * lifted
        result.data += blfA.harmonic_extension * result
        result.data += blfA.inner_solve * lifted
    else:
        result.data = inverse * f
    return result


class _CondensedA(BaseMatrix):
    """``(I - H^T)(S + A_ii)(I - H)`` (the reference's local ``myAmatrix``, :84-103)."""

    def __init__(self, blfA):
        super().__init__()
        self.blfA = blfA
        self.mat = ((IdentityMatrix() - blfA.harmonic_extension_trans)
                    @ (blfA.mat + blfA.inner_matrix)
                    @ (IdentityMatrix() - blfA.harmonic_extension))

    def Mult(self, x, y):
        y.data = self.mat * x

    def Height(self):
        return self.blfA.mat.height

    def Width(self):
        return self.blfA.mat.width

    def CreateColVector(self):
        return self.blfA.mat.CreateColVector()

    CreateVector = CreateColVector


def _explicit_condensed_matrix(blfA):
    """``(I - H^T)(S + A_ii)(I - H)`` as one CSR matrix (two sparse products on the device,
    ``nss_csr_spgemm``), so that the fused loop multiplies with it in a single SpMV instead of the
    five operator applies of ``myAmatrix`` (:84-103).  None when the operands are not native."""
    import scipy.sparse as sp
    from hipla.matrix import SparseMatrix
    parts = [getattr(blfA, name, None) for name in ("mat", "inner_matrix", "harmonic_extension",
                                                    "harmonic_extension_trans", "inner_solve")]
    if not all(isinstance(m, SparseMatrix) for m in parts):
        return None
    S, Aii, H, HT, _ = parts
    eng = S.engine
    if not hasattr(eng, "csr_spgemm") or len({m.height for m in parts} | {m.width for m in parts}) != 1:
        return None
    n = S.height
    eye = sp.identity(n, format="csr")
    left = SparseMatrix.from_scipy((eye - HT.to_scipy()).tocsr(), engine=eng)
    mid = SparseMatrix.from_scipy((S.to_scipy() + Aii.to_scipy()).tocsr(), engine=eng)
    right = SparseMatrix.from_scipy((eye - H.to_scipy()).tocsr(), engine=eng)
    inner = eng.csr_spgemm(left.handle, mid.handle)
    return SparseMatrix.from_handle(eng.csr_spgemm(inner, right.handle), eng)


class BpcgSession:
    """Everything the reference computes before its loop (solvers/bramblepasciak_new.py:105-198):
    scale factor, transformed right-hand side, initial defect ``d``, preconditioned residual
    ``w``, search direction ``s`` and ``wdn = <w,d>``.  ``fused`` is the device-resident loop
    when the operands are native, else ``None``."""

    def __init__(self, blfA, blfB, matC, f, g, preA_unscaled, preM, sol=None, initialize=True, k=None,
                 inner=InnerProduct, workspace=None):
        """``inner`` is the inner product (a row-partitioned run passes the all-reducing one);
        ``workspace`` may pre-allocate ``t1``, ``t4`` and ``s1`` (the SpMV operands of the loop)
        in halo-extended buffers -- and, for the compact partitioned plan, ``s0``, ``w0``, ``w1``, ``t3``
        (vectors whose ghost copies sit behind their owned entries)."""
        self.blfA = blfA
        self.inner = inner
        workspace = workspace or {}
        matA = self.matA = _CondensedA(blfA) if blfA.condense else blfA.mat
        matB = self.matB = blfB.mat
        self.preM = preM

        self.timer_prep = Timer("BPCG-Preparation")
        self.timer_prep.Start()
        if k is None:
            timer_prepev = Timer("BPCG-Preparation-EV")
            timer_prepev.Start()
            lams = EigenValues_Preconditioner(mat=matA, pre=preA_unscaled, tol=1e-3, inner=inner)
            timer_prepev.Stop()
            k = 1. / min(lams) + 1e-3                   # :118
            print("condition", max(lams) / min(lams))
        self.k = k
        preA = self.preA = k * preA_unscaled            # :122

        # ---- transformed right-hand side (:124-135) ----------------------------------------
        f_hat = matA.CreateColVector()
        t0 = f.CreateVector()
        harmonic_extension(f, blfA, preA, result=t0)
        f_hat.data = matA * t0 - f
        g_hat = matB.CreateColVector()
        g_hat.data = matB * t0 - g
        rhs = BlockVector([f_hat, g_hat])

        u = self.u = sol if sol else rhs.CreateVector()
        if initialize:
            u[:] = 0.0
        d, w, v = rhs.CreateVector(), rhs.CreateVector(), rhs.CreateVector()
        z, z_old, s = rhs.CreateVector(), rhs.CreateVector(), rhs.CreateVector()
        if "s1" in workspace or "s0" in workspace:
            s = BlockVector([workspace.get("s0", s[0]), workspace.get("s1", s[1])])
        if "w0" in workspace or "w1" in workspace:
            w = BlockVector([workspace.get("w0", w[0]), workspace.get("w1", w[1])])
        self.d, self.w, self.v, self.z, self.z_old, self.s = d, w, v, z, z_old, s

        t0 = self.t0 = blfA.mat.CreateColVector()
        t1 = self.t1 = workspace["t1"] if "t1" in workspace else blfA.mat.CreateColVector()
        t2 = self.t2 = blfA.mat.CreateColVector()
        t3 = self.t3 = workspace["t3"] if "t3" in workspace else matB.CreateColVector()
        t4 = self.t4 = workspace["t4"] if "t4" in workspace else blfA.mat.CreateColVector()
        self.As0 = blfA.mat.CreateColVector()
        self.BTs1 = matB.CreateRowVector()

        # ---- initial defect d = rhs^ - K^ u and preconditioned residual w (:160-183) ----------
        t0.data = matA * u[0] + matB.T * u[1]
        harmonic_extension(t0, blfA, preA, t1)
        t2.data = matA * t1
        t4.data = t1 - u[0]
        t3.data = matB * t4
        d[0].data = rhs[0] - (t2 - t0)
        d[1].data = rhs[1] - t3

        pr = rhs.CreateVector()
        harmonic_extension(f, blfA, preA, pr[0])
        t5 = matB.CreateColVector()
        t5.data = matB * pr[0] - g
        pr[1].data = preM * t5
        w[0].data = pr[0] - t1
        w[1].data = pr[1] - preM * t3

        self.wdn = inner(w, d)                           # :185
        self.err0 = sqrt(abs(self.wdn))
        s.data = w                                       # :189

        self.matBT = matB.CreateTranspose()              # explicit B^T CSR (:198), cached on the matrix
        self.fused = None
        vecs = dict(u0=u[0], u1=u[1], d0=d[0], d1=d[1], w0=w[0], w1=w[1], s0=s[0], s1=s[1], z0=z[0],
                    q=self.As0, t0=t0, t1=t1, t2=t2, t3=t3, t4=t4)
        if matC is None and not blfA.condense:
            self.fused = Bpcg2Loop.try_create(matA, matB, self.matBT, preA_unscaled, k, preM, vecs)
        elif matC is None:
            explicit = _explicit_condensed_matrix(blfA)
            if explicit is not None:
                self.fused = Bpcg2Loop.try_create(
                    explicit, matB, self.matBT, preA_unscaled, k, preM, vecs,
                    condensed=dict(HT=blfA.harmonic_extension_trans, H=blfA.harmonic_extension,
                                   inner=blfA.inner_solve))

    def first_direction(self):
        """A s0 and z0 of iteration 0 (:202-203); the fused loop starts from these."""
        self.As0.data = self.matA * self.s[0]
        self.z[0].data = self.As0

    def protocol_loop(self, tol, maxsteps, printrates, rel_err):
        """The loop (:200-249) through the operator protocol: one kernel per statement."""
        blfA, matA, matB, matBT, preA, preM = self.blfA, self.matA, self.matB, self.matBT, self.preA, self.preM
        u, d, w, v, z, z_old, s = self.u, self.d, self.w, self.v, self.z, self.z_old, self.s
        t0, t1, t2, t3, t4, As0, BTs1 = self.t0, self.t1, self.t2, self.t3, self.t4, self.As0, self.BTs1
        wdn, err0, InnerProduct = self.wdn, self.err0, self.inner
        alpha = beta = 0.0
        converged = False
        for it in range(maxsteps):
            if it == 0:                                  # :201-205
                self.first_direction()
            else:
                As0.data = beta * As0 + z_old[0] - alpha * t2
            BTs1.data = matBT * s[1]                     # :206
            t0.data = As0 + BTs1
            harmonic_extension(f=t0, blfA=blfA, inverse=preA, result=t1)   # :209
            t2.data = matA * t1                          # :210
            t4.data = t1 - s[0]
            t3.data = matB * t4                          # :213
            z_old[0].data = z[0]
            v[0].data = t2 - t0                          # K^ s  (:218-219)
            v[1].data = t3

            wd = wdn
            alpha = wd / InnerProduct(s, v)              # :222-226
            u.data += alpha * s
            d.data += (-alpha) * v
            w[0].data = w[0] + (-alpha) * t1             # :232-233
            w[1].data = w[1] + (-alpha) * preM * t3
            wdn = InnerProduct(w, d)                     # :235
            beta = wdn / wd
            z[0].data -= alpha * t2                      # :238
            s *= beta                                    # :240-241
            s.data += w

            err = sqrt(abs(wd))
            if printrates:
                print("it = ", it, " err = ", err, " " * 20)
            if err < tol * (err0 if rel_err else 1):
                converged = True
                break
        return it, converged


def BramblePasciakCG(blfA, blfB, matC, f, g, preA_unscaled, preM, sol=None, tol=1e-6, maxsteps=100,
                     printrates=True, initialize=True, rel_err=True):
    """Bramble-Pasciak CG for ``[[A, B^T], [B, 0]]`` with ``A~ = k * preA_unscaled``.

    Same contract as the reference (solvers/bramblepasciak_new.py:24-253):
    ``blfA`` / ``blfB`` are BilinearForm-like (``.mat``, ``.condense``); ``sol`` (2-component
    BlockVector) is the start vector when ``initialize=False`` and receives the solution;
    stop when ``sqrt(|<w,d>|) < tol * err0`` (``err0`` dropped if ``rel_err=False``), tested at
    the end of each iteration with the value from its start (:243-247).

    Returns ``(it, seconds)`` -- last iteration index and wall time of the iteration loop
    only (:195-196,251-253); returns the bare solution vector when the initial residual
    functional is exactly zero (:191-192)."""
    ses = BpcgSession(blfA, blfB, matC, f, g, preA_unscaled, preM, sol=sol, initialize=initialize)
    print("err0", ses.err0)
    if ses.wdn == 0:
        return ses.u

    ses.timer_prep.Stop()
    timer_its = Timer("BPCG-Iterations")
    timer_its.Start()
    if ses.fused is not None:
        ses.first_direction()
        it, history, converged = ses.fused.run(ses.wdn, ses.err0, tol, rel_err, maxsteps)
        timer_its.Stop()
        if printrates:
            for i, err in enumerate(history):
                print("it = ", i, " err = ", float(err), " " * 20)
    else:
        it, converged = ses.protocol_loop(tol, maxsteps, printrates, rel_err)
        timer_its.Stop()
    if not converged:
        print("Warning: BPCG did not converge to TOL")
    print("\n")
    return (it, timer_its.time)
