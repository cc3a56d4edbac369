"""Bramble-Pasciak CG, textbook form -- drop-in for the reference's
``bramble_pasciak_cg.bramble_pasciak_cg`` (bramble_pasciak_cg.py:65-148) and its two
operator wrappers ``ScaledPreconditioner`` (:9-36) and ``MatrixAB`` (:39-62).

Solves ``[[A, B^T], [B, C]] [u; p] = [f; g]`` (C = None -> zero block) by CG on the
Bramble-Pasciak transformed, SPD-in-a-special-inner-product system with
``A~ = k * pre_a``, ``k = 1/lambda_min(pre_a A) + 1e-3`` (:70-71).

Execution paths (both on the GPU): the fused device-resident loop behind the C ABI
(``nss_bpcg1_*``) when every operand is native, otherwise the operator protocol below,
statement for statement -- 6 SpMV, 1 pre_a, 1 pre_s, 2 inner products per iteration."""

from math import sqrt

from hipla import BaseMatrix, BlockMatrix, BlockVector, IdentityMatrix, InnerProduct, Vector
from hipla import fused
from hipla.fused import Bpcg1Loop
from hipla.la import EigenValues_Preconditioner
from hipla.ngstd import Timer

__all__ = ["ScaledPreconditioner", "MatrixAB", "bramble_pasciak_cg"]


class ScaledPreconditioner(BaseMatrix):
    """``y := factor * s * pre * x``.

    Reference quirk kept on purpose (bramble_pasciak_cg.py:16-21): ``MultAdd`` and
    ``MultTransAdd`` *overwrite* ``y`` instead of accumulating.  That is only correct
    because this operator is the sole block of its block row (:79-80)."""

    def __init__(self, factor, matrix, pre):
        super().__init__()
        self.factor, self.matrix, self.pre = factor, matrix, pre

    def MultAdd(self, s, x, y):
        y.data = self.factor * s * self.pre * x

    def MultTransAdd(self, s, x, y):          # pre symmetric: no transpose needed
        y.data = self.factor * s * self.pre * x

    def Height(self):
        return self.pre.height

    def Width(self):
        return self.pre.width

    def CreateColVector(self):
        return self.matrix.CreateColVector()

    CreateVector = CreateColVector

    def CreateRowVector(self):
        return self.matrix.CreateRowVector()


class MatrixAB(BaseMatrix):
    """``[A; B]`` applied to the velocity component: ``y0 += s A x0``, ``y1 += s B x0``
    (bramble_pasciak_cg.py:45-47).  ``Width`` reports ``a.width + b.height`` although only
    ``x[0]`` is read, and ``CreateColVector`` returns a *plain* vector (:52-59) -- both as
    in the reference; expressions evaluate into the (block) destination."""

    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b

    def MultAdd(self, s, x, y):
        y[0].data += s * self.a * x[0]
        y[1].data += s * self.b * x[0]

    def Height(self):
        return self.a.height + self.b.height

    def Width(self):
        return self.a.width + self.b.height

    def CreateColVector(self):
        return Vector(self.height)

    CreateVector = CreateColVector

    def CreateRowVector(self):
        return Vector(self.width)


def bramble_pasciak_cg(a_matrix, b_matrix, c_matrix, pre_a, pre_schur_complement,
                       upper_rhs, lower_rhs, solution=None,
                       tolerance=1e-12, max_steps=1000, print_rates=True):
    """Returns ``(solution, errors)``; ``errors[i] = err_i / err_0`` is appended *before*
    the stop test, so ``errors[0] == 1.0`` and ``len(errors) == iterations + 1``
    (bramble_pasciak_cg.py:115-121).  ``solution`` (a 2-component BlockVector) is the
    start vector and is updated in place; ``None`` starts from zero (:88-90)."""
    fused.plan_for_textbook_bpcg(a_matrix, pre_a)        # (launch plans before the first product: see there)
    ev_timer = Timer("eigenvalues")
    ev_timer.Start()
    ritz = EigenValues_Preconditioner(mat=a_matrix, pre=pre_a)
    k = 1 / min(ritz) + 1e-3
    ev_timer.Stop()
    print("scale factor: ", k)
    print("condition number: ", max(ritz) / min(ritz))

    n_u, n_p = a_matrix.width, b_matrix.height
    K = BlockMatrix([[a_matrix, b_matrix.T], [b_matrix, c_matrix]])
    Atilde = BlockMatrix([[ScaledPreconditioner(k, a_matrix, pre_a), None], [None, IdentityMatrix(n_p)]])
    Bfull = BlockMatrix([[IdentityMatrix(n_u), None], [b_matrix, -IdentityMatrix(n_p)]])
    AB = MatrixAB(a_matrix, b_matrix)
    Sfull = BlockMatrix([[IdentityMatrix(n_u), None], [None, pre_schur_complement]])

    rhs = BlockVector([upper_rhs, lower_rhs])
    if not solution:
        solution = rhs.CreateVector()
        solution[:] = 0

    r = rhs.CreateVector()         # residuum
    t1 = rhs.CreateVector()
    d = rhs.CreateVector()         # search direction ("full_preconditioned_residuum")
    ar = rhs.CreateVector()        # A~-preconditioned residual
    t2 = rhs.CreateVector()

    t2.data = rhs - K * solution                          # :98
    ar.data = Atilde * t2                                 # :99
    r.data = AB * ar - rhs + K * solution                 # :100-101
    t1.data = Sfull @ Bfull * ar                          # :102
    d.data = t1
    rho = InnerProduct(t1, r)                             # :105
    err0 = sqrt(abs(rho))
    errors = []

    fused_loop = Bpcg1Loop.try_create(a_matrix, b_matrix, c_matrix, pre_a, pre_schur_complement, k,
                                      dict(x=solution, r=r, d=d, a=ar, t1=t1, t2=t2))
    if fused_loop is None and hasattr(a_matrix, "plan"):      # row-partitioned operands (distributed.py)
        from distributed import Bpcg1DistLoop
        fused_loop = Bpcg1DistLoop.try_create(a_matrix, b_matrix, c_matrix, pre_a, pre_schur_complement, k,
                                              dict(x=solution, r=r, d=d, a=ar, t1=t1, t2=t2))
    if fused_loop is not None:
        errors, converged = fused_loop.run(rho, err0, tolerance, max_steps)
        if print_rates:
            for i, e in enumerate(errors):
                print("\rit =", i, "rel err =", e, "abs err =", e * err0, " " * 20, end="")
        if not converged:
            print("\nWarning: CG did not converge to TOL")
        print("")
        return (solution, errors)

    for iteration in range(max_steps):
        it_timer = Timer("Bramble Pasciak CG Iteration " + str(iteration))
        it_timer.Start()
        err = sqrt(abs(rho))
        if print_rates:
            print("\rit =", iteration, "rel err =", err / err0, "abs err =", err, " " * 20, end="")
        errors.append(err / err0)
        if err < tolerance * err0:
            it_timer.Stop()
            break
        rho_prev = rho

        t1.data = -K * d                                  # :125
        t2.data = -Atilde * t1                            # :126
        t1.data += AB * t2                                # :127
        alpha = rho_prev / InnerProduct(d, t1)            # :129-130
        solution.data += alpha * d                        # :131
        r.data += (-alpha) * t1                           # :132
        ar.data += (-alpha) * t2                          # :133
        t1.data = Sfull @ Bfull * ar                      # :135
        rho = InnerProduct(t1, r)                         # :137
        beta = rho / rho_prev
        d *= beta                                         # :140-141
        d.data += t1
        it_timer.Stop()
    else:
        print("\nWarning: CG did not converge to TOL")
    print("")
    return (solution, errors)
