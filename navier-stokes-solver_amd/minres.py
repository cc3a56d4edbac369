"""Preconditioned MINRES for the Stokes saddle-point system -- drop-in for the
reference's ``minres.MinRes`` (minres.py:12-149; same signature, same return value
``(u, errors)``, same two stopping rules, same warm-start semantics).

Two execution paths, both on the GPU:

* fused -- when ``mat`` is ``BlockMatrix([[A, B.T], [B, None]])`` of native
  ``SparseMatrix`` blocks and ``pre`` is ``BlockMatrix([[preA, None], [None, preS]])`` of
  native Jacobi / block-Jacobi preconditioners (the operands ``run.py:45-46`` builds), the
  whole iteration runs behind the C ABI (``nss_minres_*``): 3 SpMVs with fused vector
  updates and dot partials, scalars kept on the device, no host sync per iteration.
* protocol -- any other ``BaseMatrix`` operands (user subclasses included) are driven
  through the hipla operator protocol, one HIP kernel per vector statement.

Recurrences follow Kolmbauer's formulation used by the reference (minres.py:9)."""

from math import sqrt

from hipla import InnerProduct
from hipla.fused import MinresLoop

__all__ = ["MinRes"]


def _report(k, res, err0, printrates):
    if printrates:
        print("\rit =", k, "rel err =", res / err0, "abs err =", res, " " * 20, end="")


def MinRes(mat, rhs, pre=None, sol=None, maxsteps=100, printrates=True, initialize=True, tol=1e-7):
    """Minimal-residual iteration on ``mat * u = rhs`` with SPD preconditioner ``pre``.

    Parameters mirror the reference (minres.py:12): ``sol`` is the start vector when
    ``initialize=False`` and is overwritten with the solution (aliasing the caller's
    storage); ``tol`` is used twice -- the loop runs while the *absolute* residual
    estimate exceeds ``tol`` (minres.py:96) and breaks when the *relative* one drops
    below ``tol`` (minres.py:126); leaving through the absolute guard prints the
    "did not converge" warning exactly as the reference's while/else does.

    Returns ``(u, errors)`` with ``errors[0] == 1.0`` and one entry per iteration."""
    u = sol if sol else rhs.CreateVector()
    v_prev, v_cur, v_next = rhs.CreateVector(), rhs.CreateVector(), rhs.CreateVector()
    w_prev, w_cur, w_next = rhs.CreateVector(), rhs.CreateVector(), rhs.CreateVector()
    z_cur, z_next = rhs.CreateVector(), rhs.CreateVector()
    kz = rhs.CreateVector()

    if initialize:                                   # minres.py:62-64
        u[:] = 0.0
        v_cur.data = rhs
    else:                                            # minres.py:66
        v_cur.data = rhs - mat * u
    z_cur.data = pre * v_cur if pre else v_cur       # minres.py:68

    gamma = sqrt(InnerProduct(z_cur, v_cur))         # minres.py:71
    z_cur.data = 1 / gamma * z_cur
    v_cur.data = 1 / gamma * v_cur

    res = res_prev = err0 = gamma
    if printrates:
        print("\rit = ", 0, " err = ", res, " " * 20, end="")
    eta_prev = gamma
    c_prev = c_cur = 1
    s_prev = s_cur = s_next = 0
    v_prev[:] = 0.0
    w_prev[:] = 0.0
    w_cur[:] = 0.0

    errors = [1.0]                                   # minres.py:95
    k = 1
    hit_relative_tol = False
    fused_loop = None
    if pre and maxsteps >= 1 and res > tol:
        # ring layout expected by nss_minres_iterate at k = 1: old = [0], current = [1], new = [2]
        fused_loop = MinresLoop.try_create(mat, pre, u, (v_prev, v_cur, v_next), (w_prev, w_cur, w_next),
                                           (z_next, z_cur), kz)
    if fused_loop is not None:
        errors, hit_relative_tol = fused_loop.run(gamma, tol, maxsteps)
        if printrates:
            for i, e in enumerate(errors[1:], 1):
                _report(i, e * err0, err0, True)
        maxsteps = 0                                 # the protocol loop below is skipped
    while k < maxsteps + 1 and res > tol:            # absolute guard, minres.py:96
        kz.data = mat * z_cur                        # :97
        delta = InnerProduct(kz, z_cur)              # :98
        v_next.data = kz - delta * v_cur - gamma * v_prev   # :99
        z_next.data = pre * v_next if pre else v_next        # :101
        gamma_next = sqrt(InnerProduct(z_next, v_next))      # :103
        z_next *= 1 / gamma_next
        v_next *= 1 / gamma_next

        a0 = c_cur * delta - c_prev * s_cur * gamma  # Givens recurrences, :107-113
        a1 = sqrt(a0 * a0 + gamma_next * gamma_next)
        a2 = s_cur * delta + c_prev * c_cur * gamma
        a3 = s_prev * gamma
        c_next = a0 / a1
        s_next = gamma_next / a1

        w_next.data = z_cur - a3 * w_prev - a2 * w_cur       # :115
        w_next.data = 1 / a1 * w_next                         # :116
        u.data += c_next * eta_prev * w_next                  # :118
        eta = -s_next * eta_prev

        res = abs(s_next) * res_prev                 # residual *estimate*, :122
        _report(k, res, err0, printrates)
        errors.append(res / err0)
        if res < tol * err0:                         # relative break, :126
            hit_relative_tol = True
            break
        k += 1

        v_prev, v_cur, v_next = v_cur, v_next, v_prev        # rotate by renaming, :131-133
        w_prev, w_cur, w_next = w_cur, w_next, w_prev
        z_cur, z_next = z_next, z_cur
        eta_prev = eta
        s_prev, s_cur = s_cur, s_next
        c_prev, c_cur = c_cur, c_next
        gamma = gamma_next
        res_prev = res
    if not hit_relative_tol:
        print("\nWarning: MinRes did not converge to TOL")
    print("")
    return (u, errors)
