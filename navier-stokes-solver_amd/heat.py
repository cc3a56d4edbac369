"""BASELINE.json config 1 ("heat.py 2D diffusion, 64x64 grid, CG"): the plumbing operations the
reference's heat.py performs on single (non-block) vectors -- SpMV (heat.py:96,110,116), AXPY
(:97,140-142), InnerProduct / Norm (:89,112,117) and the Gram-Schmidt of orthonormalization.py --
driven through the protocol on the 5-point ``M + dt*K`` matrix of `staggered_grid.diffusion_2d`.

The exponential integrator itself (H1 order-10 FE space, sparse direct inverse :72, implicit
Runge-Kutta in the 5-dimensional Krylov subspace :95-138) is out of scope:
dense 5x5 algebra and a direct solver are not the Krylov path."""

from math import sqrt

import numpy as np

import hipla
from hipla import InnerProduct, Norm
from orthonormalization import orthonormalize
from staggered_grid import diffusion_2d


def conjugate_gradients(mat, rhs, tol=1e-10, maxsteps=1000):
    """Textbook CG on ``mat * x = rhs`` with protocol operations only; returns (x, history) with
    history[i] = sqrt(<r_i, r_i>)."""
    x, r, p, q = (rhs.CreateVector() for _ in range(4))
    x[:] = 0.0
    r.data = rhs
    p.data = r
    rz = InnerProduct(r, r)
    history = [sqrt(rz)]
    for _ in range(maxsteps):
        q.data = mat * p
        alpha = rz / InnerProduct(p, q)
        x.data += alpha * p
        r.data -= alpha * q
        rz_new = InnerProduct(r, r)
        history.append(sqrt(rz_new))
        if history[-1] < tol * history[0]:
            break
        p.data = r + (rz_new / rz) * p
        rz = rz_new
    return x, history


def conjugate_gradients_fused(mat, rhs, tol=1e-10, maxsteps=1000):
    """The same CG through `hipla.CGSolver` (device-resident loop ``nss_cg_*`` on the GPU)."""
    solver = hipla.CGSolver(mat, pre=None, precision=tol, maxsteps=maxsteps)
    x = rhs.CreateVector()
    x.data = solver * rhs
    return x, solver.errors


def krylov_galerkin(mat, start, dimension=5):
    """Krylov basis {v, Mv, .., M^(d-1) v}, orthonormalised (heat.py:95-100), and its Galerkin
    matrix G[r, c] = <b_r, M b_c> (heat.py:109-118)."""
    basis = [start.Copy()]
    for _ in range(1, dimension):
        nxt = basis[-1].CreateVector()
        nxt.data = mat * basis[-1]
        basis.append(nxt)
    basis = orthonormalize(basis)
    galerkin = np.zeros((dimension, dimension))
    residual = start.CreateVector()
    for col in range(dimension):
        residual.data = mat * basis[col]
        for row in range(dimension):
            galerkin[row, col] = InnerProduct(basis[row], residual)
    return basis, galerkin


def solve(n=64, dt=1e-3, seed=3, tol=1e-10):
    """Config-1 run: returns (x, cg_history, galerkin_matrix) for the n x n diffusion matrix."""
    mat = hipla.SparseMatrix.from_scipy(diffusion_2d(n, dt))
    rhs = hipla.Vector.from_numpy(np.random.default_rng(seed).standard_normal(n * n))
    x, history = conjugate_gradients(mat, rhs, tol=tol)
    _, galerkin = krylov_galerkin(mat, rhs)
    return x, history, galerkin


if __name__ == "__main__":
    x, hist, gal = solve()
    print("CG iterations:", len(hist) - 1, " final residual:", hist[-1], " |x| =", Norm(x))
