"""The CPU oracle (oracle/krylov_ref.py) against the golden vectors produced by the
reference's own unmodified solver files (tests/golden/make_golden.py).

Tolerance contract: (i) history relative difference <= 1e-8
over the recorded stable window W; (ii) iteration count within the band of conftest.iteration_tolerance;
(iii) solution norm / true residual agree; quirk cases reproduce flags exactly."""

import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_path, iteration_tolerance
from golden_cases import Case
from oracle import krylov_ref as kr
from staggered_grid import diffusion_2d, mac_stokes

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "stokes*.npz")))


def _system(d):
    case = Case(d)
    _, _, pa, ps, _ = case.oracle_operands(kr)
    return case.system, case.f, case.g, pa, ps


def _close_history(h, ref, window, rtol=1e-8):
    w = min(int(window), len(h), len(ref))
    assert w > 10
    rel = np.abs(h[:w] - ref[:w]) / np.abs(ref[:w])
    assert rel.max() <= rtol, "history differs inside the stable window: %g" % rel.max()


def _close_iterations(it, ref, d=None):
    tol = iteration_tolerance(d) if d is not None else max(3, int(0.03 * int(ref) + 0.999))
    assert abs(int(it) - int(ref)) <= tol, (it, ref, tol)


def test_goldens_present():
    # {2-D n=12,24,48; 3-D n=6,10} x {jacobi, bjac} x {v1, v2, MINRES}  +  4 condensed (v2)
    # +  {2-D x5, 3-D x12 facet blocks} x 3  +  {Re 400, 1000} x 3
    assert len(CASES) == 30 + 4 + 6 + 6
    assert sum("_cond_" in c for c in CASES) == 4 and sum("_x12_" in c for c in CASES) == 3
    for q in ("quirk_minres_absolute_guard", "quirk_minres_warm_start", "quirk_bpcg2_zero_rhs",
              "quirk_bpcg2_warm_start", "quirk_bpcg2_abs_err", "quirk_bpcg1_warm_start_maxsteps",
              "cfg1_heat_plumbing"):
        assert os.path.exists(golden_path(q))


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_run(case):
    d = np.load(golden_path(case))
    c = Case(d)
    sysm, f, g = c.system, c.f, c.g
    oA, _, pa, ps, ocond = c.oracle_operands(kr)
    Ks = sysm.saddle_matrix()
    b = np.concatenate([f, g])
    solver = str(d["solver"])
    # the two stand-ins the reference ran over agreed (recorded by make_golden.py)
    if "standin_head_rel_diff" in d:
        assert float(d["standin_head_rel_diff"]) <= 1e-12 and float(d["standin_window_rel_diff"]) <= 1e-9
    if solver == "bpcg1":
        u, p, errors, conv = kr.bpcg_v1(sysm.A, sysm.B, pa, ps, f, g, float(d["k"]),
                                        tolerance=float(d["tol"]), max_steps=int(d["maxsteps"]))
        _close_history(errors, d["errors"], d["window"])
        _close_iterations(len(errors) - 1, d["iterations"], d)
        assert conv == (not bool(d["warned"]))
    elif solver == "bpcg2":
        it, u, p, hist, err0 = kr.bpcg_v2(oA, sysm.B, pa, ps, f, g, float(d["k"]),
                                          tol=float(d["tol"]), maxsteps=int(d["maxsteps"]), condensed=ocond)
        assert abs(err0 - float(d["err0"])) <= 1e-10 * float(d["err0"])
        _close_history(hist, d["history"], d["window"])
        _close_iterations(it, d["iterations"], d)
    else:
        u, p, errors, warned = kr.minres(sysm.A, sysm.B, pa, ps, f, g,
                                         maxsteps=int(d["maxsteps"]), tol=float(d["tol"]))
        _close_history(errors, d["errors"], d["window"])
        _close_iterations(len(errors) - 1, d["iterations"], d)
        assert warned == bool(d["warned"])
    x = np.concatenate([u, p])
    assert abs(np.linalg.norm(x) - float(d["x_norm"])) <= 1e-6 * float(d["x_norm"])
    res = np.linalg.norm(b - Ks @ x)
    assert res <= 10 * max(float(d["residual"]), 1e-12 * float(d["b_norm"]))
    np.testing.assert_allclose(x[d["sample_idx"]], d["sample_val"], rtol=0, atol=1e-6 * np.abs(x).max())


def test_oracle_lanczos_scale_factor_matches_golden():
    """k recorded in the goldens came from hipla/eigen.py; oracle/krylov_ref.lanczos_ritz is an
    independent statement of the same recurrence (NGSolve's own is upstream: parity unpinned)."""
    for case, tol in [("stokes2d_n12_jacobi_bpcg1", 1e-10), ("stokes2d_n12_jacobi_bpcg2", 1e-3),
                      ("stokes3d_n10_bjac_bpcg2", 1e-3), ("stokes2d_n24_bjac_bpcg1", 1e-10)]:
        d = np.load(golden_path(case))
        sysm, f, g, pa, ps = _system(d)
        ritz = kr.lanczos_ritz(sysm.A, pa, tol=tol)
        assert abs(kr.scale_factor(ritz) - float(d["k"])) <= 1e-9 * float(d["k"])
        assert abs(ritz.min() - float(d["lam_min"])) <= 1e-9 * float(d["lam_min"])
    # Ritz values bracket the true spectrum of pre*A from inside
    sysm = mac_stokes(2, 12)
    dinv = 1.0 / sysm.A.diagonal()
    true = np.linalg.eigvalsh((np.sqrt(dinv)[:, None] * sysm.A.toarray()) * np.sqrt(dinv)[None, :])
    ritz = kr.lanczos_ritz(sysm.A, kr.jacobi(sysm.A), tol=1e-10)
    assert ritz.min() >= true.min() * (1 - 1e-9) and ritz.max() <= true.max() * (1 + 1e-9)
    assert abs(ritz.min() - true.min()) <= 1e-6 * true.min()


def test_quirk_minres_absolute_guard():
    d = np.load(golden_path("quirk_minres_absolute_guard"))
    sysm, f, g, pa, ps = _system(d)
    u, p, errors, warned = kr.minres(sysm.A, sysm.B, pa, ps, float(d["rhs_scale"]) * f, g,
                                     maxsteps=int(d["maxsteps"]), tol=float(d["tol"]))
    assert bool(d["warned"]) and warned          # converged absolutely, still warns (minres.py:96,145-146)
    _close_iterations(len(errors) - 1, d["iterations"], d)
    assert errors[-1] > float(d["tol"])          # relative criterion was NOT met


def test_quirk_warm_starts_and_abs_err():
    rng = np.random.default_rng(7)
    d = np.load(golden_path("quirk_minres_warm_start"))
    sysm, f, g, pa, ps = _system(d)
    x0 = (0.1 * rng.standard_normal(sysm.n_u), 0.1 * rng.standard_normal(sysm.n_p))
    u, p, errors, warned = kr.minres(sysm.A, sysm.B, pa, ps, f, g, x0=x0, maxsteps=int(d["maxsteps"]), tol=float(d["tol"]))
    _close_iterations(len(errors) - 1, d["iterations"], d)
    np.testing.assert_allclose(errors[:60], d["errors"][:60], rtol=1e-8)
    assert bool(d["aliased"])

    for name in ("quirk_bpcg2_warm_start", "quirk_bpcg2_abs_err"):
        d = np.load(golden_path(name))
        start = x0 if not bool(d["initialize"]) else None
        it, u, p, hist, err0 = kr.bpcg_v2(sysm.A, sysm.B, pa, ps, f, g, float(d["k"]), x0=start,
                                          tol=float(d["tol"]), maxsteps=int(d["maxsteps"]), rel_err=bool(d["rel_err"]))
        _close_iterations(it, d["iterations"], d)
        np.testing.assert_allclose(hist[:40], d["history"][:40], rtol=1e-8)
        assert abs(err0 - float(d["err0"])) <= 1e-10 * err0

    d = np.load(golden_path("quirk_bpcg1_warm_start_maxsteps"))
    u, p, errors, conv = kr.bpcg_v1(sysm.A, sysm.B, pa, ps, f, g, float(d["k"]), x0=x0,
                                    tolerance=float(d["tol"]), max_steps=int(d["maxsteps"]))
    assert not conv and bool(d["warned"])
    assert len(errors) == int(d["iterations"]) == int(d["maxsteps"])
    np.testing.assert_allclose(errors, d["errors"], rtol=1e-8)


def test_quirk_bpcg2_zero_rhs():
    d = np.load(golden_path("quirk_bpcg2_zero_rhs"))
    sysm, f, g, pa, ps = _system(d)
    it, u, p, hist, err0 = kr.bpcg_v2(sysm.A, sysm.B, pa, ps, 0 * f, g, 10.0)
    assert it == -1 and err0 == 0.0 and len(hist) == 0
    assert bool(d["returned_solution_object"]) and float(d["x_norm"]) == 0.0


def test_cfg1_heat_plumbing():
    d = np.load(golden_path("cfg1_heat_plumbing"))
    M = diffusion_2d(int(d["n"]))
    assert M.shape[0] == int(d["rows"]) == 4096 and M.nnz == int(d["nnz"]) == 20224
    np.testing.assert_allclose(d["gram"], np.eye(5), atol=1e-12)
    x0 = np.random.default_rng(int(d["seed"])).standard_normal(M.shape[0])
    x, hist = kr.cg(M, x0, tol=1e-10, maxsteps=500)
    assert len(hist) - 1 == int(d["cg_iterations"])
    np.testing.assert_allclose(hist, d["cg_history"], rtol=1e-9)
    import scipy.sparse.linalg as spl
    np.testing.assert_allclose(x, spl.spsolve(M.tocsc(), x0), rtol=0, atol=1e-9)


def test_golden_standin_is_independent_of_the_product():
    """The stand-in the stored goldens were produced over (tests/ngsolve_numpy) shares no code with
    the product: it imports only numpy / scipy / the standard library."""
    import ast
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ngsolve_numpy", "ngsolve")
    allowed = {"math", "time", "numpy", "scipy", ""}
    for name in sorted(os.listdir(root)):
        if not name.endswith(".py"):
            continue
        with open(os.path.join(root, name)) as fh:
            tree = ast.parse(fh.read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""] if node.level == 0 else []
            for m in mods:
                assert m.split(".")[0] in allowed, (name, m)


def test_oracle_mypre_a_is_the_symmetric_multiplicative_composition():
    """`kr.mypre_a` (MypreA.Mult, templates/NavierStokesSIMPLE_iterative.py:375-383) against the closed form of
    what its five statements compose.  GS=True: with the block lower / upper triangles (D + L), (D + U) of A in the
    block order of the sweep, y = P x has the error propagation I - P A = (I - (D+U)^-1 A)(I - M A)(I - (D+L)^-1 A)
    (forward sweep, auxiliary correction, backward sweep) and P is symmetric positive definite; GS=False:
    P = M + J.  M = T (sum_c E_c L_c^-1 E_c^T) T^T built densely here, independently of `kr.auxiliary_space_term`."""
    s = mac_stokes(2, 7, 0.01)
    A = s.A.toarray()
    n = s.n_u
    blocks = s.line_blocks(3)
    space = s.auxiliary_space()
    T = space["transform"].toarray()
    inner = np.zeros((T.shape[1], T.shape[1]))
    for lap, rng in zip(space["laplacians"], space["ranges"]):
        idx = np.arange(rng.start, rng.stop)
        inner[np.ix_(idx, idx)] = np.linalg.inv(lap.toarray())
    M = T @ inner @ T.T
    aux = kr.auxiliary_space_term(space["transform"], space["laplacians"], space["ranges"])
    x = np.random.default_rng(4).standard_normal(n)
    np.testing.assert_allclose(aux(x), M @ x, rtol=1e-11, atol=1e-13)
    # block triangles in the order of the sweep
    order = [blocks[:, b][blocks[:, b] >= 0] for b in range(blocks.shape[1])]
    covered = np.concatenate(order)
    assert np.array_equal(np.sort(covered), np.arange(n))              # the line blocks tile the velocity dofs
    rank = np.empty(n, dtype=np.int64)
    for b, dofs in enumerate(order):
        rank[dofs] = b
    lower = np.where(rank[:, None] >= rank[None, :], A, 0.0)           # D + L (block lower triangle)
    upper = np.where(rank[:, None] <= rank[None, :], A, 0.0)           # D + U
    eye = np.eye(n)
    E = (eye - np.linalg.solve(upper, A)) @ (eye - M @ A) @ (eye - np.linalg.solve(lower, A))
    P = (eye - E) @ np.linalg.inv(A)
    gs = kr.mypre_a(s.A, blocks, aux, True)
    got = np.column_stack([gs(eye[:, j]) for j in range(n)])
    np.testing.assert_allclose(got, P, rtol=0, atol=1e-10 * np.abs(P).max())
    assert np.abs(got - got.T).max() < 1e-10 * np.abs(got).max() and np.linalg.eigvalsh(0.5 * (got + got.T)).min() > 0
    add = kr.mypre_a(s.A, blocks, aux, False)
    J = np.column_stack([kr.block_jacobi(s.A, blocks)(eye[:, j]) for j in range(n)])
    np.testing.assert_allclose(add(x), (M + J) @ x, rtol=1e-11, atol=1e-13)
    # no auxiliary term: the symmetric sweep of this file
    np.testing.assert_allclose(kr.mypre_a(s.A, blocks, None, True)(x), kr.symmetric_block_gauss_seidel(s.A, blocks)(x),
                               rtol=1e-13, atol=1e-15)


def test_oracle_time_step_is_the_kkt_solution():
    """`kr.do_time_step` / `kr.project` (DoTimeStep, Project: templates/NavierStokesSIMPLE_iterative.py:424-443)
    against one dense solve of what the statements amount to: temp2 is the M_u-orthogonal projection of
    mstar^-1 (conv(u) + f - A u) onto the discretely divergence-free fields, i.e. the first component of
    [[M_u, B^T], [B, 0]] [t; phi] = [M_u raw; 0]; the donor-cell convection term against direct loops over
    the grid is covered in test_drivers_cpu.py."""
    s = mac_stokes(2, 9, 0.01)
    rng = np.random.default_rng(6)
    u0, f = rng.standard_normal(s.n_u), rng.standard_normal(s.n_u)
    m_u = np.full(s.n_u, s.h ** s.dim)
    dt = 0.05
    cops = s.convection_operators()
    out = kr.do_time_step(s.A, s.B, m_u, dt, u0, f, lambda u: kr.upwind_convection(cops, u))
    A, B = s.A.toarray(), s.B.toarray()
    adv, avg, dif = cops["adv"] @ u0, cops["avg"] @ u0, cops["diff"] @ u0
    rhs = -(cops["div"] @ (adv * avg - 0.5 * np.abs(adv) * dif)) + f - A @ u0
    np.testing.assert_allclose(out["temp"], rhs, rtol=1e-13, atol=1e-13)
    raw = np.linalg.solve(np.diag(m_u) + dt * A, rhs)
    np.testing.assert_allclose(out["temp2_unprojected"], raw, rtol=1e-10, atol=1e-12)
    # KKT system, pressure determined up to a constant: least-squares solve of the consistent system
    K = np.block([[np.diag(m_u), B.T], [B, np.zeros((s.n_p, s.n_p))]])
    sol = np.linalg.lstsq(K, np.concatenate([m_u * raw, np.zeros(s.n_p)]), rcond=None)[0]
    np.testing.assert_allclose(out["temp2"], sol[:s.n_u], rtol=0, atol=1e-9 * np.abs(raw).max())
    assert np.linalg.norm(B @ out["temp2"]) < 1e-10 * np.linalg.norm(B @ raw)
    np.testing.assert_allclose(out["u"], u0 + dt * out["temp2"], rtol=0, atol=1e-15)
    # Project alone, idempotent
    v1, _ = kr.project(s.B, m_u, u0)
    v2, _ = kr.project(s.B, m_u, v1)
    assert np.linalg.norm(B @ v1) < 1e-10 * np.linalg.norm(B @ u0) and np.linalg.norm(v2 - v1) < 1e-10 * np.linalg.norm(v1)
