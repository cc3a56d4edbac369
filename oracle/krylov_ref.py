"""TEST INFRASTRUCTURE -- CPU oracle: plain numpy/scipy restatement of the
reference's three Krylov loops on ``[[A, B^T], [B, 0]]``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; it is the checker, never the product path.

Pinning: the reference has no tests, goldens or fixtures, and
its arithmetic lives in NGSolve, which is not installable here.  This restatement is
pinned by golden vectors generated in the build container from the reference's
*unmodified* files (minres.py, bramble_pasciak_cg.py,
solvers/bramblepasciak_new.py) run over the protocol layer with numpy arithmetic
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).  Against real NGSolve output: parity unpinned.

Each function cites the reference lines it follows.  Operands: scipy CSR ``A``
(n_u x n_u), ``B`` (n_p x n_u); ``pre_a`` / ``pre_s`` are callables ``x -> P x``.

``mypre_a`` / ``auxiliary_space_term`` (MypreA.Mult, templates/NavierStokesSIMPLE_iterative.py:375-383) and
``do_time_step`` / ``project`` / ``upwind_convection`` (:424-443) restate the reference's statements over
the staggered-grid operators; the reference holds no fixture for them and their FE operands (facet blocks,
h1amg, BDDC, the HDG convection form) are NGSolve's: **parity unpinned** -- they are checked against their own
algebraic identities in ``tests/test_oracle_golden.py`` and are what the GPU path is compared with.
"""

from math import sqrt
import time

import numpy as np
import scipy.sparse as sp


# --------------------------------------------------------------------------
# preconditioners
# --------------------------------------------------------------------------
def jacobi(A):
    """Point Jacobi = Preconditioner(.., 'local') (run.py:62)."""
    dinv = 1.0 / A.diagonal()
    return lambda x: dinv * x


def diag_inverse(d):
    dinv = 1.0 / np.asarray(d, dtype=np.float64)
    return lambda x: dinv * x


def block_jacobi(A, idx):
    """Additive block Jacobi over disjoint blocks ``idx`` (bs, nblocks), -1 = padding:
    J = sum_b E_b A_bb^-1 E_b^T  (templates/NavierStokesSIMPLE_iterative.py:360-373,383)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    bs, nb = idx.shape
    safe = np.where(idx >= 0, idx, 0)
    blocks = np.zeros((nb, bs, bs))
    for r in range(bs):
        for c in range(bs):
            both = (idx[r] >= 0) & (idx[c] >= 0)
            vals = np.asarray(A[safe[r], safe[c]]).ravel()
            blocks[:, r, c] = np.where(both, vals, 1.0 if r == c else 0.0)
    inv = np.linalg.inv(blocks)
    live = idx >= 0
    n = A.shape[0]

    def apply(x):
        xb = np.where(live, x[safe], 0.0)
        yb = np.einsum("krc,ck->rk", inv, xb)
        y = np.zeros(n)
        y[idx[live]] = yb[live]
        return y

    return apply


def block_gauss_seidel_sweep(A, idx, x, y, backward=False):
    """One *sequential* block Gauss-Seidel sweep for A y = x over the blocks of `idx` in the given
    order (reversed if `backward`): y_b += A_bb^-1 (x_b - (A y)_b), one block after the other --
    `jacobi.Smooth` / `jacobi.SmoothBack` of templates/NavierStokesSIMPLE_iterative.py:378,381.
    Plain Python loop: small cases only.  The multicolour GPU sweep must agree with this when
    handed its colour-major block order (same-colour blocks are uncoupled)."""
    A = sp.csr_matrix(A)
    y = np.array(y, dtype=np.float64)
    nb = idx.shape[1]
    order = range(nb - 1, -1, -1) if backward else range(nb)
    for b in order:
        dofs = idx[:, b]
        dofs = dofs[dofs >= 0]
        res = x[dofs] - A[dofs] @ y
        y[dofs] += np.linalg.solve(A[dofs][:, dofs].toarray(), res)
    return y


def symmetric_block_gauss_seidel(A, idx):
    """y = 0; forward sweep; backward sweep (MypreA with GS=True minus the AMG term, :376-381)."""
    def apply(x):
        y = block_gauss_seidel_sweep(A, idx, x, np.zeros(A.shape[0]), backward=False)
        return block_gauss_seidel_sweep(A, idx, x, y, backward=True)
    return apply


def auxiliary_space_term(transform, laplacians, ranges, component_solve=None):
    """``transform @ preAh1 @ transform.T`` with ``preAh1 = sum_c emb_c @ pre_c @ emb_c.T``
    (templates/NavierStokesSIMPLE_iterative.py:334-337,353-357 as used at :380,:383): ``transform`` maps the
    stacked per-component auxiliary (nodal) spaces to the velocity dofs, ``ranges[c]`` are the dofs of
    component c inside the stacked space (``Embedding(fesh1.ndof, fesh1.Range(c))``), ``pre_c`` is the
    component's ``Preconditioner(aH1_c, 'h1amg')``.  Default ``pre_c`` = the exact inverse of the component
    Laplacian (sparse LU): what an AMG hierarchy that ends on its first level applies, and an operator this file
    can state without restating NGSolve's upstream hierarchy.  ``component_solve(c)`` may supply another
    callable per component (e.g. a V-cycle applied as a black box)."""
    import scipy.sparse.linalg as spl
    T = sp.csr_matrix(transform)
    TT = T.T.tocsr()
    solves = []
    for c, lap in enumerate(laplacians):
        solves.append(component_solve(c) if component_solve is not None else spl.splu(sp.csc_matrix(lap)).solve)

    def apply(x):
        r = TT @ x                                   # transform.T
        e = np.zeros_like(r)
        for rng, solve in zip(ranges, solves):       # emb_c @ pre_c @ emb_c.T, summed over the components
            sl = np.asarray(rng) if not isinstance(rng, (range, slice)) else rng
            e[sl] = e[sl] + solve(r[sl])
        return T @ e                                 # transform
    return apply


def mypre_a(A, blocks, aux, gs):
    """``MypreA.Mult`` (templates/NavierStokesSIMPLE_iterative.py:375-383), statement by statement, with the
    *sequential* block sweeps of this file: `blocks` (bs, nblocks) in the order the sweep visits them, `aux` =
    callable for ``transform @ preAh1 @ transform.T`` (None: no auxiliary term).  Returns x -> y."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    jac = block_jacobi(A, blocks)
    middle = aux if aux is not None else (lambda r: np.zeros_like(r))

    def apply(x):
        if gs:
            y = np.zeros(n)                                                   # :377  y[:] = 0
            y = block_gauss_seidel_sweep(A, blocks, x, y, backward=False)     # :378  jacobi.Smooth(y, x)
            temp = x - A @ y                                                  # :379
            y = y + middle(temp)                                              # :380
            y = block_gauss_seidel_sweep(A, blocks, x, y, backward=True)      # :381  jacobi.SmoothBack(y, x)
            return y
        return middle(x) + jac(x)                                             # :383
    return apply


# --------------------------------------------------------------------------
# IMEX time step (scope row N4)
# --------------------------------------------------------------------------
def upwind_convection(ops, u):
    """``conv_operator * gfu`` (templates/NavierStokesSIMPLE_iterative.py:106-113,429) on the staggered grid: the
    weak form of -div(u (x) u) with upwind fluxes = donor-cell fluxes F = adv * avg - |adv| * diff / 2 through the
    faces of every control volume, conv = -D F.  `ops`: dict of the sparse operators adv, avg, diff, div."""
    adv, avg, dif = ops["adv"] @ u, ops["avg"] @ u, ops["diff"] @ u
    return -(ops["div"] @ (adv * avg - 0.5 * np.abs(adv) * dif))


def project(B, m_u, vel, solve=None):
    """``Project(vel)`` (templates/NavierStokesSIMPLE_iterative.py:440-443) on the staggered grid:
    phi = (B M_u^-1 B^T)^-1 B vel; vel -= M_u^-1 B^T phi.  Direct solve (sparse LU); on an enclosed domain the
    pressure operator has the constants in its kernel and B vel is orthogonal to them: one pressure is pinned
    (the correction M_u^-1 B^T phi does not see the constant).  Returns (projected velocity, phi)."""
    import scipy.sparse.linalg as spl
    B = sp.csr_matrix(B)
    bt = (sp.diags(1.0 / m_u) @ B.T).tocsr()
    L = (B @ bt).tocsc()
    rhs = B @ vel
    if solve is not None:
        phi = solve(rhs)
    elif np.linalg.norm(L @ np.ones(L.shape[0])) <= 1e-12 * abs(L).sum():
        keep = np.arange(L.shape[0] - 1)
        phi = np.zeros(L.shape[0])
        phi[keep] = spl.splu(sp.csc_matrix(L[keep][:, keep])).solve(rhs[keep])
    else:
        phi = spl.splu(L).solve(rhs)
    return vel - bt @ phi, phi


def do_time_step(A, B, m_u, timestep, u, f, conv, solve_mstar=None, solve_proj=None):
    """``DoTimeStep`` (templates/NavierStokesSIMPLE_iterative.py:424-438), statement by statement:
    temp = conv(u) + f - A u (:429-431); temp2 = invmstar temp with mstar = M_u + timestep A (:84-85,433);
    Project(temp2) (:434); u += timestep temp2 (:438).  The reference applies ``invmstar`` and ``invproj`` by
    inner CG solves (precision 1e-4, :92,130); this restatement solves directly (sparse LU) unless a callable is
    given.  Returns dict(u=new velocity, temp=..., temp2_unprojected=..., temp2=..., phi=...)."""
    import scipy.sparse.linalg as spl
    A = sp.csr_matrix(A)
    temp = conv(u)                                    # :429
    temp = temp + f                                   # :430
    temp = temp + (-(A @ u))                          # :431
    if solve_mstar is None:
        solve_mstar = spl.splu((sp.diags(m_u) + timestep * A).tocsc()).solve
    raw = solve_mstar(temp)                           # :433
    temp2, phi = project(B, m_u, raw, solve_proj)     # :434
    return dict(u=u + timestep * temp2, temp=temp, temp2_unprojected=raw, temp2=temp2, phi=phi)   # :438


# --------------------------------------------------------------------------
# eigenvalue estimate -> scale factor k
# --------------------------------------------------------------------------
def lanczos_start(n, offset=0):
    i = (np.arange(offset, offset + n, dtype=np.uint64) + np.uint64(1)) * np.uint64(2654435761)
    i = (i ^ (i >> np.uint64(15))) & np.uint64(0xFFFFFFFF)
    return i.astype(np.float64) / 4294967296.0 - 0.5


def lanczos_ritz(A, pre, tol=1e-10, maxsteps=2000, check_every=5, start=None):
    """Ritz values of pre*A (the build's EigenValues_Preconditioner; call sites
    bramble_pasciak_cg.py:70-71, solvers/bramblepasciak_new.py:115-118).  NGSolve's own
    Lanczos is upstream: parity unpinned; this mirrors hipla/eigen.py's recurrence."""
    from scipy.linalg import eigvalsh_tridiagonal
    n = A.shape[0]
    v = lanczos_start(n) if start is None else np.array(start, dtype=np.float64)
    v_old = np.zeros(n)
    z = pre(v)
    gamma = sqrt(abs(np.dot(z, v)))
    if gamma == 0.0:
        return np.zeros(0)
    z = z / gamma
    v = v / gamma
    diag, off = [], []
    lo_prev = hi_prev = None
    ritz = np.zeros(0)
    scale0 = None
    for j in range(maxsteps):
        p = A @ z
        delta = float(np.dot(p, z))
        v_new = p - delta * v - gamma * v_old
        z_new = pre(v_new)
        gamma_new = sqrt(abs(float(np.dot(z_new, v_new))))
        diag.append(delta)
        if scale0 is None:
            scale0 = abs(delta)
        breakdown = gamma_new <= 1e-14 * max(scale0, abs(delta))
        if breakdown or (j + 1) % check_every == 0 or j + 1 == maxsteps:
            ritz = (np.array(diag) if len(diag) == 1
                    else eigvalsh_tridiagonal(np.array(diag), np.array(off[: len(diag) - 1])))
            lo, hi = float(ritz[0]), float(ritz[-1])
            if breakdown:
                break
            if lo_prev is not None and abs(lo - lo_prev) <= tol * abs(lo) and abs(hi - hi_prev) <= tol * abs(hi):
                break
            lo_prev, hi_prev = lo, hi
        off.append(gamma_new)
        z_new = z_new / gamma_new
        v_new = v_new / gamma_new
        v_old, v = v, v_new
        z = z_new
        gamma = gamma_new
    return ritz


def scale_factor(ritz):
    """k = 1/min(lambda) + 1e-3 (bramble_pasciak_cg.py:71, bramblepasciak_new.py:118)."""
    return 1.0 / float(np.min(ritz)) + 1e-3


# --------------------------------------------------------------------------
# BPCG v1  (bramble_pasciak_cg.py:65-148)
# --------------------------------------------------------------------------
def bpcg_v1(A, B, pre_a, pre_s, f, g, k, x0=None, tolerance=1e-12, max_steps=1000):
    """Returns (u, p, errors) with errors[i] = err_i / err_0 (len = iterations + 1).

    full_pre_a = diag(k*pre_a, I) (:75,79-80); full_b = [[I,0],[B,-I]] (:81-82);
    a_b = [A;B] applied to the velocity part (:39-47,83);
    full_pre_schur = diag(I, pre_s) (:84-85)."""
    BT = B.T.tocsr()
    n_u, n_p = A.shape[0], B.shape[0]
    u = np.zeros(n_u) if x0 is None else np.array(x0[0], dtype=np.float64)
    p = np.zeros(n_p) if x0 is None else np.array(x0[1], dtype=np.float64)

    def K(xu, xp):                               # original_matrix (:77-78), C = None
        return A @ xu + BT @ xp, B @ xu

    def schur_b(au, ap):                         # full_pre_schur @ full_b (:102,135)
        return au.copy(), pre_s(B @ au - ap)

    ku, kp = K(u, p)
    t2u, t2p = f - ku, g - kp                    # :98
    au, ap = k * pre_a(t2u), t2p.copy()          # :99   a_preconditioned_residuum
    ru = A @ au - f + ku                         # :100-101 residuum
    rp = B @ au - g + kp
    t1u, t1p = schur_b(au, ap)                   # :102
    du, dp = t1u.copy(), t1p.copy()              # :103  full_preconditioned_residuum
    cur = float(np.dot(t1u, ru) + np.dot(t1p, rp))   # :105
    err0 = sqrt(abs(cur))
    errors = []
    converged = False
    for _ in range(max_steps):
        err = sqrt(abs(cur))                     # :115
        errors.append(err / err0)                # :118
        if err < tolerance * err0:               # :119
            converged = True
            break
        prev = cur
        ku, kp = K(du, dp)
        t1u, t1p = -ku, -kp                      # :125
        t2u, t2p = k * pre_a(ku), -t1p           # :126  temp_2 = -full_pre_a * temp_1
        t1u = t1u + A @ t2u                      # :127
        t1p = t1p + B @ t2u
        alpha = prev / float(np.dot(du, t1u) + np.dot(dp, t1p))   # :129-130
        u += alpha * du                          # :131
        p += alpha * dp
        ru -= alpha * t1u                        # :132
        rp -= alpha * t1p
        au -= alpha * t2u                        # :133
        ap -= alpha * t2p
        t1u, t1p = schur_b(au, ap)               # :135
        cur = float(np.dot(t1u, ru) + np.dot(t1p, rp))            # :137
        beta = cur / prev                        # :138
        du = beta * du + t1u                     # :140-141
        dp = beta * dp + t1p
    return u, p, np.array(errors), converged


# --------------------------------------------------------------------------
# BPCG v2  (solvers/bramblepasciak_new.py:24-253, both branches of harmonic_extension)
# --------------------------------------------------------------------------
class CondensedOperator:
    """`myAmatrix` (solvers/bramblepasciak_new.py:84-103): x -> (I - H^T)(S + A_ii)(I - H) x,
    evaluated right to left as the reference's composite operator does (:88,91)."""

    def __init__(self, S, inner_matrix, H, HT):
        self.mid = (sp.csr_matrix(S) + sp.csr_matrix(inner_matrix)).tocsr()
        self.H, self.HT = sp.csr_matrix(H), sp.csr_matrix(HT)
        self.shape = self.mid.shape

    def __matmul__(self, x):
        t = x - self.H @ x
        t = self.mid @ t
        return t - self.HT @ t


def bpcg_v2(A, B, pre_a_unscaled, pre_m, f, g, k, x0=None, tol=1e-6, maxsteps=100,
            rel_err=True, timing=None, condensed=None):
    """Returns (it, u, p, history, err0) where history[i] = sqrt(|wd|) printed at
    iteration i (:243-245) and history has it+1 entries.  `x0` = (u0, p0) is the
    warm start (initialize=False, :137-139); None = zero start.

    `condensed` = dict(harmonic_extension, harmonic_extension_trans, inner_solve, inner_matrix)
    (scipy CSR, n_u x n_u) selects the `blfA.condense` branch: `A` is then the Schur complement
    `blfA.mat`, the operator is `myAmatrix` (:84-109) and every preconditioner apply goes through
    the condensed branch of `harmonic_extension` (:11-18)."""
    BT = B.T.tocsr()                             # :198 (built once)
    n_u, n_p = A.shape[0], B.shape[0]

    if condensed is None:
        def pre_a(x):                            # preA = k * preA_unscaled (:122), plain branch of harmonic_extension (:20)
            return k * pre_a_unscaled(x)
    else:
        H, HT = sp.csr_matrix(condensed["harmonic_extension"]), sp.csr_matrix(condensed["harmonic_extension_trans"])
        inner_solve = sp.csr_matrix(condensed["inner_solve"])
        A = CondensedOperator(A, condensed["inner_matrix"], H, HT)     # :105-106

        def pre_a(x):                            # harmonic_extension(), condensed branch
            f_residual = x + HT @ x              # :12-13
            result = k * pre_a_unscaled(f_residual)   # :15
            result = result + H @ result         # :16
            return result + inner_solve @ f_residual  # :17

    tmp0 = pre_a(f)                              # :129
    f_new = A @ tmp0 - f                         # :130
    g_new = B @ tmp0 - g                         # :133
    u0 = np.zeros(n_u) if x0 is None else np.array(x0[0], dtype=np.float64)
    u1 = np.zeros(n_p) if x0 is None else np.array(x0[1], dtype=np.float64)

    tmp0 = A @ u0 + BT @ u1                      # :160
    tmp1 = pre_a(tmp0)                           # :162
    tmp2 = A @ tmp1                              # :163
    tmp4 = tmp1 - u0                             # :165
    tmp3 = B @ tmp4                              # :166
    d0 = f_new - (tmp2 - tmp0)                   # :169
    d1 = g_new - tmp3                            # :170
    pr0 = pre_a(f)                               # :173
    pr1 = pre_m(B @ pr0 - g)                     # :177-179
    w0 = pr0 - tmp1                              # :182
    w1 = pr1 - pre_m(tmp3)                       # :183
    wdn = float(np.dot(w0, d0) + np.dot(w1, d1))  # :185
    err0 = sqrt(abs(wdn))                        # :187
    s0, s1 = w0.copy(), w1.copy()                # :189
    hist = []
    if wdn == 0:                                 # :191-192
        return -1, u0, u1, np.array(hist), err0
    z0 = np.zeros(n_u)
    z_old0 = np.zeros(n_u)
    matA_s0 = np.zeros(n_u)
    alpha = beta = 0.0
    t_start = time.perf_counter()
    it = -1
    converged = False
    for it in range(maxsteps):                   # :200
        if it == 0:
            matA_s0 = A @ s0                     # :202
            z0 = matA_s0.copy()                  # :203
        else:
            matA_s0 = beta * matA_s0 + z_old0 - alpha * tmp2   # :205
        matB_s1 = BT @ s1                        # :206
        tmp0 = matA_s0 + matB_s1                 # :207
        tmp1 = pre_a(tmp0)                       # :209
        tmp2 = A @ tmp1                          # :210
        tmp4 = tmp1 - s0                         # :212
        tmp3 = B @ tmp4                          # :213
        z_old0 = z0.copy()                       # :215
        v0 = tmp2 - tmp0                         # :218
        v1 = tmp3                                # :219
        wd = wdn                                 # :221
        as_s = float(np.dot(s0, v0) + np.dot(s1, v1))   # :222
        alpha = wd / as_s                        # :226
        u0 += alpha * s0                         # :228
        u1 += alpha * s1
        d0 += (-alpha) * v0                      # :229
        d1 += (-alpha) * v1
        w0 = w0 + (-alpha) * tmp1                # :232
        w1 = w1 + (-alpha) * pre_m(tmp3)         # :233
        wdn = float(np.dot(w0, d0) + np.dot(w1, d1))    # :235
        beta = wdn / wd                          # :236
        z0 -= alpha * tmp2                       # :238
        s0 = beta * s0 + w0                      # :240-241
        s1 = beta * s1 + w1
        err = sqrt(abs(wd))                      # :243
        hist.append(err)
        if err < tol * (err0 if rel_err else 1): # :246
            converged = True
            break
    if timing is not None:
        timing["loop_seconds"] = time.perf_counter() - t_start
        timing["converged"] = converged
    return it, u0, u1, np.array(hist), err0


# --------------------------------------------------------------------------
# MINRES  (minres.py:12-149)
# --------------------------------------------------------------------------
def minres(A, B, pre_a, pre_s, f, g, x0=None, maxsteps=100, tol=1e-7):
    """K = [[A,B^T],[B,0]], C = diag(pre_a, pre_s) (run.py:45-46).  Returns
    (u, p, errors, warned) -- `warned` reproduces the while/else "did not converge"
    message path (:96,145-146), taken whenever the loop ends without the relative
    break at :126 (also when the absolute guard ``ResNorm > tol`` ends it)."""
    BT = B.T.tocsr()
    n_u, n_p = A.shape[0], B.shape[0]

    def K(xu, xp):
        return A @ xu + BT @ xp, B @ xu

    def dot2(au, ap, bu, bp):
        return float(np.dot(au, bu) + np.dot(ap, bp))

    if x0 is None:                               # :62-64
        uu, up = np.zeros(n_u), np.zeros(n_p)
        vu, vp = f.copy(), g.copy()
    else:                                        # :66
        uu, up = np.array(x0[0], dtype=np.float64), np.array(x0[1], dtype=np.float64)
        ku, kp = K(uu, up)
        vu, vp = f - ku, g - kp
    zu, zp = pre_a(vu), pre_s(vp)                # :68
    gamma = sqrt(dot2(zu, zp, vu, vp))           # :71
    gamma_new = 0.0
    zu, zp = (1 / gamma) * zu, (1 / gamma) * zp  # :73
    vu, vp = (1 / gamma) * vu, (1 / gamma) * vp  # :74
    ResNorm = gamma
    err0 = ResNorm
    ResNorm_old = gamma
    eta_old = gamma
    c_old, c = 1.0, 1.0
    s_new, s, s_old = 0.0, 0.0, 0.0
    v_oldu, v_oldp = np.zeros(n_u), np.zeros(n_p)
    w_oldu, w_oldp = np.zeros(n_u), np.zeros(n_p)
    wu, wp = np.zeros(n_u), np.zeros(n_p)
    k = 1
    errors = [1.0]                               # :95
    warned = True
    while k < maxsteps + 1 and ResNorm > tol:    # :96
        mzu, mzp = K(zu, zp)                     # :97
        delta = dot2(mzu, mzp, zu, zp)           # :98
        v_newu = mzu - delta * vu - gamma * v_oldu   # :99
        v_newp = mzp - delta * vp - gamma * v_oldp
        z_newu, z_newp = pre_a(v_newu), pre_s(v_newp)   # :101
        gamma_new = sqrt(dot2(z_newu, z_newp, v_newu, v_newp))   # :103
        z_newu, z_newp = z_newu * (1 / gamma_new), z_newp * (1 / gamma_new)   # :104
        v_newu, v_newp = v_newu * (1 / gamma_new), v_newp * (1 / gamma_new)   # :105
        alpha0 = c * delta - c_old * s * gamma   # :107
        alpha1 = sqrt(alpha0 * alpha0 + gamma_new * gamma_new)
        alpha2 = s * delta + c_old * c * gamma
        alpha3 = s_old * gamma
        c_new = alpha0 / alpha1                  # :112
        s_new = gamma_new / alpha1
        w_newu = (zu - alpha3 * w_oldu - alpha2 * wu) * (1 / alpha1)   # :115-116
        w_newp = (zp - alpha3 * w_oldp - alpha2 * wp) * (1 / alpha1)
        uu += c_new * eta_old * w_newu           # :118
        up += c_new * eta_old * w_newp
        eta = -s_new * eta_old                   # :119
        ResNorm = abs(s_new) * ResNorm_old       # :122
        errors.append(ResNorm / err0)            # :125
        if ResNorm < tol * err0:                 # :126
            warned = False
            break
        k += 1
        v_oldu, v_oldp, vu, vp = vu, vp, v_newu, v_newp      # :131
        w_oldu, w_oldp, wu, wp = wu, wp, w_newu, w_newp      # :132
        zu, zp = z_newu, z_newp                  # :133
        eta_old = eta
        s_old, s = s, s_new
        c_old, c = c, c_new
        gamma = gamma_new
        ResNorm_old = ResNorm
    return uu, up, np.array(errors), warned


# --------------------------------------------------------------------------
# plain CG (cfg1 plumbing: SpMV + dot + AXPY on the 64x64 diffusion matrix)
# --------------------------------------------------------------------------
def cg(M, b, pre=None, tol=1e-10, maxsteps=1000):
    """Textbook preconditioned CG used for the cfg1 plumbing case (the reference's
    heat.py exercises SpMV/InnerProduct/AXPY only: heat.py:89,96-97,110-118)."""
    x = np.zeros_like(b)
    r = b.copy()
    z = pre(r) if pre else r.copy()
    p = z.copy()
    rz = float(np.dot(r, z))
    err0 = sqrt(abs(rz))
    hist = [err0]
    for it in range(maxsteps):
        q = M @ p
        alpha = rz / float(np.dot(p, q))
        x += alpha * p
        r -= alpha * q
        z = pre(r) if pre else r.copy()
        rz_new = float(np.dot(r, z))
        hist.append(sqrt(abs(rz_new)))
        if hist[-1] < tol * err0:
            break
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, np.array(hist)


# ---- smoothed-aggregation set-up (restates csrc/amg_setup.hip; scope row N3) -----------------------
def _neighbour_max(g, values):
    """max over graph neighbours of `values` (0 for isolated nodes)."""
    out = np.zeros(g.shape[0], dtype=values.dtype)
    has = np.diff(g.indptr) > 0
    if g.nnz:
        out[has] = np.maximum.reduceat(values[g.indices], g.indptr[:-1][has])
    return out


def sa_strength_graph(A, theta):
    """Pattern of the strong couplings: off-diagonal and ``|a_ij| >= theta * sqrt(|a_ii| |a_jj|)``
    (theta <= 0: every stored off-diagonal entry)."""
    coo = sp.csr_matrix(A).tocoo()
    d = np.abs(A.diagonal())
    keep = coo.row != coo.col
    if theta > 0.0:
        keep &= np.abs(coo.data) >= theta * np.sqrt(d[coo.row] * d[coo.col])
    g = sp.csr_matrix((np.ones(int(keep.sum()), dtype=np.int8), (coo.row[keep], coo.col[keep])), shape=A.shape)
    g.sort_indices()
    return g


def sa_mis2(g, priority):
    """Roots of the aggregates: maximal independent set of the distance-2 graph of `g` by Luby
    rounds with the given distinct positive priorities (two-hop propagation, g @ g never formed)."""
    cand = np.ones(g.shape[0], dtype=bool)
    roots = np.zeros(g.shape[0], dtype=bool)
    while cand.any():
        pri = np.where(cand, priority, 0)
        one = _neighbour_max(g, pri)
        two = _neighbour_max(g, np.maximum(pri, one))
        winners = cand & (pri >= np.maximum(one, two)) & (pri > one)
        roots |= winners
        w = winners.astype(np.int64)
        near1 = _neighbour_max(g, w)
        near2 = _neighbour_max(g, np.maximum(w, near1))
        cand &= ~(winners | (near1 > 0) | (near2 > 0))
    return roots


def sa_aggregate(A, theta, priority):
    """(aggregate id per node, number of aggregates): roots numbered in index order, four sweeps in
    which an unaggregated node joins its first aggregated strong neighbour, the rest singletons."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    g = sa_strength_graph(A, theta)
    roots = np.nonzero(sa_mis2(g, np.asarray(priority, dtype=np.int64)))[0]
    agg = -np.ones(n, dtype=np.int64)
    agg[roots] = np.arange(roots.size)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(g.indptr))
    cols = g.indices
    for _ in range(4):
        left = agg < 0
        m = left[rows] & (agg[cols] >= 0)
        r, first = np.unique(rows[m], return_index=True)
        agg[r] = agg[cols[m][first]]
    left = np.nonzero(agg < 0)[0]
    agg[left] = roots.size + np.arange(left.size)
    return agg, int(roots.size + left.size)


def sa_spgemm(X, Y):
    """C = X Y row by row in ascending k, every product and sum rounded once (scipy's csr_matmat;
    no FMA) -- the association csrc/amg_setup.hip reproduces on the device."""
    C = (sp.csr_matrix(X) @ sp.csr_matrix(Y)).tocsr()
    C.sort_indices()
    return C


def sa_prolongator(A, agg, nagg, omega):
    """P = T - omega * (D^-1 (A T)), T the piecewise-constant prolongator of `agg`."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    tent = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, int(nagg)))
    AT = sa_spgemm(A, tent)                               # sums that are exactly zero are not stored
    rows = np.repeat(np.arange(n), np.diff(AT.indptr))
    AT.data = omega * ((1.0 / A.diagonal())[rows] * AT.data)
    P = (tent - AT).tocsr()                               # 1 - x, 1 (zero sum) or -x; zeros not stored
    P.sort_indices()
    return P
