"""TEST INFRASTRUCTURE -- numpy/scipy checker engine for the hipla protocol.

Only ``tests/``, ``tests/golden/make_golden.py``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.  It implements the
engine interface of ``hipla/engine.py`` with host arithmetic so that (a) the
reference's *unmodified* solver files can be run over the protocol layer to produce
golden vectors, (b) the host logic is testable without a GPU and (c) the gloo
multi-process tests have a CPU compute leg.  The product never selects it.
"""

import numpy as np
import scipy.sparse as sp


class _Csr:
    def __init__(self, mat):
        self.mat = mat

    @property
    def m(self):
        return self.mat.shape[0]

    @property
    def n(self):
        return self.mat.shape[1]

    @property
    def nnz(self):
        return self.mat.nnz


class _Bjac:
    def __init__(self, idx, inv):
        self.idx = idx      # (bs, nblocks) int32, -1 = padding
        self.inv = inv      # (nblocks, bs, bs)


class NumpyEngine:
    name = "numpy-oracle"

    # ---- buffers -----------------------------------------------------------
    def zeros(self, n):
        return np.zeros(int(n), dtype=np.float64)

    def length(self, buf):
        return int(buf.shape[0])

    def from_host(self, arr):
        return np.array(arr, dtype=np.float64, copy=True)

    def upload(self, arr, buf):
        buf[:] = arr

    def to_host(self, buf):
        return np.array(buf, copy=True)

    def view(self, buf, a, b):
        return buf[a:b]

    def same_buffer(self, a, b):
        return a is b or (a.shape == b.shape and a.__array_interface__["data"][0] == b.__array_interface__["data"][0])

    def overlaps(self, a, b):
        return np.shares_memory(a, b)

    def synchronize(self):
        pass

    # ---- BLAS-1 --------------------------------------------------------------
    def fill(self, buf, c):
        buf[:] = c

    def copy(self, src, dst):
        dst[:] = src

    def scal(self, buf, a):
        buf *= a

    def lincomb(self, dst, terms):
        s0, b0 = terms[0]
        acc = b0 * s0 if s0 != 1.0 else b0.copy()
        for s, b in terms[1:]:
            acc += s * b
        dst[:] = acc

    def index_buffer(self, idx):
        return np.array(idx, dtype=np.int32, copy=True)

    def gather(self, idx, src, dst):
        dst[:] = src[idx]

    def dot(self, x, y):
        return float(np.dot(x, y))

    def dot_multi(self, pairs):
        total = 0.0
        for x, y in pairs:
            total += float(np.dot(x, y))
        return total

    # ---- operators -------------------------------------------------------------
    def csr_create(self, m, n, rowptr, col, val, cuts=None):
        return _Csr(sp.csr_matrix((val, col, rowptr), shape=(m, n)))

    def csr_transpose(self, h):
        t = h.mat.T.tocsr()
        t.sort_indices()
        return _Csr(t)

    def csr_spmv(self, h, alpha, x, beta, y):
        ax = h.mat @ x
        if beta == 0.0:
            y[:] = alpha * ax if alpha != 1.0 else ax
        else:
            if beta != 1.0:
                y *= beta
            y += alpha * ax if alpha != 1.0 else ax

    def csr_to_host(self, h):
        m = h.mat.tocsr()
        m.sort_indices()
        return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float64)

    def csr_inverse_diagonal(self, h):
        return 1.0 / h.mat.diagonal()

    def index_to_host(self, buf):
        return np.asarray(buf)

    # ---- smoothed-aggregation set-up: the CPU restatement in oracle/krylov_ref.py ------------------
    def csr_spgemm(self, x, y, max_products_per_pass=0):
        from . import krylov_ref as kr
        return _Csr(kr.sa_spgemm(x.mat, y.mat))

    def amg_aggregate(self, h, theta, priority):
        from . import krylov_ref as kr
        return kr.sa_aggregate(h.mat, theta, priority)

    def amg_prolongator(self, h, agg, nagg, omega):
        from . import krylov_ref as kr
        return _Csr(kr.sa_prolongator(h.mat, agg, nagg, omega))

    # ---- smoothed-aggregation V(1,1)-cycle on the hierarchy built by hipla/amg.py -----------------
    def amg_create(self, levels, omega):
        return {"levels": levels, "omega": float(omega)}

    def upwind_flux(self, adv, avg, diff, out):
        out[:] = adv * avg - 0.5 * (np.abs(adv) * diff)

    def amg_create_auxiliary(self, T, TT, comps):
        """Auxiliary-space term T (sum_c E_c V_c E_c^T) T^T (nss_amg_create_auxiliary)."""
        return {"T": T.mat, "TT": TT.mat, "comps": list(comps)}

    def amg_apply(self, h, bscale, b, x):
        if "T" in h:
            r = h["TT"] @ (bscale * b)
            z = np.zeros_like(r)
            off = 0
            for comp in h["comps"]:
                n = comp["levels"][0]["n"]
                z[off:off + n] = self._vcycle(comp, 0, r[off:off + n])
                off += n
            x[:] = h["T"] @ z
            return
        x[:] = self._vcycle(h, 0, bscale * b)

    def _vcycle(self, h, l, b):
        lv, w = h["levels"][l], h["omega"]
        if "inv" in lv:
            return lv["inv"].handle.mat @ b
        A, dinv = lv["A"].handle.mat, lv["dinv"]
        x = w * (dinv * b)                                   # pre-smoothing from x = 0
        r = b - A @ x
        x = x + lv["P"].handle.mat @ self._vcycle(h, l + 1, lv["R"].handle.mat @ r)
        return x + w * (dinv * (b - A @ x))                  # post-smoothing

    def diag_apply(self, d, alpha, x, beta, y):
        dx = d * x
        if beta == 0.0:
            y[:] = alpha * dx
        else:
            if beta != 1.0:
                y *= beta
            y += alpha * dx

    def bjac_create(self, csr_handle, idx):
        bs, nb = idx.shape
        a = csr_handle.mat.tocsr()
        a.sort_indices()
        blocks = np.zeros((nb, bs, bs))
        safe = np.where(idx >= 0, idx, 0)
        for r in range(bs):
            for c in range(bs):
                both = (idx[r] >= 0) & (idx[c] >= 0)
                vals = np.asarray(a[safe[r], safe[c]]).ravel()
                blocks[:, r, c] = np.where(both, vals, 1.0 if r == c else 0.0)
        inv = np.linalg.inv(blocks)
        return _Bjac(idx.copy(), inv)

    def bjac_set_colors(self, h, perm_handle, color_ptr, color_rowptr, rowdof, ridx):
        # undo the row permutation: the checker sweeps on the matrix in its original row order
        perm = perm_handle.mat.tocsr()
        coo = perm.tocoo()
        h.mat = sp.csr_matrix((coo.data, (np.asarray(rowdof, dtype=np.int64)[coo.row], coo.col)),
                              shape=(perm.shape[1], perm.shape[1]))
        h.color_ptr = np.array(color_ptr, dtype=np.int64)

    def bjac_smooth(self, h, xscale, x, y, backward):
        """One multicolour block Gauss-Seidel sweep (colour by colour, vectorised per colour)."""
        idx, inv, a = h.idx, h.inv, h.mat
        nc = h.color_ptr.size - 1
        for k in range(nc):
            c = nc - 1 - k if backward else k
            b0, b1 = int(h.color_ptr[c]), int(h.color_ptr[c + 1])
            if b1 <= b0:
                continue
            sub = idx[:, b0:b1]
            live = sub >= 0
            safe = np.where(live, sub, 0)
            r = xscale * x - a @ y
            res = np.where(live, r[safe], 0.0)
            upd = np.einsum("krc,ck->rk", inv[b0:b1], res)
            y[sub[live]] += upd[live]

    def bjac_apply(self, h, alpha, x, beta, y):
        if getattr(h, "color_ptr", None) is not None:      # Gauss-Seidel mode: symmetric sweep operator
            if beta != 0.0:
                raise ValueError("Gauss-Seidel mode supports beta == 0 only")
            y[:] = 0.0
            self.bjac_smooth(h, alpha, x, y, False)
            self.bjac_smooth(h, alpha, x, y, True)
            return
        idx, inv = h.idx, h.inv
        safe = np.where(idx >= 0, idx, 0)
        xb = np.where(idx >= 0, x[safe], 0.0)          # (bs, nb)
        yb = np.einsum("krc,ck->rk", inv, xb)          # (bs, nb)
        if beta == 0.0:
            y[:] = 0.0
        elif beta != 1.0:
            y *= beta
        live = idx >= 0
        y[idx[live]] += alpha * yb[live]
