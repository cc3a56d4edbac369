"""Operators of the hipla protocol: BaseMatrix (subclassable from user code),
SparseMatrix (CSR in HBM), block / scaled / product / sum / transpose / identity
wrappers, and the two native preconditioners of the hot path (point / block Jacobi).

Reference call sites this mirrors: operator algebra
``M*v, -M*v, s*M, M.T, M@N, M+N, IdentityMatrix()-M, -IdentityMatrix(n),
M.CreateTranspose(), .height/.width, Create{Col,Row,}Vector, BlockMatrix([[..]])``
(bramble_pasciak_cg.py:24-36,50-62,77-85,102,125-127,135;
solvers/bramblepasciak_new.py:88,122,160,198; run.py:45-46) and the hooks the
runtime calls back on user subclasses: ``Mult, MultAdd, MultTrans, MultTransAdd,
Height, Width, Create*Vector`` (bramble_pasciak_cg.py:16-36,45-62;
templates/NavierStokesSIMPLE_iterative.py:272-288,375-389).
"""

import numpy as np

from .engine import get_engine
from .vector import BaseVector, BlockVector, Expr, Term, Vector, _DataProxy, _is_scalar, as_expr


class BaseMatrix:
    """Abstract linear operator.  Subclass and override ``Mult`` or ``MultAdd``
    (and ``Height``/``Width``); everything else has defaults."""

    def __init__(self):
        pass

    # ---- shape -----------------------------------------------------------
    def Height(self):
        raise NotImplementedError("%s must implement Height()" % type(self).__name__)

    def Width(self):
        raise NotImplementedError("%s must implement Width()" % type(self).__name__)

    @property
    def height(self):
        return self.Height()

    @property
    def width(self):
        return self.Width()

    def Shape(self):
        return (self.height, self.width)

    # ---- application hooks -----------------------------------------------
    def _overrides(self, name):
        return getattr(type(self), name) is not getattr(BaseMatrix, name)

    def Mult(self, x, y):
        if not self._overrides("MultAdd"):
            raise NotImplementedError("%s implements neither Mult nor MultAdd" % type(self).__name__)
        y[:] = 0.0
        self.MultAdd(1.0, x, y)

    def MultAdd(self, s, x, y):
        if not self._overrides("Mult"):
            raise NotImplementedError("%s implements neither Mult nor MultAdd" % type(self).__name__)
        tmp = y.CreateVector()
        self.Mult(x, tmp)
        y._accumulate(as_expr(tmp) * s)

    def MultTrans(self, x, y):
        if not self._overrides("MultTransAdd"):
            raise NotImplementedError("%s implements neither MultTrans nor MultTransAdd" % type(self).__name__)
        y[:] = 0.0
        self.MultTransAdd(1.0, x, y)

    def MultTransAdd(self, s, x, y):
        if not self._overrides("MultTrans"):
            raise NotImplementedError("%s implements neither MultTrans nor MultTransAdd" % type(self).__name__)
        tmp = y.CreateVector()
        self.MultTrans(x, tmp)
        y._accumulate(as_expr(tmp) * s)

    # ---- vector factories --------------------------------------------------
    def CreateColVector(self):
        return Vector(self.height)

    def CreateRowVector(self):
        return Vector(self.width)

    def CreateVector(self):
        return self.CreateColVector()

    # ---- algebra -------------------------------------------------------------
    def __mul__(self, other):
        if isinstance(other, _DataProxy):
            other = other.owner
        if isinstance(other, BaseVector):
            return Expr([Term(1.0, self, other)])
        if isinstance(other, Expr):
            if len(other.terms) == 1 and other.terms[0].mat is None:
                t = other.terms[0]
                return Expr([Term(t.scale, self, t.vec)])
            return Expr([Term(1.0, self, other.materialize())])
        if _is_scalar(other):
            return ScaledMatrix(float(other), self)
        return NotImplemented

    def __rmul__(self, other):
        if _is_scalar(other):
            return ScaledMatrix(float(other), self)
        return NotImplemented

    def __neg__(self):
        return ScaledMatrix(-1.0, self)

    def __matmul__(self, other):
        if isinstance(other, BaseMatrix):
            return ProductMatrix(self, other)
        return NotImplemented

    def __add__(self, other):
        if isinstance(other, BaseMatrix):
            return SumMatrix(self, other, 1.0)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, BaseMatrix):
            return SumMatrix(self, other, -1.0)
        return NotImplemented

    @property
    def T(self):
        return TransposeMatrix(self)

    def CreateTranspose(self):
        return TransposeMatrix(self)


def _col_like(mat, fallback):
    """Column vector for an intermediate result of `mat`; user subclasses may
    return a plain vector where a block one is needed (reference quirk,
    bramble_pasciak_cg.py:58-59) -- then follow the layout of `fallback`."""
    try:
        v = mat.CreateColVector()
    except NotImplementedError:
        return fallback.CreateVector()
    return v


class ScaledMatrix(BaseMatrix):
    def __init__(self, scale, mat):
        super().__init__()
        if isinstance(mat, ScaledMatrix):
            scale, mat = scale * mat.scale, mat.mat
        self.scale = float(scale)
        self.mat = mat

    def Height(self):
        return self.mat.height

    def Width(self):
        return self.mat.width

    def Mult(self, x, y):
        self.mat.Mult(x, y)
        if self.scale != 1.0:
            y *= self.scale

    def MultAdd(self, s, x, y):
        self.mat.MultAdd(s * self.scale, x, y)

    def MultTrans(self, x, y):
        self.mat.MultTrans(x, y)
        if self.scale != 1.0:
            y *= self.scale

    def MultTransAdd(self, s, x, y):
        self.mat.MultTransAdd(s * self.scale, x, y)

    def CreateColVector(self):
        return self.mat.CreateColVector()

    def CreateRowVector(self):
        return self.mat.CreateRowVector()


class TransposeMatrix(BaseMatrix):
    def __init__(self, mat):
        super().__init__()
        self.mat = mat

    def Height(self):
        return self.mat.width

    def Width(self):
        return self.mat.height

    def Mult(self, x, y):
        self.mat.MultTrans(x, y)

    def MultAdd(self, s, x, y):
        self.mat.MultTransAdd(s, x, y)

    def MultTrans(self, x, y):
        self.mat.Mult(x, y)

    def MultTransAdd(self, s, x, y):
        self.mat.MultAdd(s, x, y)

    def CreateColVector(self):
        return self.mat.CreateRowVector()

    def CreateRowVector(self):
        return self.mat.CreateColVector()

    @property
    def T(self):
        return self.mat


class ProductMatrix(BaseMatrix):
    """``a @ b``: y = a (b x) through one intermediate vector."""

    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b
        self._tmp = None

    def Height(self):
        return self.a.height

    def Width(self):
        return self.b.width

    def _mid(self, x):
        if self._tmp is None:
            self._tmp = _col_like(self.b, x)
        return self._tmp

    def Mult(self, x, y):
        t = self._mid(x)
        self.b.Mult(x, t)
        self.a.Mult(t, y)

    def MultAdd(self, s, x, y):
        t = self._mid(x)
        self.b.Mult(x, t)
        self.a.MultAdd(s, t, y)

    def MultTrans(self, x, y):
        t = self.a.CreateRowVector()
        self.a.MultTrans(x, t)
        self.b.MultTrans(t, y)

    def MultTransAdd(self, s, x, y):
        t = self.a.CreateRowVector()
        self.a.MultTrans(x, t)
        self.b.MultTransAdd(s, t, y)

    def CreateColVector(self):
        return self.a.CreateColVector()

    def CreateRowVector(self):
        return self.b.CreateRowVector()


class SumMatrix(BaseMatrix):
    """``a + sb * b``"""

    def __init__(self, a, b, sb):
        super().__init__()
        self.a, self.b, self.sb = a, b, float(sb)

    def _shape_of(self, which):
        for m in (self.a, self.b):
            v = getattr(m, which)
            if v is not None:
                return v
        return None

    def Height(self):
        return self._shape_of("height")

    def Width(self):
        return self._shape_of("width")

    def Mult(self, x, y):
        self.a.Mult(x, y)
        self.b.MultAdd(self.sb, x, y)

    def MultAdd(self, s, x, y):
        self.a.MultAdd(s, x, y)
        self.b.MultAdd(s * self.sb, x, y)

    def MultTrans(self, x, y):
        self.a.MultTrans(x, y)
        self.b.MultTransAdd(self.sb, x, y)

    def MultTransAdd(self, s, x, y):
        self.a.MultTransAdd(s, x, y)
        self.b.MultTransAdd(s * self.sb, x, y)

    def CreateColVector(self):
        return (self.a if self.a.height is not None else self.b).CreateColVector()

    def CreateRowVector(self):
        return (self.a if self.a.width is not None else self.b).CreateRowVector()


class IdentityMatrix(BaseMatrix):
    """``IdentityMatrix(n)``; ``IdentityMatrix()`` is size-agnostic
    (solvers/bramblepasciak_new.py:88)."""

    def __init__(self, n=None):
        super().__init__()
        self.n = None if n is None else int(n)

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def Mult(self, x, y):
        y._assign(as_expr(x))

    def MultAdd(self, s, x, y):
        y._accumulate(as_expr(x) * s)

    MultTrans = Mult
    MultTransAdd = MultAdd

    @property
    def T(self):
        return self


class BlockMatrix(BaseMatrix):
    """2-D list of operators; ``None`` is a zero block (bramble_pasciak_cg.py:77-85,
    run.py:45-46).  ``y_i = sum_j M_ij x_j`` evaluated block-row by block-row."""

    def __init__(self, rows):
        super().__init__()
        self.rows = [list(r) for r in rows]
        self.nrows = len(self.rows)
        self.ncols = len(self.rows[0])
        if any(len(r) != self.ncols for r in self.rows):
            raise ValueError("ragged BlockMatrix")

    def __getitem__(self, ij):
        i, j = ij
        return self.rows[i][j]

    def _row_op(self, i):
        for m in self.rows[i]:
            if m is not None and m.height is not None:
                return m
        raise ValueError("block row %d has no sized operator" % i)

    def _col_op(self, j):
        for r in self.rows:
            if r[j] is not None and r[j].width is not None:
                return r[j]
        raise ValueError("block column %d has no sized operator" % j)

    def Height(self):
        return sum(self._row_op(i).height for i in range(self.nrows))

    def Width(self):
        return sum(self._col_op(j).width for j in range(self.ncols))

    def CreateColVector(self):
        return BlockVector([Vector(self._row_op(i).height) for i in range(self.nrows)])

    def CreateRowVector(self):
        return BlockVector([Vector(self._col_op(j).width) for j in range(self.ncols)])

    def Mult(self, x, y):
        y[:] = 0.0
        self.MultAdd(1.0, x, y)

    def MultAdd(self, s, x, y):
        for i, row in enumerate(self.rows):
            for j, m in enumerate(row):
                if m is not None:
                    m.MultAdd(s, x[j], y[i])

    def MultTrans(self, x, y):
        y[:] = 0.0
        self.MultTransAdd(1.0, x, y)

    def MultTransAdd(self, s, x, y):
        for i, row in enumerate(self.rows):
            for j, m in enumerate(row):
                if m is not None:
                    m.MultTransAdd(s, x[i], y[j])


class SparseMatrix(BaseMatrix):
    """fp64 CSR matrix resident in engine memory (int32 indices).

    ``SparseMatrix.from_scipy(csr)`` / ``SparseMatrix(m, n, rowptr, col, val)``
    upload host CSR arrays; SpMV runs in the engine (``csr_spmv_f64``).  The
    explicit transpose is built once and cached (reference:
    solvers/bramblepasciak_new.py:198 ``matB.CreateTranspose()``)."""

    def __init__(self, m, n, rowptr, col, val, engine=None, handle=None):
        super().__init__()
        self.engine = engine if engine is not None else get_engine()
        self.m, self.n = int(m), int(n)
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if rowptr.shape != (self.m + 1,) or col.shape != val.shape or int(rowptr[-1]) != col.size:
            raise ValueError("inconsistent CSR arrays")
        if col.size and (col.min() < 0 or col.max() >= self.n):
            raise ValueError("CSR column index out of range")
        self.nnz = int(col.size)
        self.handle = handle if handle is not None else self.engine.csr_create(self.m, self.n, rowptr, col, val)
        self._transpose = None
        self._host = (rowptr, col, val)

    @classmethod
    def from_scipy(cls, csr, engine=None):
        csr = csr.tocsr()
        csr.sort_indices()
        return cls(csr.shape[0], csr.shape[1], csr.indptr, csr.indices, csr.data, engine=engine)

    @classmethod
    def from_handle(cls, handle, engine=None):
        """Wrap a matrix the engine built itself (transpose, sparse product, prolongator); the host
        copy of its arrays is fetched only if somebody asks for it."""
        self = cls.__new__(cls)
        BaseMatrix.__init__(self)
        self.engine = engine if engine is not None else get_engine()
        self.m, self.n, self.nnz = int(handle.m), int(handle.n), int(handle.nnz)
        self.handle = handle
        self._transpose = None
        self._host = None
        return self

    def to_scipy(self):
        import scipy.sparse as sp
        rowptr, col, val = self.host_csr()
        return sp.csr_matrix((val, col, rowptr), shape=(self.m, self.n))

    def host_csr(self):
        if self._host is None:
            self._host = self.engine.csr_to_host(self.handle)
        return self._host

    def Height(self):
        return self.m

    def Width(self):
        return self.n

    def Mult(self, x, y):
        self.engine.csr_spmv(self.handle, 1.0, x.buf, 0.0, y.buf)

    def MultAdd(self, s, x, y):
        self.engine.csr_spmv(self.handle, float(s), x.buf, 1.0, y.buf)

    def MultTrans(self, x, y):
        self.CreateTranspose().Mult(x, y)

    def MultTransAdd(self, s, x, y):
        self.CreateTranspose().MultAdd(s, x, y)

    def CreateTranspose(self):
        """Explicit transpose, built once by the engine (``nss_csr_transpose``) and cached."""
        if self._transpose is None:
            tm = SparseMatrix.from_handle(self.engine.csr_transpose(self.handle), self.engine)
            tm._transpose = self
            self._transpose = tm
        return self._transpose

    @property
    def T(self):
        return self.CreateTranspose()

    def diagonal(self):
        rowptr, col, val = self.host_csr()
        d = np.zeros(min(self.m, self.n))
        rows = np.repeat(np.arange(self.m, dtype=np.int64), np.diff(rowptr))
        on = rows == col
        d[rows[on]] = val[on]
        return d

    # ---- smoothers (templates/NavierStokesSIMPLE_iterative.py:253,373) -------
    def CreateSmoother(self, freedofs=None):
        return JacobiPreconditioner(self, freedofs=freedofs)

    def CreateBlockSmoother(self, blocks):
        return BlockJacobi(self, blocks)


class DiagonalMatrix(BaseMatrix):
    """``y = diag(d) x`` with ``d`` resident in engine memory (``diag_scale_f64``)."""

    def __init__(self, d, engine=None):
        super().__init__()
        self.engine = engine if engine is not None else get_engine()
        d = np.ascontiguousarray(d, dtype=np.float64)
        self.n = d.size
        self.d_host = d
        self.d = self.engine.from_host(d)

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def Mult(self, x, y):
        self.engine.diag_apply(self.d, 1.0, x.buf, 0.0, y.buf)

    def MultAdd(self, s, x, y):
        self.engine.diag_apply(self.d, float(s), x.buf, 1.0, y.buf)

    MultTrans = Mult
    MultTransAdd = MultAdd

    @property
    def T(self):
        return self


class JacobiPreconditioner(DiagonalMatrix):
    """Point Jacobi ``y_i = x_i / A_ii`` on the free dofs -- what
    ``Preconditioner(mass, 'local')`` is for the pressure mass matrix (reference
    call sites: templates/NavierStokesSIMPLE_iterative.py:197-200, run.py:62,
    stokes_hcurldiv.py:50-53)."""

    def __init__(self, mat, freedofs=None):
        diag = mat.diagonal()
        if np.any(diag == 0.0):
            if freedofs is None:
                raise ZeroDivisionError("zero diagonal entry in Jacobi preconditioner")
        inv = np.zeros_like(diag)
        mask = np.ones(diag.size, dtype=bool) if freedofs is None else np.asarray(freedofs, dtype=bool)
        inv[mask] = 1.0 / diag[mask]
        super().__init__(inv, engine=mat.engine)
        self.mat = mat


class BlockJacobi(BaseMatrix):
    """Additive block-Jacobi ``J = sum_b E_b A_bb^{-1} E_b^T`` over disjoint dof
    blocks -- the smoother ``a.mat.CreateBlockSmoother(blocks)`` used as an operator
    (templates/NavierStokesSIMPLE_iterative.py:360-373,383).  Dense ``bs x bs``
    inverses are computed once in the engine and stored block-interleaved in HBM;
    apply is ``block_jacobi_apply_f64``.  Dofs in no block map to zero.

    The multiplicative sweeps ``Smooth``/``SmoothBack`` (GS=True, :376-381) live in
    `BlockGaussSeidel` (multicolour ordering, scope row N1 of SURVEY.md section 8f)."""

    def __init__(self, mat, blocks):
        super().__init__()
        self.engine = mat.engine
        self.mat = mat
        self.n = mat.height
        idx = self._as_table(blocks)
        self.bs, self.nblocks = idx.shape
        if self.nblocks == 0:
            raise ValueError("BlockJacobi needs at least one non-empty block")
        if self.bs > 16:
            raise ValueError("block size %d > 16 not supported by block_jacobi_apply_f64" % self.bs)
        flat = idx[idx >= 0]
        if flat.size and (flat.max() >= self.n or np.unique(flat).size != flat.size):
            raise ValueError("BlockJacobi blocks must be disjoint and in range")
        self.idx_host = idx
        self.handle = self.engine.bjac_create(mat.handle, idx)

    @staticmethod
    def _as_table(blocks):
        """(bs, nblocks) int32 table, -1 = padding, from a list of dof lists or a prebuilt table
        (staggered_grid.facet_blocks / line_blocks)."""
        if isinstance(blocks, np.ndarray) and blocks.ndim == 2:
            return np.ascontiguousarray(blocks, dtype=np.int32)
        blocks = [np.asarray(b, dtype=np.int64).ravel() for b in blocks]
        blocks = [b for b in blocks if b.size]
        bs = max(b.size for b in blocks) if blocks else 0
        idx = -np.ones((bs, len(blocks)), dtype=np.int32)
        for k, b in enumerate(blocks):
            idx[: b.size, k] = b
        return idx

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def Mult(self, x, y):
        self.engine.bjac_apply(self.handle, 1.0, x.buf, 0.0, y.buf)

    def MultAdd(self, s, x, y):
        self.engine.bjac_apply(self.handle, float(s), x.buf, 1.0, y.buf)

    MultTrans = Mult          # A_bb symmetric on this path
    MultTransAdd = MultAdd

    @property
    def T(self):
        return self

    # ---- multiplicative sweeps (templates/NavierStokesSIMPLE_iterative.py:376-381) -----------
    def gauss_seidel(self):
        """The companion handle in Gauss-Seidel mode: same blocks in a multicolour ordering
        (hipla/coloring.py), built on first use."""
        if getattr(self, "_gs", None) is None:
            self._gs = BlockGaussSeidel(self.mat, self.idx_host)
        return self._gs

    def Smooth(self, y, x):
        """Forward block Gauss-Seidel sweep for ``mat * y = x`` starting from the given y."""
        self.gauss_seidel().Smooth(y, x)

    def SmoothBack(self, y, x):
        self.gauss_seidel().SmoothBack(y, x)


class BlockGaussSeidel(BaseMatrix):
    """Symmetric multiplicative block Gauss-Seidel as an operator: ``y = 0; Smooth(y, x);
    SmoothBack(y, x)`` -- the reference's ``MypreA`` with ``GS=True``
    (templates/NavierStokesSIMPLE_iterative.py:376-381) without its auxiliary-space AMG term
    (scope row N3).  SURVEY.md section 8f row N1: the sweep runs over a *multicolour* block
    ordering so that all blocks of a colour update in one kernel launch
    (``nss_bjac_smooth_f64``); the ordering differs from NGSolve's mesh-facet order (upstream,
    not visible), so iteration counts are pinned against the build's own CPU oracle only."""

    def __init__(self, mat, blocks, seed=0, colors=None, middle=None, layout=None, coloring_method=None):
        """`colors` (one int per block) may be supplied when the system has already been
        re-ordered colour-major on the host (`coloring.colour_permutation`): the sweep then
        touches x, y and the inverse blocks contiguously.

        `layout`: "colour-major" (default on the HIP engine; NSS_GS_LAYOUT overrides) -- inside the sweep the
        iterate, the right-hand side and the matrix (P A P^T) live in the colour-major block numbering: x and y are
        gathered once on entry and y scattered once on exit, every colour is one launch (block solve in the SpMV's
        epilogue) over contiguous data; "rows" -- round 1's form: rows of A permuted, columns and vectors in the
        original numbering, two launches per colour.  Same bits for the same colours.
        `coloring_method`: "greedy" (default; first fit in block order on the host: the parity colouring of grid-like
        block graphs, 2 - 4 balanced colours) or "luby" (maximal independent sets on the device: 5 - 6 colours with a
        tail of tiny ones).  NSS_GS_COLORING overrides.

        `middle` (an operator M, e.g. the auxiliary-space term ``T @ AMG @ T.T``) reproduces the
        full ``MypreA.Mult`` of the reference with ``GS=True`` (:376-381):
        ``y = 0; Smooth(y, x); r = x - A y; y += M r; SmoothBack(y, x)``."""
        super().__init__()
        from . import coloring
        self.engine = mat.engine
        self.mat = mat
        self.middle = middle
        self.n = mat.height
        import os
        base = BlockJacobi._as_table(blocks)
        on_device = colors is None and hasattr(self.engine, "graph_color")
        layout = layout or os.environ.get("NSS_GS_LAYOUT", "colour-major")
        method = coloring_method or os.environ.get("NSS_GS_COLORING", "greedy")
        if layout not in ("colour-major", "rows") or method not in ("greedy", "luby"):
            raise ValueError("layout: 'colour-major' | 'rows'; coloring_method: 'greedy' | 'luby'")
        if not hasattr(self.engine, "csr_permute"):
            layout = "rows"
        self.layout, self.coloring_method = layout, (method if colors is None else "given")
        if on_device:
            # block graph (two sparse products of patterns) on the GPU, then first-fit colours in block order (host) or
            # Luby rounds (device): proper colourings by construction
            colors = self._device_colors(mat, base, seed, method)
        else:
            graph = coloring.block_graph(mat.to_scipy(), base)
            if colors is None:
                colors = coloring.color_blocks_greedy(graph) if method == "greedy" else coloring.color_blocks(graph, seed)
            colors = np.asarray(colors, dtype=np.int32)
            if colors.shape != (base.shape[1],) or not coloring.check_coloring(graph, colors):
                raise RuntimeError("block colouring is not proper")
        order, ptr = coloring.colour_major_order(colors)
        self.colors = np.asarray(colors, dtype=np.int32)            # per block, in the caller's block order
        self.idx_host = np.ascontiguousarray(base[:, order])
        self.color_ptr = ptr
        self.ncolors = int(ptr.size - 1)
        self.bs, self.nblocks = self.idx_host.shape
        self.handle = self.engine.bjac_create(mat.handle, self.idx_host)
        # rows of A re-ordered block by block in colour-major order (columns unchanged): the
        # residual of one colour becomes a streaming SpMV over a contiguous row range
        live = self.idx_host.T >= 0                                  # (nblocks, bs), block-major
        rowdof = self.idx_host.T[live].astype(np.int64)              # original dof of permuted row r
        ridx = -np.ones(self.idx_host.T.shape, dtype=np.int32)
        ridx[live] = np.arange(rowdof.size, dtype=np.int32)
        rows_per_block = live.sum(axis=1)
        block_row0 = np.concatenate([[0], np.cumsum(rows_per_block)])
        color_rowptr = block_row0[ptr]
        if layout == "colour-major":
            # P A P^T: columns renamed too (dofs outside every block -> the extra column n_perm, which stays 0); row
            # blocks of at most 256 rows that hold whole Gauss-Seidel blocks
            n_perm = int(rowdof.size)
            colmap = np.full(self.n, n_perm, dtype=np.int32)
            colmap[rowdof] = np.arange(n_perm, dtype=np.int32)
            pos = (np.arange(n_perm, dtype=np.int64) - np.repeat(block_row0[:-1], rows_per_block)).astype(np.uint8)
            self.perm_handle = self.engine.csr_permute(mat.handle, rowdof, colmap, n_perm + 1, cuts=color_rowptr,
                                                       max_rows=256, row_pos=pos)
            self.engine.bjac_set_colors_permuted(self.handle, self.perm_handle, ptr, color_rowptr, rowdof,
                                                 np.ascontiguousarray(ridx.T))
            return
        if hasattr(self.engine, "csr_select_rows"):
            self.perm_handle = self.engine.csr_select_rows(mat.handle, rowdof, cuts=color_rowptr)
        else:
            perm = mat.to_scipy()[rowdof]
            perm.sort_indices()
            self.perm_handle = self.engine.csr_create(perm.shape[0], perm.shape[1], perm.indptr, perm.indices,
                                                      perm.data, cuts=color_rowptr)
        self.engine.bjac_set_colors(self.handle, self.perm_handle, ptr, color_rowptr, rowdof,
                                    np.ascontiguousarray(ridx.T))

    @staticmethod
    def _device_colors(mat, base, seed, method="luby"):
        """Colours of the blocks `base` (bs x nblocks table): adjacency M^T |A| M of the blocks by two
        device SpGEMMs over 0/1 patterns (no cancellation), then `nss_graph_color` with the priorities
        `coloring.color_blocks` would use -- the result equals the host colouring."""
        import scipy.sparse as sp
        eng = mat.engine
        n, nb = mat.height, base.shape[1]
        live = base >= 0
        dofs = base[live]
        blocks = np.broadcast_to(np.arange(nb, dtype=np.int64), base.shape)[live]
        member = SparseMatrix.from_scipy(sp.csr_matrix((np.ones(dofs.size), (dofs, blocks)), shape=(n, nb)), engine=eng)
        pattern = eng.csr_ones_like(mat.handle)
        graph = SparseMatrix.from_handle(
            eng.csr_spgemm(member.CreateTranspose().handle, eng.csr_spgemm(pattern, member.handle)), eng)
        if method == "greedy" and hasattr(eng, "graph_color_greedy"):
            colors, _ = eng.graph_color_greedy(graph.handle, graph.CreateTranspose().handle)
        else:
            priority = np.random.default_rng(seed).permutation(nb).astype(np.int64) + 1
            colors, _ = eng.graph_color(graph.handle, graph.CreateTranspose().handle, priority)
        if hasattr(eng, "scratch_trim"):
            eng.scratch_trim()
        return colors

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def Smooth(self, y, x):
        self.engine.bjac_smooth(self.handle, 1.0, x.buf, y.buf, False)

    def SmoothBack(self, y, x):
        self.engine.bjac_smooth(self.handle, 1.0, x.buf, y.buf, True)

    def Mult(self, x, y):
        if self.middle is None:
            self.engine.bjac_apply(self.handle, 1.0, x.buf, 0.0, y.buf)  # GS-mode handle: symmetric sweep
            return
        y[:] = 0.0                                                       # :377
        self.Smooth(y, x)                                                # :378
        res = x.CreateVector()
        res.data = x - self.mat * y                                      # :379
        y.data += self.middle * res                                      # :380
        self.SmoothBack(y, x)                                            # :381

    MultTrans = Mult          # forward then backward sweep (symmetric middle term): symmetric operator

    @property
    def T(self):
        return self


class CGSolver(BaseMatrix):
    """``CGSolver(mat, pre, precision, maxsteps, printrates)`` as an operator ``y = mat^-1 x`` --
    the NGSolve class the reference uses for its inner solves
    (templates/NavierStokesSIMPLE_iterative.py:92 ``invmstar1``, :130 ``invproj1``).  Preconditioned
    conjugate gradients through the protocol (SpMV / lincomb / dot kernels); stops when
    ``sqrt(<r, pre r>)`` has dropped by ``precision`` relative to its initial value."""

    def __init__(self, mat, pre=None, precision=1e-8, maxsteps=200, printrates=False, inner=None):
        super().__init__()
        from .vector import InnerProduct
        self.mat, self.pre = mat, pre
        self.precision, self.maxsteps, self.printrates = float(precision), int(maxsteps), printrates
        self.inner = inner if inner is not None else InnerProduct
        self.iterations = 0
        self.errors = []
        self._fused = False           # device-resident loop (nss_cg_*), created on first use

    def Height(self):
        return self.mat.height

    def Width(self):
        return self.mat.width

    def Mult(self, x, y):
        from math import sqrt
        from .vector import InnerProduct, Vector
        if self._fused is False:
            from .fused import CgLoop
            self._fused = CgLoop.try_create(self.mat, self.pre) if self.inner is InnerProduct else None
        if self._fused is not None and isinstance(x, Vector) and isinstance(y, Vector) and not self.printrates:
            self.iterations, self.errors = self._fused.solve(x.buf, y.buf, self.precision, self.maxsteps)
            return
        dot = self.inner
        r, z, p, q = (y.CreateVector() for _ in range(4))
        y[:] = 0.0
        r.data = x
        z.data = self.pre * r if self.pre is not None else r
        p.data = z
        rz = dot(r, z)
        err0 = sqrt(abs(rz))
        self.errors = [err0]
        self.iterations = 0
        if err0 == 0.0:
            return
        for it in range(self.maxsteps):
            q.data = self.mat * p
            alpha = rz / dot(p, q)
            y.data += alpha * p
            r.data -= alpha * q
            z.data = self.pre * r if self.pre is not None else r
            rz_new = dot(r, z)
            err = sqrt(abs(rz_new))
            self.errors.append(err)
            self.iterations = it + 1
            if self.printrates:
                print("it = ", it, " err = ", err)
            if err < self.precision * err0:
                break
            p.data = z + (rz_new / rz) * p
            rz = rz_new

    MultTrans = Mult          # mat symmetric on this path

    @property
    def T(self):
        return self


class Embedding(BaseMatrix):
    """``Embedding(height, range)``: places a vector of ``len(range)`` entries at rows
    ``range.start .. range.stop`` of a zero vector of ``height`` entries; ``.T`` restricts.  The reference
    stacks the per-component auxiliary preconditioners with it
    (templates/NavierStokesSIMPLE_iterative.py:334-337,353-357)."""

    def __init__(self, height, rng):
        super().__init__()
        self.h = int(height)
        self.start, self.stop = int(rng.start), int(rng.stop)
        if not 0 <= self.start <= self.stop <= self.h:
            raise ValueError("Embedding: range outside the target")

    def Height(self):
        return self.h

    def Width(self):
        return self.stop - self.start

    def _piece(self, v):
        from .vector import Vector
        return Vector(buf=v.engine.view(v.buf, self.start, self.stop), engine=v.engine, comm=getattr(v, "comm", None))

    def MultAdd(self, s, x, y):
        self._piece(y).data += s * x

    def Mult(self, x, y):
        y[:] = 0.0
        self._piece(y).data = x

    def MultTransAdd(self, s, x, y):
        y.data += s * self._piece(x)

    def MultTrans(self, x, y):
        y.data = self._piece(x)


class Projector(BaseMatrix):
    """``Projector(mask, range)``: keeps entries where ``mask == range``."""

    def __init__(self, mask, range=True, engine=None):
        super().__init__()
        m = np.asarray(mask, dtype=bool)
        self._diag = DiagonalMatrix((m == bool(range)).astype(np.float64), engine=engine)

    def Height(self):
        return self._diag.n

    def Width(self):
        return self._diag.n

    def Mult(self, x, y):
        self._diag.Mult(x, y)

    def MultAdd(self, s, x, y):
        self._diag.MultAdd(s, x, y)

    MultTrans = Mult
    MultTransAdd = MultAdd


_WARNED = {}


def Preconditioner(form, kind, blocks=None, **_):
    """``Preconditioner(blf, 'local')`` -> point Jacobi; ``'blockjacobi'`` -> additive block Jacobi
    over `blocks`; ``'h1amg'`` / ``'multigrid'`` -> the smoothed-aggregation V-cycle on the assembled
    matrix (`hipla.amg`).  ``'bddc'`` (stokes_hcurldiv.py:48, templates/...iterative.py:77,88,122,306)
    is NGSolve's domain-decomposition preconditioner built on its FE spaces; it has no algebraic
    counterpart here, and the same V-cycle is returned in its place -- a mesh-independent SPD
    preconditioner for the same operator, with different iteration counts than the reference's."""
    mat = form.mat if hasattr(form, "mat") else form
    if kind == "local":
        if blocks is not None:
            return BlockJacobi(mat, blocks)
        return JacobiPreconditioner(mat)
    if kind in ("blockjacobi", "block_jacobi"):
        if blocks is None:
            raise ValueError("blockjacobi needs blocks=")
        return BlockJacobi(mat, blocks)
    if kind in ("h1amg", "multigrid", "bddc"):
        from .amg import SmoothedAggregationAMG
        if kind == "bddc" and not _WARNED.get(kind):
            _WARNED[kind] = True
            import warnings
            warnings.warn("Preconditioner(..., 'bddc'): NGSolve's BDDC needs its FE spaces; a smoothed-aggregation "
                          "V-cycle on the assembled matrix is used in its place (different iteration counts than "
                          "the reference's)", stacklevel=2)
        return SmoothedAggregationAMG(mat)
    raise NotImplementedError("preconditioner %r is outside the hot-path scope (SURVEY.md section 8f)" % (kind,))
