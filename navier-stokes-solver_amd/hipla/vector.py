"""Vectors and lazy linear-combination expressions of the hipla operator protocol.

This is the vector half of the drop-in boundary: the call
sites of the reference Krylov loops -- ``v.data = expr``, ``v.data += expr``,
``v[:] = c``, ``v *= c``, ``InnerProduct``, ``Norm``, ``CreateVector``, ``Copy`` --
(reference: minres.py:50-118, bramble_pasciak_cg.py:87-141,
solvers/bramblepasciak_new.py:124-241) keep working unchanged.  Storage lives in
an engine buffer (HBM for the HIP engine); every statement is evaluated *into the
destination* by engine kernels, never on the host.
"""

from numbers import Number

import numpy as np

from .engine import get_engine

_MAX_FUSED_TERMS = 4


def _is_scalar(x):
    return isinstance(x, (Number, np.floating, np.integer)) and not isinstance(x, bool)


class Term:
    """``scale * (mat @ vec)`` (mat may be None).  `vec` is a Vector or BlockVector."""

    __slots__ = ("scale", "mat", "vec")

    def __init__(self, scale, mat, vec):
        self.scale = float(scale)
        self.mat = mat
        self.vec = vec


class Expr:
    """Sum of terms; built by operator overloading, consumed by ``dest.data = expr``."""

    __slots__ = ("terms",)

    def __init__(self, terms):
        self.terms = list(terms)

    # ---- algebra -------------------------------------------------------
    def __add__(self, other):
        return Expr(self.terms + as_expr(other).terms)

    def __radd__(self, other):
        return Expr(as_expr(other).terms + self.terms)

    def __sub__(self, other):
        return Expr(self.terms + (-as_expr(other)).terms)

    def __rsub__(self, other):
        return Expr(as_expr(other).terms + (-self).terms)

    def __neg__(self):
        return Expr([Term(-t.scale, t.mat, t.vec) for t in self.terms])

    def __mul__(self, s):
        if not _is_scalar(s):
            return NotImplemented
        return Expr([Term(t.scale * float(s), t.mat, t.vec) for t in self.terms])

    __rmul__ = __mul__

    def __truediv__(self, s):
        return self * (1.0 / float(s))

    # ---- shape helpers ---------------------------------------------------
    def materialize(self, like=None):
        """Evaluate into a fresh vector (used for ``M * (a + b)``)."""
        t0 = self.terms[0]
        if like is None:
            like = t0.mat.CreateColVector() if t0.mat is not None else t0.vec.CreateVector()
        else:
            like = like.CreateVector()
        like.data = self
        return like


def as_expr(x):
    if isinstance(x, Expr):
        return x
    if isinstance(x, _DataProxy):
        return Expr([Term(1.0, None, x.owner)])
    if isinstance(x, (BaseVector,)):
        return Expr([Term(1.0, None, x)])
    raise TypeError("cannot use %r in a vector expression" % (type(x).__name__,))


class _DataProxy:
    """What ``v.data`` returns: supports ``v.data += e`` / ``v.data -= e``.

    Python expands ``v.data += e`` to ``tmp = v.data; tmp = tmp.__iadd__(e);
    v.data = tmp``; the setter recognises the proxy of the same vector and does
    nothing (the update already happened in place).
    """

    __slots__ = ("owner",)

    def __init__(self, owner):
        self.owner = owner

    def __iadd__(self, other):
        self.owner._accumulate(as_expr(other))
        return self

    def __isub__(self, other):
        self.owner._accumulate(-as_expr(other))
        return self


class BaseVector:
    """Operator sugar shared by Vector and BlockVector."""

    # -- lazy algebra ------------------------------------------------------
    def __add__(self, other):
        return as_expr(self) + other

    def __radd__(self, other):
        return as_expr(other) + self

    def __sub__(self, other):
        return as_expr(self) - other

    def __rsub__(self, other):
        return as_expr(other) - self

    def __neg__(self):
        return -as_expr(self)

    def __mul__(self, s):
        if _is_scalar(s):
            return as_expr(self) * s
        return NotImplemented

    def __rmul__(self, s):
        if _is_scalar(s):
            return as_expr(self) * s
        return NotImplemented

    def __truediv__(self, s):
        return as_expr(self) * (1.0 / float(s))

    # -- .data protocol ------------------------------------------------------
    @property
    def data(self):
        return _DataProxy(self)

    @data.setter
    def data(self, value):
        if isinstance(value, _DataProxy) and value.owner is self:
            return  # tail of ``self.data += ...``
        self._assign(as_expr(value))

    def Assign(self, other, s=1.0):
        self._assign(as_expr(other) * s)

    def Add(self, other, s=1.0):
        self._accumulate(as_expr(other) * s)

    @staticmethod
    def _merge_self_terms(terms, is_self):
        """Plain terms that alias the destination are consumed first, as ONE term carrying the
        sum of their coefficients: evaluating left to right would otherwise overwrite the
        destination before a later ``c * dest`` term reads it."""
        mine = [t for t in terms if t.mat is None and is_self(t.vec)]
        if not mine or (len(mine) == 1 and terms[0] is mine[0]):
            return terms
        rest = [t for t in terms if not (t.mat is None and is_self(t.vec))]
        return [Term(sum(t.scale for t in mine), None, mine[0].vec)] + rest

    def CreateColVector(self):
        return self.CreateVector()

    def CreateRowVector(self):
        return self.CreateVector()

    def Norm(self):
        from math import sqrt
        return sqrt(InnerProduct(self, self))

    def __bool__(self):
        return len(self) > 0


class Vector(BaseVector):
    """Plain fp64 vector in engine memory.  ``Vector(n)`` allocates n zeros."""

    def __init__(self, n=None, *, buf=None, engine=None, comm=None):
        """`comm`: set on the slab of a row-partitioned vector (distributed.py); inner products of
        such vectors are summed over the ranks, and vectors created from them inherit it."""
        self.engine = engine if engine is not None else get_engine()
        if buf is None:
            if n is None:
                raise TypeError("Vector(n) needs a size")
            buf = self.engine.zeros(int(n))
        self.buf = buf
        self.size = self.engine.length(buf)
        self.comm = comm

    # -- construction helpers ----------------------------------------------
    @classmethod
    def from_numpy(cls, arr, engine=None):
        eng = engine if engine is not None else get_engine()
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        return cls(buf=eng.from_host(arr), engine=eng)

    FromNumPy = from_numpy

    def numpy(self):
        """Host copy (synchronises the stream)."""
        return self.engine.to_host(self.buf)

    NumPy = numpy

    def set_from(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if arr.shape != (self.size,):
            raise ValueError("size mismatch: %r vs %d" % (arr.shape, self.size))
        self.engine.upload(arr, self.buf)

    def CreateVector(self):
        return Vector(self.size, engine=self.engine, comm=self.comm)

    def Copy(self):
        v = self.CreateVector()
        self.engine.copy(self.buf, v.buf)
        return v

    def __len__(self):
        return self.size

    # -- element / slice access ---------------------------------------------
    def __setitem__(self, key, value):
        if isinstance(key, slice) and _is_scalar(value):
            if key == slice(None, None, None):
                self.engine.fill(self.buf, float(value))
            else:
                self.engine.fill(self.engine.view(self.buf, *key.indices(self.size)[:2]), float(value))
            return
        if isinstance(key, (int, np.integer)) and _is_scalar(value):
            i = int(key) % self.size
            self.engine.fill(self.engine.view(self.buf, i, i + 1), float(value))
            return
        if isinstance(key, slice):
            sub = self[key]
            sub.data = value
            return
        raise TypeError("unsupported vector assignment")

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b, st = key.indices(self.size)
            if st != 1:
                raise IndexError("only contiguous ranges are supported")
            return Vector(buf=self.engine.view(self.buf, a, b), engine=self.engine, comm=self.comm)
        i = int(key)
        if i < 0:
            i += self.size
        if not 0 <= i < self.size:
            raise IndexError(key)
        return float(self.engine.to_host(self.engine.view(self.buf, i, i + 1))[0])

    def Range(self, a, b):
        return self[a:b]

    def __imul__(self, s):
        self.engine.scal(self.buf, float(s))
        return self

    def __itruediv__(self, s):
        self.engine.scal(self.buf, 1.0 / float(s))
        return self

    def __iadd__(self, other):
        self._accumulate(as_expr(other))
        return self

    def __isub__(self, other):
        self._accumulate(-as_expr(other))
        return self

    # -- evaluation ------------------------------------------------------------
    def _same_storage(self, other):
        return isinstance(other, Vector) and self.engine.same_buffer(self.buf, other.buf)

    def _prepare(self, expr):
        """Resolve aliasing of matvec operands with the destination."""
        terms = []
        for t in expr.terms:
            if t.mat is not None and self._overlaps(t.vec):
                tmp = self.CreateVector()
                t.mat.Mult(t.vec, tmp)
                terms.append(Term(t.scale, None, tmp))
            else:
                terms.append(t)
        return terms

    def _overlaps(self, vec):
        if isinstance(vec, Vector):
            return self.engine.overlaps(self.buf, vec.buf)
        if isinstance(vec, BlockVector):
            return any(self._overlaps(c) for c in vec.components)
        return False

    def _check_plain(self, t):
        if not isinstance(t.vec, Vector):
            raise TypeError("block operand assigned to a plain vector")
        if t.vec.size != self.size:
            raise ValueError("vector size mismatch: %d vs %d" % (t.vec.size, self.size))

    def _assign(self, expr):
        terms = self._merge_self_terms(self._prepare(expr), self._same_storage)
        eng = self.engine
        if all(t.mat is None for t in terms) and len(terms) <= _MAX_FUSED_TERMS:
            for t in terms:
                self._check_plain(t)
            eng.lincomb(self.buf, [(t.scale, t.vec.buf) for t in terms])
            return
        first, rest = terms[0], terms[1:]
        if first.mat is None:
            self._check_plain(first)
            eng.lincomb(self.buf, [(first.scale, first.vec.buf)])
        else:
            first.mat.Mult(first.vec, self)
            if first.scale != 1.0:
                eng.scal(self.buf, first.scale)
        self._add_terms(rest)

    def _accumulate(self, expr):
        terms = self._prepare(expr)
        if any(t.mat is None and self._same_storage(t.vec) for t in terms):
            # ``x += c * x + ...``: every c * x must see the old x -> evaluate as one assignment
            self._assign(Expr([Term(1.0, None, self)] + terms))
            return
        self._add_terms(terms)

    def _add_terms(self, terms):
        eng = self.engine
        plain = []

        def flush():
            while plain:
                chunk = plain[: _MAX_FUSED_TERMS - 1]
                del plain[: _MAX_FUSED_TERMS - 1]
                eng.lincomb(self.buf, [(1.0, self.buf)] + [(t.scale, t.vec.buf) for t in chunk])

        for t in terms:
            if t.mat is None:
                self._check_plain(t)
                plain.append(t)
            else:
                flush()
                t.mat.MultAdd(t.scale, t.vec, self)
        flush()

    def __repr__(self):
        return "Vector(size=%d, engine=%s)" % (self.size, self.engine.name)


class BlockVector(BaseVector):
    """List of separately allocated components; ``bv[i]`` returns the component by
    reference (reference use: run.py:33, templates/NavierStokesSIMPLE_iterative.py:206)."""

    def __init__(self, components):
        self.components = list(components)
        if not self.components:
            raise ValueError("BlockVector needs at least one component")

    @property
    def nblocks(self):
        return len(self.components)

    @property
    def engine(self):
        return self.components[0].engine

    @property
    def comm(self):
        return getattr(self.components[0], "comm", None)

    @property
    def size(self):
        return sum(len(c) for c in self.components)

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        if isinstance(i, slice):
            raise TypeError("BlockVector supports component access bv[i] and bv[:] = c only")
        return self.components[i]

    def __setitem__(self, key, value):
        if isinstance(key, slice) and key == slice(None, None, None) and _is_scalar(value):
            for c in self.components:
                c[:] = value
            return
        if isinstance(key, (int, np.integer)):
            self.components[key].data = value
            return
        raise TypeError("unsupported block-vector assignment")

    def CreateVector(self):
        return BlockVector([c.CreateVector() for c in self.components])

    def Copy(self):
        return BlockVector([c.Copy() for c in self.components])

    def numpy(self):
        return np.concatenate([c.numpy() for c in self.components])

    NumPy = numpy

    def __imul__(self, s):
        for c in self.components:
            c *= s
        return self

    def __itruediv__(self, s):
        return self.__imul__(1.0 / float(s))

    def __iadd__(self, other):
        self._accumulate(as_expr(other))
        return self

    def __isub__(self, other):
        self._accumulate(-as_expr(other))
        return self

    def _overlaps(self, vec):
        return any(c._overlaps(vec) for c in self.components)

    def _component_terms(self, t, i):
        if isinstance(t.vec, BlockVector):
            if t.vec.nblocks != self.nblocks:
                raise ValueError("block layout mismatch")
            return Term(t.scale, None, t.vec.components[i])
        raise TypeError("plain operand assigned to a block vector")

    def _prepare(self, expr):
        terms = []
        for t in expr.terms:
            if t.mat is not None and self._overlaps(t.vec):
                tmp = self.CreateVector()
                t.mat.Mult(t.vec, tmp)
                terms.append(Term(t.scale, None, tmp))
            else:
                terms.append(t)
        return terms

    def _assign(self, expr):
        terms = self._merge_self_terms(self._prepare(expr), lambda v: v is self)
        if all(t.mat is None for t in terms):
            for i, c in enumerate(self.components):
                c._assign(Expr([self._component_terms(t, i) for t in terms]))
            return
        first, rest = terms[0], terms[1:]
        if first.mat is None:
            for i, c in enumerate(self.components):
                c._assign(Expr([self._component_terms(first, i)]))
        else:
            first.mat.Mult(first.vec, self)
            if first.scale != 1.0:
                self *= first.scale
        self._add_terms(rest)

    def _accumulate(self, expr):
        terms = self._prepare(expr)
        if any(t.mat is None and t.vec is self for t in terms):
            self._assign(Expr([Term(1.0, None, self)] + terms))
            return
        self._add_terms(terms)

    def _add_terms(self, terms):
        run = []

        def flush():
            if run:
                for i, c in enumerate(self.components):
                    c._add_terms([self._component_terms(t, i) for t in run])
                del run[:]

        for t in terms:
            if t.mat is None:
                run.append(t)
            else:
                flush()
                t.mat.MultAdd(t.scale, t.vec, self)
        flush()

    def __repr__(self):
        return "BlockVector(%s)" % ", ".join(str(len(c)) for c in self.components)


def InnerProduct(a, b):
    """Euclidean inner product -> Python float (reference: minres.py:71,98,103;
    bramble_pasciak_cg.py:105,130,137; solvers/bramblepasciak_new.py:185,222,235).
    Block vectors: sum of the component dots, component 0 first.  Slabs of row-partitioned vectors
    (``comm`` set): the local value is summed over the ranks (one all-reduce of a double)."""
    local = _local_inner(a, b)
    comm = getattr(a, "comm", None)
    if comm is not None and comm.size > 1:
        return comm.allreduce_scalar(local)
    return local


def _local_inner(a, b):
    if isinstance(a, BlockVector) or isinstance(b, BlockVector):
        if not (isinstance(a, BlockVector) and isinstance(b, BlockVector)) or a.nblocks != b.nblocks:
            raise TypeError("InnerProduct of mismatching block layouts")
        eng = a.engine
        if hasattr(eng, "dot_multi"):
            return eng.dot_multi([(x.buf, y.buf) for x, y in zip(a.components, b.components)])
        total = 0.0
        for x, y in zip(a.components, b.components):
            total += _local_inner(x, y)
        return total
    if a.size != b.size:
        raise ValueError("InnerProduct size mismatch: %d vs %d" % (a.size, b.size))
    return float(a.engine.dot(a.buf, b.buf))


def Norm(v):
    from math import sqrt
    return sqrt(InnerProduct(v, v))
