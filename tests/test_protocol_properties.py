"""Property tests of the operator protocol's expression evaluation (host logic, numpy checker
engine): random linear combinations of vectors and operator applications assigned / accumulated
into possibly aliased destinations must equal the numpy evaluation; block vectors behave
component-wise; user BaseMatrix subclasses are called back with the documented defaults."""

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st


@st.composite
def combos(draw):
    n = draw(st.integers(1, 40))
    nv = draw(st.integers(1, 4))
    terms = draw(st.lists(st.tuples(st.floats(-3, 3, allow_nan=False, width=32), st.integers(0, nv - 1),
                                    st.booleans()), min_size=1, max_size=6))
    dest = draw(st.integers(0, nv - 1))
    accumulate = draw(st.sampled_from(["=", "+=", "-="]))
    seed = draw(st.integers(0, 2 ** 16))
    return n, nv, terms, dest, accumulate, seed


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(combos())
def test_random_expressions_match_numpy(numpy_engine, case):
    import hipla
    import scipy.sparse as sp
    n, nv, terms, dest, mode, seed = case
    rng = np.random.default_rng(seed)
    host = [rng.standard_normal(n) for _ in range(nv)]
    vecs = [hipla.Vector.from_numpy(h) for h in host]
    m = sp.random(n, n, density=0.3, random_state=seed, format="csr") + sp.identity(n)
    M = hipla.SparseMatrix.from_scipy(m.tocsr())
    expr, ref = None, np.zeros(n)
    for scale, k, with_mat in terms:
        piece = scale * (M * vecs[k]) if with_mat else scale * vecs[k]
        expr = piece if expr is None else expr + piece
        ref = ref + scale * ((m @ host[k]) if with_mat else host[k])
    if mode == "=":
        vecs[dest].data = expr
        want = ref
    elif mode == "+=":
        vecs[dest].data += expr
        want = host[dest] + ref
    else:
        vecs[dest].data -= expr
        want = host[dest] - ref
    np.testing.assert_allclose(vecs[dest].numpy(), want, rtol=1e-10, atol=1e-10)
    for k in range(nv):                                   # operands other than the destination are untouched
        if k != dest:
            np.testing.assert_array_equal(vecs[k].numpy(), host[k])


def test_block_vectors_and_subclass_defaults(numpy_engine):
    import hipla
    rng = np.random.default_rng(0)
    a = [rng.standard_normal(5), rng.standard_normal(3)]
    b = [rng.standard_normal(5), rng.standard_normal(3)]
    A = hipla.BlockVector([hipla.Vector.from_numpy(x) for x in a])
    B = hipla.BlockVector([hipla.Vector.from_numpy(x) for x in b])
    C = A.CreateVector()
    C.data = 2.0 * A - B
    np.testing.assert_allclose(C.numpy(), np.concatenate([2 * a[0] - b[0], 2 * a[1] - b[1]]))
    C *= 0.5
    C.data += A
    np.testing.assert_allclose(C[1].numpy(), 0.5 * (2 * a[1] - b[1]) + a[1])
    assert abs(hipla.InnerProduct(A, B) - (a[0] @ b[0] + a[1] @ b[1])) < 1e-12
    assert len(A) == 8 and bool(A) and C[0] is C.components[0]
    C[:] = 0
    assert not C.numpy().any()
    with pytest.raises(TypeError):
        C.data = A[0] + A[0]                               # plain expression into a block vector

    calls = []

    class OnlyMult(hipla.BaseMatrix):
        def Mult(self, x, y):
            calls.append("Mult")
            y.data = 3.0 * x

        def Height(self):
            return 5

        def Width(self):
            return 5

    class OnlyMultAdd(hipla.BaseMatrix):
        def MultAdd(self, s, x, y):
            calls.append("MultAdd")
            y.data += (2.0 * s) * x

        def Height(self):
            return 5

        def Width(self):
            return 5

    x = hipla.Vector.from_numpy(a[0])
    y = hipla.Vector(5)
    y.data = OnlyMult() * x                                # Mult straight into the destination
    y.data += 2.0 * OnlyMult() * x                         # default MultAdd: Mult into a temp, then axpy
    np.testing.assert_allclose(y.numpy(), 9.0 * a[0])
    y.data = OnlyMultAdd() * x                             # default Mult: zero + MultAdd(1, ...)
    y.data -= OnlyMultAdd() * x
    np.testing.assert_allclose(y.numpy(), 0.0 * a[0], atol=1e-15)
    assert calls == ["Mult", "Mult", "MultAdd", "MultAdd"]
    t = (OnlyMult() @ OnlyMultAdd()).T
    with pytest.raises(NotImplementedError):
        y.data = t * x                                     # no MultTrans on the user classes
    y.data = (hipla.IdentityMatrix(5) - 0.5 * OnlyMult()) * x
    np.testing.assert_allclose(y.numpy(), -0.5 * a[0])
    y.data = x
    y.data = OnlyMult() * y                                # operand aliases the destination: temp inserted
    np.testing.assert_allclose(y.numpy(), 3.0 * a[0])
