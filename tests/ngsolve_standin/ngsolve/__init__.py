"""TEST-ONLY stand-in for the ``ngsolve`` import surface of the reference's solver
modules.

It re-exports the product's own protocol layer (``hipla``) so that the reference's
*unmodified* files ``minres.py``, ``bramble_pasciak_cg.py`` and
``solvers/bramblepasciak_new.py`` can be imported from ``/root/reference`` in the
build container and run over it with the numpy checker engine
(``oracle/numpy_engine.py``) to produce golden vectors
(``tests/golden/make_golden.py``).  Never imported by the product.
"""
from hipla import *          # noqa: F401,F403
from hipla import la, ngstd  # noqa: F401
