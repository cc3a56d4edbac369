"""hipla -- the NGSolve-style linear-algebra protocol of the Stokes / SIMPLE solve
path, backed by hand-written gfx950 HIP kernels through a ctypes C ABI.

``from hipla import *`` exports what the reference's solver modules take from
``from ngsolve import *`` (minres.py:3-7, bramble_pasciak_cg.py:4-6,
solvers/bramblepasciak_new.py:1-5): BaseMatrix, BlockMatrix, BlockVector,
IdentityMatrix, Vector, InnerProduct, Norm, Projector; plus ``hipla.la`` and
``hipla.ngstd`` submodules shaped like ``ngsolve.la`` / ``ngsolve.ngstd``.
"""

from . import la, ngstd
from .engine import EngineUnavailable, get_engine, set_engine
from .la import *  # noqa: F401,F403
from .ngstd import SetHeapSize, TaskManager, Timer

__all__ = list(la.__all__) + ["la", "ngstd", "TaskManager", "Timer", "SetHeapSize",
                              "get_engine", "set_engine", "EngineUnavailable"]
