"""``ngsolve.la`` of the numpy-only test stand-in (see the package docstring)."""
from . import (BaseMatrix, BaseVector, BlockMatrix, BlockVector, EigenValues_Preconditioner,  # noqa: F401
               IdentityMatrix, InnerProduct, Norm, Projector, SparseMatrix, Vector)

__all__ = ["BaseMatrix", "BaseVector", "BlockMatrix", "BlockVector", "IdentityMatrix", "InnerProduct", "Norm",
           "Projector", "SparseMatrix", "Vector"]
