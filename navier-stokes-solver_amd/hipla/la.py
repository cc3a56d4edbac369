"""``hipla.la`` -- same import surface as ``ngsolve.la`` for the solve path
(solvers/bramblepasciak_new.py:1-2, minres.py:3, bramble_pasciak_cg.py:6)."""

from .vector import BaseVector, BlockVector, Expr, InnerProduct, Norm, Vector
from .matrix import (BaseMatrix, BlockGaussSeidel, BlockJacobi, BlockMatrix, CGSolver, DiagonalMatrix, Embedding,
                     IdentityMatrix,
                     JacobiPreconditioner, Preconditioner, ProductMatrix, Projector,
                     ScaledMatrix, SparseMatrix, SumMatrix, TransposeMatrix)
from .amg import AuxiliarySpaceAMG, SmoothedAggregationAMG
from .eigen import EigenValues_Preconditioner, lanczos_ritz, lanczos_start_values

__all__ = ["BaseVector", "BlockVector", "Expr", "InnerProduct", "Norm", "Vector",
           "BaseMatrix", "BlockGaussSeidel", "BlockJacobi", "BlockMatrix", "CGSolver", "DiagonalMatrix", "IdentityMatrix",
           "JacobiPreconditioner", "Preconditioner", "ProductMatrix", "Projector",
           "ScaledMatrix", "SparseMatrix", "SumMatrix", "TransposeMatrix",
           "EigenValues_Preconditioner", "lanczos_ritz", "lanczos_start_values", "SmoothedAggregationAMG",
           "AuxiliarySpaceAMG", "Embedding"]
