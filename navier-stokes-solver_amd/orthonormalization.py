"""Repeated modified Gram-Schmidt with the reference's signature (orthonormalization.py:5-16):
``orthonormalize(basis, tries=3)`` over protocol vectors -- InnerProduct, Norm and AXPY kernels
only (the plumbing operations of BASELINE.json config 1)."""

from hipla import InnerProduct, Norm
from hipla.ngstd import Timer


def orthonormalize(basis, tries=3):
    timer = Timer("orthonormalization")
    timer.Start()
    for _ in range(tries):
        for j, bj in enumerate(basis):
            for bi in basis[:j]:
                bj.data -= InnerProduct(bi, bj) / InnerProduct(bi, bi) * bi
            bj.data = 1 / Norm(bj) * bj
    timer.Stop()
    return basis
