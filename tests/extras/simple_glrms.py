"""TEST-SIDE helper (out of scope for the product): pca / qpca / nnmf / kmeans / rpca model builders (reference: src/simple_glrms.jl:1-42): five-line constructors on top of GLRM."""
from __future__ import annotations

from lowrankmodels.jl_amd.glrm import GLRM
from lowrankmodels.jl_amd.losses import HuberLoss, QuadLoss
from lowrankmodels.jl_amd.regularizers import NonNegConstraint, QuadReg, UnitOneSparseConstraint, ZeroReg


def pca(A, k, **kwargs):
    """minimize ||A - XY||^2"""
    return GLRM(A, QuadLoss(), ZeroReg(), ZeroReg(), k, **kwargs)


def qpca(A, k, scale=1.0, **kwargs):
    """minimize ||A - XY||^2 + scale*||X||^2 + scale*||Y||^2"""
    return GLRM(A, QuadLoss(), QuadReg(scale), QuadReg(scale), k, **kwargs)


def nnmf(A, k, **kwargs):
    """minimize_{X>=0, Y>=0} ||A - XY||^2"""
    return GLRM(A, QuadLoss(), NonNegConstraint(), NonNegConstraint(), k, **kwargs)


def kmeans(A, k, **kwargs):
    """minimize_{columns of X are unit vectors} ||A - XY||^2"""
    return GLRM(A, QuadLoss(), UnitOneSparseConstraint(), ZeroReg(), k, **kwargs)


def rpca(A, k, scale=1.0, **kwargs):
    """minimize HuberLoss(A - XY) + scale*||X||^2 + scale*||Y||^2"""
    return GLRM(A, HuberLoss(), QuadReg(scale), QuadReg(scale), k, **kwargs)
