"""lowrankmodels.jl_amd -- MI355X-native engine for LowRankModels.jl's proximal-gradient ``fit!``.

Host-side mirror of the reference interface for that one path (GLRM / losses / regularizers /
ProxGradParams / fit! / ConvergenceHistory) over the C ABI of ``libglrm_hip.so``
(include/glrm_hip.h).  See DESIGN.md for scope and INTEGRATION.md for the Julia binding.
"""
from . import _capi
from ._capi import GLRMError
from .convergence import ConvergenceHistory, update_ch
from .fit import ShardedFit, fit, fit_b, objective, partition
from .crossval import (cross_validate, cv_by_iter, flatten_observations, get_train_and_test, getfolds, loss_fn,
                             regularization_path)
from .initialize import init_svd_
from .domains import (BoolDomain, CategoricalDomain, CountDomain, Domain, OrdinalDomain, PeriodicDomain, RealDomain, default_domain,
                      error_metric, error_metric_entry, impute, impute_entry, impute_missing)
from .glrm import GLRM, add_offset_, copy_estimate, parameter_estimate, scale_regularizer_, sort_observations
from .losses import (BvSLoss, HingeLoss, HuberLoss, L1Loss, LogisticLoss, Loss, MultinomialLoss, MultinomialOrdinalLoss,
                     OrdinalHingeLoss, OrdisticLoss, OvALoss, PeriodicLoss, PoissonLoss, QuadLoss, QuantileLoss,
                     WeightedHingeLoss, embedding_dim, evaluate, get_yidxs, grad)
from .params import AbstractParams, HipProxGradParams, Params, ProxGradParams, SparseProxGradParams
from .regularizers import (MNLOrdinalReg, NonNegConstraint, OneReg, OrdinalReg, QuadReg, Regularizer, UnitOneSparseConstraint,
                           ZeroReg, lastentry1, lastentry_unpenalized, prox)

fit_inplace = fit_b  # Julia's `fit!`

__all__ = [n for n in dir() if not n.startswith("_")]
