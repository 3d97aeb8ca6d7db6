# HipGLRMDescriptors.jl -- included by HipGLRM.jl: the reference's loss / regularizer TYPES as the descriptor tables of
# include/glrm_hip.h (glrm_loss: kind, dim, scale, p0, p1; glrm_reg: kind, wrap, scale), and the test "does the engine take this
# model?" (anything else falls back to the reference solver: src/algorithms/proxgrad.jl).  Pure table look-ups.
import LowRankModels: Loss, Regularizer, QuadLoss, L1Loss, HuberLoss, QuantileLoss, PeriodicLoss, PoissonLoss, OrdinalHingeLoss,
                      LogisticLoss, WeightedHingeLoss, MultinomialLoss, OvALoss, BvSLoss, OrdisticLoss, MultinomialOrdinalLoss,
                      embedding_dim, ZeroReg, QuadReg, OneReg, NonNegConstraint, UnitOneSparseConstraint,
                      lastentry1, lastentry_unpenalized, OrdinalReg, MNLOrdinalReg

closs(l::QuadLoss) = CLoss(0, 0, l.scale, 0, 0)                               # src/losses.jl:138-148
closs(l::L1Loss) = CLoss(1, 0, l.scale, 0, 0)                                 # :152-162
closs(l::HuberLoss) = CLoss(2, 0, l.scale, l.crossover, 0)                    # :166-179
closs(l::QuantileLoss) = CLoss(3, 0, l.scale, l.quantile, 0)                  # :186-203
closs(l::PeriodicLoss) = CLoss(4, 0, l.scale, l.T, 0)                         # :209-224
closs(l::PoissonLoss) = CLoss(5, 0, l.scale, 0, 0)                            # :231-243
closs(l::OrdinalHingeLoss) = CLoss(6, 0, l.scale, l.min, l.max)               # :247-294
closs(l::LogisticLoss) = CLoss(7, 0, l.scale, 0, 0)                           # :298-311
closs(l::WeightedHingeLoss) = CLoss(8, 0, l.scale, l.case_weight_ratio, 0)    # :317-352
# multi-dimensional losses (src/losses.jl:360-620): dim columns of Y per column of A; bin_loss must be Logistic / Hinge
binkind(b::LogisticLoss) = 7.0
binkind(b::WeightedHingeLoss) = b.case_weight_ratio == 1 ? 8.0 : NaN
binkind(b) = NaN
closs(l::MultinomialLoss) = CLoss(9, l.max, l.scale, 0, 0)
closs(l::OvALoss) = isnan(binkind(l.bin_loss)) ? nothing : CLoss(10, l.max, l.scale, l.bin_loss.scale, binkind(l.bin_loss))
closs(l::BvSLoss) = isnan(binkind(l.bin_loss)) ? nothing : CLoss(11, l.max - 1, l.scale, l.bin_loss.scale, binkind(l.bin_loss))
closs(l::OrdisticLoss) = CLoss(12, l.max, l.scale, 0, 0)
closs(l::MultinomialOrdinalLoss) = CLoss(13, l.max - 1, l.scale, 0, 0)
closs(l::Loss) = nothing                     # anything else: reference path
creg(r::ZeroReg) = CReg(0, 0, 1.0)                                            # src/regularizers.jl:91-97
creg(r::QuadReg) = CReg(1, 0, r.scale)                                        # :52-58
creg(r::OneReg) = CReg(2, 0, r.scale)                                         # :79-88
creg(r::NonNegConstraint) = CReg(3, 0, 1.0)                                   # :101-114
creg(r::UnitOneSparseConstraint) = CReg(4, 0, 1.0)                            # :295-318
# wrappers around one of the five base regularizers (src/regularizers.jl:163-189,356-411)
wrapped(r, flag) = (b = creg(r.r); (b === nothing || b.wrap != 0) ? nothing : CReg(b.kind, flag, b.scale))
creg(r::lastentry1) = wrapped(r, 1)
creg(r::lastentry_unpenalized) = wrapped(r, 2)
creg(r::OrdinalReg) = wrapped(r, 4)
creg(r::MNLOrdinalReg) = wrapped(r, 8)
creg(r::Regularizer) = nothing

isclass(l) = l isa LogisticLoss || l isa WeightedHingeLoss
value(l, a) = isclass(l) ? (a isa Bool ? Float64(a) : Float64(LowRankModels.myBool(Int(a)))) : Float64(a)   # src/losses.jl:104-106
collapse(v) = all(==(v[1]), v) ? v[1:1] : v          # one descriptor when every column / row carries the same one

fallback(glrm, p; kw...) = fit!(glrm, ProxGradParams(p.stepsize; max_iter=p.max_iter, inner_iter_X=p.inner_iter_X, inner_iter_Y=p.inner_iter_Y,
                                             abs_tol=p.abs_tol, rel_tol=p.rel_tol, min_stepsize=p.min_stepsize); kw...)

# descriptors of a model, or nothing if some loss / regularizer type is outside include/glrm_hip.h
function descriptors(glrm::GLRM)
    cl = map(closs, glrm.losses); crx = map(creg, glrm.rx); cry = map(creg, glrm.ry)
    (any(isnothing, cl) || any(isnothing, crx) || any(isnothing, cry)) && return nothing
    n = size(glrm.A, 2)
    general = embedding_dim(glrm.losses) != n || any(c -> c.wrap != 0, crx) || any(c -> c.wrap != 0, cry)
    (general && glrm.k > 64) && return nothing
    collapse(Vector{CLoss}(cl)), collapse(Vector{CReg}(crx)), collapse(Vector{CReg}(cry))
end

# the dense hand-over applies when every entry is observed (the constructor's default UnitRanges) under one QuadLoss
fully_observed(glrm) = (s = size(glrm.A); all(==(1:s[2]), glrm.observed_features) && all(==(1:s[1]), glrm.observed_examples))
dense_ok(glrm, desc, p) = p.dense && glrm.A isa Matrix{Float64} && length(desc[1]) == 1 && desc[1][1].kind == 0 &&
                          9 <= glrm.k <= 64 && fully_observed(glrm)
