"""Solver parameters (reference: src/algorithms/proxgrad.jl:4-31, src/fit.jl:4-5)."""


class AbstractParams:
    pass


class ProxGradParams(AbstractParams):
    """ProxGradParams(stepsize=1.0; max_iter=100, inner_iter_X=1, inner_iter_Y=1, inner_iter=1,
    abs_tol=1e-5, rel_tol=1e-4, min_stepsize=0.01*stepsize) -- same defaults and the same
    max-merge of ``inner_iter`` into both inner counts (src/algorithms/proxgrad.jl:13-31)."""

    def __init__(self, stepsize=1.0, *, max_iter=100, inner_iter_X=1, inner_iter_Y=1, inner_iter=1,
                 abs_tol=0.00001, rel_tol=0.0001, min_stepsize=None):
        self.stepsize = float(stepsize)
        self.max_iter = int(max_iter)
        self.inner_iter_X = max(int(inner_iter_X), int(inner_iter))
        self.inner_iter_Y = max(int(inner_iter_Y), int(inner_iter))
        self.abs_tol = float(abs_tol)
        self.rel_tol = float(rel_tol)
        self.min_stepsize = float(0.01 * self.stepsize if min_stepsize is None else min_stepsize)

    def __repr__(self):
        f = ("stepsize", "max_iter", "inner_iter_X", "inner_iter_Y", "abs_tol", "rel_tol", "min_stepsize")
        return f"{type(self).__name__}(" + ", ".join(f"{k}={getattr(self, k)}" for k in f) + ")"


class HipProxGradParams(ProxGradParams):
    """The drop-in params type: the ProxGradParams fields plus engine knobs.  In Julia this is
    ``struct HipProxGradParams <: AbstractParams`` (julia/HipGLRM.jl); `fit!(glrm, params=p)`
    dispatches on it exactly like on the built-in solvers (src/fit.jl:8-12).

    Tolerance (north star: 1e-5 relative on trajectory and factor values).  ``mode="fast"`` -- production, what bench.py times -- keeps the
    objective trajectory within 1e-5 of the reference-order oracle's on every committed fixture (C4 recipe at 1e8 observations: 7.9e-6 after
    100 iterations; C2 2e-11; C5 4e-13) and equals the oracle bit for bit in the summation order the engine reports.  Factor ENTRIES of recipes
    that amplify rounding (the NNMF of BASELINE config 4: 7e-3 after 100 iterations -- the oracle against itself in two orders) meet 1e-5 only
    with ``mode="reference_order"`` (every sum added as the reference adds it, ~7x slower): there the whole recorded objective vector and the
    factor samples equal the reference-order oracle's to the last bit (tests/test_gpu_jref.py)."""

    def __init__(self, stepsize=1.0, *, device_id=-1, profile=False, waves_row=0, waves_col=0, tiled=0, dense=True, ngpus=1,
                 device_ids=None, exchange="direct", x_chunks=0, quad_gram=False, mode="fast", **kw):
        super().__init__(stepsize, **kw)
        # SURVEY.md 8(b) `mode` / `line_search_sum_order`: "fast" = the engine's summation orders; "reference_order" = the validation
        # sweeps that add every sum like the reference does (glrm_options.sum_order = 1: scalar losses, list problems, k <= 64; slow)
        if mode not in ("fast", "reference_order"):
            raise ValueError("mode must be 'fast' or 'reference_order'")
        self.mode = mode
        # fully observed QuadLoss models only (glrm_options.quad_gram): line-search trials from the quadratic form
        # J(x) + g.s + scale s'(YY')s instead of another pass over A; same iterates up to rounding.  Off by default.
        self.quad_gram = bool(quad_gram)
        self.device_id, self.profile = int(device_id), bool(profile)
        self.waves_row, self.waves_col, self.tiled = int(waves_row), int(waves_col), int(tiled)
        self.dense = bool(dense)  # fully observed QuadLoss models: run the half-steps on the matrix cores
        # ngpus > 1: ONE process drives `ngpus` devices through glrm_hip_multi_* (rows / columns sharded by the library, X and Y
        # replicated, direct peer-copy or RCCL exchange after every half-step) -- SURVEY.md section 8(b)'s `ngpus` field.
        self.ngpus = int(ngpus)
        self.device_ids = None if device_ids is None else [int(d) for d in device_ids]
        if self.device_ids is not None and len(self.device_ids) != self.ngpus:
            raise ValueError("device_ids must list one device per shard (ngpus entries)")
        if exchange not in ("direct", "rccl"):
            raise ValueError("exchange must be 'direct' or 'rccl'")
        self.exchange, self.x_chunks = exchange, int(x_chunks)


def Params(*args, **kwargs):  # src/fit.jl:5
    return ProxGradParams(*args, **kwargs)


class SparseProxGradParams(AbstractParams):
    """SparseProxGradParams(stepsize=1.0; max_iter=100, inner_iter=1, abs_tol=1e-5, min_stepsize=0.01*stepsize)
    (src/algorithms/sparse_proxgrad.jl:4-19): one global step size, whole-iteration accept / revert.  The reference's
    `fit!(glrm)` picks it for SparseMatrixCSC input (src/fit.jl:13-15)."""

    def __init__(self, stepsize=1.0, *, max_iter=100, inner_iter=1, abs_tol=0.00001, min_stepsize=None, device_id=-1, tiled=0):
        self.stepsize = float(stepsize)
        self.max_iter, self.inner_iter = int(max_iter), int(inner_iter)
        self.abs_tol = float(abs_tol)
        self.min_stepsize = float(0.01 * self.stepsize if min_stepsize is None else min_stepsize)
        self.device_id, self.tiled = int(device_id), int(tiled)

    def __repr__(self):  # Julia prints the struct positionally (println(params), sparse_proxgrad.jl:26)
        return f"SparseProxGradParams({self.stepsize}, {self.max_iter}, {self.inner_iter}, {self.abs_tol}, {self.min_stepsize})"
