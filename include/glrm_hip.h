/*
 * glrm_hip.h -- C ABI of libglrm_hip.so, the MI355X (gfx950) engine behind
 * LowRankModels.jl's `fit!(glrm::GLRM, params::ProxGradParams)`.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * A Julia `ccall`, a Python `ctypes` or a C caller binds exactly these symbols.
 * The same entry points exist with a `glrm_cpu_` prefix in oracle/ (the CPU
 * restatement used ONLY as the parity checker and CPU baseline).
 *
 * Reference interfaces replaced (paths relative to the LowRankModels.jl tree):
 *   glrm_params    <- ProxGradParams                src/algorithms/proxgrad.jl:4-31
 *   glrm_problem   <- GLRM struct + Omega lists     src/glrm.jl:12-22, src/modify_glrm.jl:5-18
 *   glrm_loss      <- Loss subtypes                 src/losses.jl:138-352 (scalar), :360-620 (multi-dimensional)
 *   glrm_reg       <- Regularizer subtypes          src/regularizers.jl:52-114,295-318; wrappers :163-189,356-411
 *   glrm_domain    <- Domain subtypes               src/domains.jl
 *   glrm_hip_fit   <- fit!(::GLRM,::ProxGradParams) src/algorithms/proxgrad.jl:34-220
 *   glrm_hip_fit_sparse <- fit!(::GLRM,::SparseProxGradParams) src/algorithms/sparse_proxgrad.jl:22-134
 *   glrm_hip_objective <- objective(glrm,X,Y;...)   src/evaluate_fit.jl:57-81
 *   glrm_hip_init_svd  <- init_svd!(glrm)           src/initialize.jl:35-132
 *   glrm_hip_subset    <- fold models of cross_validate / cv_by_iter / regularization_path   src/cross_validate.jl:20-37,54-105
 *   glrm_hip_error_metric / glrm_hip_impute <- error_metric / impute   src/evaluate_fit.jl:107-168, src/impute_and_err.jl
 *   objective/seconds arrays <- ConvergenceHistory  src/convergence.jl:3-27
 *
 * Conventions
 *   - X is k x m, Y is k x d, both COLUMN-major with leading dimension k, i.e.
 *     X[:,e] and Y[:,c] are k contiguous doubles (same as the Julia arrays,
 *     src/algorithms/proxgrad.jl:90-91).  d = sum of the losses' embedding dimensions
 *     (= n when every loss is scalar); column f owns Y columns [ys_f, ys_f + dim_f)
 *     in column order (get_yidxs, src/losses.jl:76-93).
 *   - All indices are 0-based across this ABI (the Julia shim shifts by one).
 *   - Omega is handed over twice and the two views are NEVER derived from each
 *     other: CSR-by-row = observed_features, CSC-by-column = observed_examples,
 *     both in the reference's list order with duplicates kept
 *     (src/modify_glrm.jl:8-12).
 *   - Every function returns 0 on success or a negative glrm_status; the message
 *     is available from glrm_hip_last_error() (thread-local).  Nothing throws or
 *     aborts across the boundary.
 *   - A handle is not thread-safe (one call at a time per handle).
 */
#ifndef GLRM_HIP_H
#define GLRM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: glrm_options grew by quad_gram / reserved (round 2), glrm_signature + glrm_hip_signature / glrm_hip_finalize and
 *    GLRM_PROBLEM_DEFER_SETUP were added (round 3).  A host built against ABI 1 fails the version check instead of handing over
 *    a 32-byte glrm_options. */
/* 3: glrm_options grew by sum_order and one more reserved word (48 bytes), glrm_kernel_stats by ms_wait_y, glrm_multi_options.reserved became
 *    `arrival`; glrm_arrival + glrm_hip_step_y_arrival were added (round 5). */
#define GLRM_HIP_ABI_VERSION 3

typedef enum glrm_status {
  GLRM_OK = 0,
  GLRM_ERR_INVALID = -1,     /* bad argument / inconsistent sizes (reference: error(...) src/glrm.jl:39-43) */
  GLRM_ERR_UNSUPPORTED = -2, /* loss / regularizer kind or rank not implemented by the engine */
  GLRM_ERR_HIP = -3,         /* HIP runtime error */
  GLRM_ERR_COMM = -4,        /* inter-GPU exchange failed (peer copy / RCCL) in the glrm_hip_multi_* entry points */
  GLRM_ERR_OOM = -5,         /* device or host allocation failed */
  GLRM_ERR_NONFINITE = -6    /* NaN observation (src/glrm.jl:63-71) or bad Bool label (src/losses.jl:104) */
} glrm_status;

/* Scalar losses, src/losses.jl. */
typedef enum glrm_loss_kind {
  GLRM_LOSS_QUAD = 0,           /* QuadLoss          :138-148  scale            */
  GLRM_LOSS_L1 = 1,             /* L1Loss            :152-162  scale            */
  GLRM_LOSS_HUBER = 2,          /* HuberLoss         :166-179  scale, p0=crossover */
  GLRM_LOSS_QUANTILE = 3,       /* QuantileLoss      :186-203  scale, p0=quantile  */
  GLRM_LOSS_PERIODIC = 4,       /* PeriodicLoss      :209-224  scale, p0=T         */
  GLRM_LOSS_POISSON = 5,        /* PoissonLoss       :231-243  scale            */
  GLRM_LOSS_ORDINAL_HINGE = 6,  /* OrdinalHingeLoss  :247-294  scale, p0=min, p1=max */
  GLRM_LOSS_LOGISTIC = 7,       /* LogisticLoss      :298-311  scale; a in {1.0 (true), 0.0 (false)} */
  GLRM_LOSS_WEIGHTED_HINGE = 8, /* WeightedHingeLoss :317-352  scale, p0=case_weight_ratio; a in {1.0,0.0} */
  /* multi-dimensional losses: the column owns `dim` consecutive columns of Y (embedding_dim, :72-93); a is the level 1..max */
  GLRM_LOSS_MULTINOMIAL = 9,          /* MultinomialLoss(max)        :360-409  dim = max                              */
  GLRM_LOSS_OVA = 10,                 /* OvALoss(max; bin_loss)      :413-446  dim = max,   p0 = bin_loss.scale, p1 = bin kind (7|8) */
  GLRM_LOSS_BVS = 11,                 /* BvSLoss(max; bin_loss)      :450-483  dim = max-1, p0 = bin_loss.scale, p1 = bin kind (7|8) */
  GLRM_LOSS_ORDISTIC = 12,            /* OrdisticLoss(max)           :490-530  dim = max                              */
  GLRM_LOSS_MULTINOMIAL_ORDINAL = 13, /* MultinomialOrdinalLoss(max) :562-620  dim = max-1                            */
  GLRM_LOSS_KIND_COUNT = 14
} glrm_loss_kind;

/* Regularizers named by the north star, src/regularizers.jl. */
typedef enum glrm_reg_kind {
  GLRM_REG_ZERO = 0,            /* ZeroReg                 :91-97   */
  GLRM_REG_QUAD = 1,            /* QuadReg(scale)          :52-58   */
  GLRM_REG_ONE = 2,             /* OneReg(scale)           :79-88   */
  GLRM_REG_NONNEG = 3,          /* NonNegConstraint        :101-114 */
  GLRM_REG_UNIT_ONE_SPARSE = 4, /* UnitOneSparseConstraint :295-318 */
  GLRM_REG_KIND_COUNT = 5
} glrm_reg_kind;

#define GLRM_MAX_EMBEDDING_DIM 32

typedef struct glrm_loss {
  int32_t kind;     /* glrm_loss_kind */
  int32_t dim;      /* embedding dimension: 0 or 1 for the scalar losses, 2..GLRM_MAX_EMBEDDING_DIM for kinds >= 9 */
  double scale;
  double p0;
  double p1;
} glrm_loss; /* 32 bytes */

/* wrappers around the base regularizer (bit set in glrm_reg.wrap) */
#define GLRM_WRAP_LASTENTRY1 1            /* lastentry1(r)            :163-175: last entry pinned to 1 (offset, on X)      */
#define GLRM_WRAP_LASTENTRY_UNPENALIZED 2 /* lastentry_unpenalized(r) :177-189: last row exempt from r (offset, on Y)      */
#define GLRM_WRAP_ORDINAL 4               /* OrdinalReg(r)            :356-383: block regularizer of ordinal multi-dim losses */
#define GLRM_WRAP_MNL_ORDINAL 8           /* MNLOrdinalReg(r)         :388-409                                               */

typedef struct glrm_reg {
  int32_t kind;     /* glrm_reg_kind of the base regularizer */
  int32_t wrap;     /* 0 or one GLRM_WRAP_* flag */
  double scale;
} glrm_reg; /* 16 bytes */

/* Domains (src/domains.jl): how a column's values are imputed from the fitted model, used only by the post-fit evaluation
 * entry points glrm_hip_impute / glrm_hip_error_metric (src/impute_and_err.jl). */
enum glrm_domain_kind {
  GLRM_DOMAIN_REAL = 0,        /* RealDomain() */
  GLRM_DOMAIN_BOOL = 1,        /* BoolDomain(): values true (1.0) / false (0.0) */
  GLRM_DOMAIN_ORDINAL = 2,     /* OrdinalDomain(lo, hi) */
  GLRM_DOMAIN_PERIODIC = 3,    /* PeriodicDomain(T): lo = T */
  GLRM_DOMAIN_COUNT = 4,       /* CountDomain(max_count): hi = max_count */
  GLRM_DOMAIN_CATEGORICAL = 5, /* CategoricalDomain(1, hi) */
  GLRM_DOMAIN_KIND_COUNT = 6
};
typedef struct glrm_domain {
  int32_t kind;
  int32_t reserved; /* must be 0 */
  double lo, hi;
} glrm_domain; /* 24 bytes */

#define GLRM_PROBLEM_DEVICE_ARRAYS 1 /* flags bit 0: rowptr..colvals are DEVICE pointers (copied, not adopted) */
#define GLRM_PROBLEM_BORROW_DEVICE_ARRAYS 4 /* flags bit 2, with bit 0: the engine reads the caller's device arrays IN PLACE instead of
                                        copying them; the caller keeps them alive and unchanged until glrm_hip_destroy.  The engine never
                                        writes to them (a view it has to reorder -- tile sort, grouping by loss kind -- gets a private
                                        copy).  For hosts whose Omega already lives in HBM and is too large to hold twice (BASELINE
                                        configs[4]: 120 GB of lists).  Not available for dense_A, whose packed copies replace the original. */
#define GLRM_PROBLEM_ROWS_FROM_COLS 8 /* flags bit 3: Omega is a sparse matrix's pattern -- both views list the SAME entries, each ascending,
                                        which is what the reference builds from `findall(!iszero, A)` for SparseMatrixCSC input
                                        (src/glrm.jl:46-48, src/modify_glrm.jl:8-12).  rowptr / colidx / rowvals must be NULL: the engine derives
                                        the row view from the column view on the device (one stable sort of the column-major stream by
                                        row id).  A host holding a CSC matrix hands over colptr / rowval / nzval after an index shift and
                                        nothing else.  Whole-problem list handles (row and column range complete), any observation count
                                        (beyond 1.5e9 the stream is sorted in row ranges), not with GLRM_PROBLEM_BORROW_DEVICE_ARRAYS. */
#define GLRM_PROBLEM_DEFER_SETUP 2   /* flags bit 1: this is one shard of a sharded fit -- glrm_hip_create only uploads it; the host
                                        combines the shards' glrm_signature and calls glrm_hip_finalize on every shard before the
                                        first step (see glrm_signature below) */

/*
 * One shard of a GLRM.  A single-GPU problem is the shard
 * [row_begin,row_end) = [0,m), [col_begin,col_end) = [0,n).
 * The shard owns the X half-step of its rows and the Y half-step of its
 * columns; X (k x m) and Y (k x n) are replicated on every shard
 * (row/column independence: src/algorithms/proxgrad_multithread.jl:118,163).
 */
typedef struct glrm_problem {
  int64_t m, n;               /* global matrix size */
  int32_t k;                  /* rank */
  int32_t flags;              /* GLRM_PROBLEM_* */
  int64_t row_begin, row_end; /* rows whose x_e this shard updates */
  int64_t col_begin, col_end; /* columns whose y_f this shard updates */
  /* CSR over the local rows: observed_features[e] for e in [row_begin,row_end) */
  const int64_t* rowptr;      /* (row_end-row_begin)+1 entries, rowptr[0]==0 */
  const int32_t* colidx;      /* global column ids, list order, duplicates kept */
  const double* rowvals;      /* A[e,f] for each listed (e,f) */
  /* CSC over the local columns: observed_examples[f] for f in [col_begin,col_end) */
  const int64_t* colptr;      /* (col_end-col_begin)+1 entries, colptr[0]==0 */
  const int32_t* rowidx;      /* global row ids */
  const double* colvals;
  const glrm_loss* losses;    /* n_losses entries (host memory) */
  int64_t n_losses;           /* 1 (every column alike) or n */
  const glrm_reg* rx;         /* n_rx entries (host memory) */
  int64_t n_rx;               /* 1 (every row alike) or row_end-row_begin */
  const glrm_reg* ry;
  int64_t n_ry;               /* 1 or col_end-col_begin */
  /* Fully observed QuadLoss case (every (e,f) observed, one QuadLoss descriptor): hand over the whole dense
   * m x n matrix instead of the two index lists (rowptr..colvals must then be NULL).  The engine keeps a
   * row-major copy of the shard's rows and a column-major copy of its columns and runs the half-steps on the
   * fp64 matrix cores.  GLRM_PROBLEM_DEVICE_ARRAYS applies to dense_A as well. */
  const double* dense_A;      /* NULL = sparse (list) problem */
  int64_t dense_ld;           /* leading dimension in elements */
  int32_t dense_colmajor;     /* 1: A(i,j) = dense_A[i + j*dense_ld] (Julia), 0: dense_A[i*dense_ld + j] (C / numpy) */
  int32_t dense_reserved;     /* must be 0 */
} glrm_problem;

/* ProxGradParams, src/algorithms/proxgrad.jl:4-12 (inner_iter already merged, :22-23). */
typedef struct glrm_params {
  double stepsize;
  int64_t max_iter;
  int64_t inner_iter_X;
  int64_t inner_iter_Y;
  double abs_tol;
  double rel_tol;
  double min_stepsize;
} glrm_params;

/* SparseProxGradParams, src/algorithms/sparse_proxgrad.jl:4-19: the solver `fit!(glrm)` picks for SparseMatrixCSC input. */
typedef struct glrm_sparse_params {
  double stepsize;
  int64_t max_iter;
  int64_t inner_iter;
  double abs_tol;
  double min_stepsize;
} glrm_sparse_params;

typedef struct glrm_options {
  int32_t device_id; /* HIP device ordinal; -1 = current device */
  int32_t profile;   /* 1 = bracket every sweep launch with HIP events (glrm_hip_kernel_stats) */
  int32_t waves_row; /* waves cooperating on one row   (0 = choose from mean |Omega_e|; 1, 4 or 8) */
  int32_t waves_col; /* waves cooperating on one column (0 = choose from mean |Omega^f|) */
  void* stream;      /* hipStream_t to launch on */
  int32_t caller_stream; /* 0: stream==NULL means "the handle creates a private non-blocking stream";
                            1: launch on `stream` exactly as given, even NULL (the legacy default stream) --
                            what a host that orders its own collectives on that stream must pass */
  int32_t tiled;     /* sweep kernels: 0 = choose (LDS-tiled when the index lists are tile-ordered -- e.g. sorted -- and the
                        problem is large enough), 1 = gather sweeps only, 2 = LDS-tiled sweeps wherever the lists allow */
  int32_t quad_gram; /* dense_A hand-over only.  1 = the line-search trials of a half-step are evaluated from the quadratic form
                        J(x') = J(x) + g.(x'-x) + scale (x'-x)' (Y Y') (x'-x) instead of one more pass over A per trial (the
                        objective of a fully observed QuadLoss row IS that quadratic; Y Y' is shared by all rows).  Same
                        algorithm and iterates as row_objective (src/evaluate_fit.jl:24-38) up to rounding (<< 1e-5; a trial whose
                        decrease is below the rounding of the two sums the reference compares may be decided differently).
                        0 = evaluate every trial by a pass over A, like the reference.  Default 0. */
  int32_t sum_order; /* SURVEY.md section 8(b) `line_search_sum_order`.  0 = the engine's own orders (the fast kernel families; what they
                        are is reported by glrm_hip_sum_order).  1 = GLRM_ORDER_REFERENCE, a VALIDATION mode: every sum of the half-steps
                        is added as the reference adds it -- the k-term dot product in component order, the gradient axpys and the row
                        loss sum in list order into ONE accumulator (src/algorithms/proxgrad.jl:122-132,165-175, src/evaluate_fit.jl:28-36),
                        the column loss sum by Julia's pairwise reduce(+) with blocks of 1024 for DiffLoss / ClassificationLoss columns
                        and sequentially for OrdinalHinge / Poisson (src/losses.jl:623-638), sum(obj_by_col) pairwise (proxgrad.jl:205)
                        -- by one lane per segment (csrc/glrm_reforder.hip).  Slow (no lane-level parallelism inside a segment, one
                        gather per observation and pass); exists so that a checker can hold the engine against the REFERENCE-order
                        oracle to rounding on trajectories that amplify summation order.  Scalar losses, list
                        problems. */
  int32_t reserved0; /* must be 0 */
  int32_t reserved;  /* must be 0 */
} glrm_options; /* 48 bytes */

typedef struct glrm_handle glrm_handle;

/*
 * What the engine's kernel choice looks at besides (m, n, k, losses, options).  The sweep families (gather, LDS-tiled,
 * phase-aligned passes, cached rows) add the same numbers in different orders, so a sharded fit is bit-identical to the
 * single-shard fit only if every shard makes the SAME choice: the choice is therefore a function of the WHOLE problem's signature,
 * never of the shard's.  (Per-segment choices -- waves per segment, which rows the cached sweep holds in registers -- are functions
 * of the segment's own length.)  A host that shards a problem creates every shard with GLRM_PROBLEM_DEFER_SETUP, reads each
 * shard's signature, combines them (sum the counts, max the rest) and hands the result to glrm_hip_finalize on every shard.  The
 * in-library multi-device fit (glrm_hip_multi_*) and lowrankmodels.jl_amd/fit.py::ShardedFit do exactly that.
 * The reference has no counterpart: its threads share one address space (src/algorithms/proxgrad_multithread.jl:118,163).
 */
typedef struct glrm_signature {
  int64_t nnz_rows, nnz_cols;       /* observations in the row / column view                  (whole problem: sum over shards) */
  int64_t max_row_len, max_col_len; /* longest row / column list                              (max over shards) */
  int32_t rows_unordered;           /* 1: some row list is not ordered by LDS tile            (max over shards) */
  int32_t cols_unordered;           /* 1: some column list is not ordered by LDS tile         (max over shards) */
} glrm_signature; /* 40 bytes */

/* ---- whole-fit API (what the Julia `fit!` shim calls) -------------------------------- */

int glrm_hip_version(void);
const char* glrm_hip_last_error(void);

/* Copies the shard's Omega views, values and descriptors to the device. */
int glrm_hip_create(glrm_handle** out, const glrm_problem* p, const glrm_options* o);
void glrm_hip_destroy(glrm_handle* h); /* NULL is a no-op */
/* This shard's contribution to the signature of the whole problem (any handle, any time after create). */
int glrm_hip_signature(glrm_handle* h, glrm_signature* local);
/* Second half of glrm_hip_create for a handle created with GLRM_PROBLEM_DEFER_SETUP: chooses the kernel families from the
 * signature of the WHOLE problem (NULL = this shard is the whole problem) and allocates their buffers.  Every step-level call on
 * a deferred handle fails with GLRM_ERR_INVALID until this has run; calling it twice, or on a handle created without the flag,
 * is GLRM_ERR_INVALID as well. */
int glrm_hip_finalize(glrm_handle* h, const glrm_signature* whole);

/*
 * fit!(glrm, ProxGradParams) for a single-shard handle (src/algorithms/proxgrad.jl:34-220).
 * X (k x m) and Y (k x n) are read on entry (warm start) and overwritten on exit.
 * objective[0] is the full initial objective (loss + rx + ry, :76); objective[i>=1] is
 * sum(obj_by_col) after outer iteration i (:205, excludes rx).  seconds[] is cumulative
 * wall-clock like ch.times (src/convergence.jl:22-26).  cap must be >= max_iter+1.
 */
int glrm_hip_fit(glrm_handle* h, const glrm_params* prm, double* X, double* Y,
                 double* objective, double* seconds, int64_t cap, int64_t* n_recorded);

/*
 * fit!(glrm, SparseProxGradParams) (src/algorithms/sparse_proxgrad.jl:22-134): one global step size, one gradient + prox
 * step per factor and iteration (no per-row line search), the whole iteration is accepted or reverted on the full
 * objective(glrm, X, Y; sparse=true).  objective[] holds the initial objective, one entry per ACCEPTED iteration and the
 * last value once more (:126-127); cap must be >= max_iter+2.  X, Y return the best model found.
 */
int glrm_hip_fit_sparse(glrm_handle* h, const glrm_sparse_params* prm, double* X, double* Y, double* objective,
                        double* seconds, int64_t cap, int64_t* n_recorded);

/* objective(glrm, X, Y; include_regularization) over observed_examples (src/evaluate_fit.jl:57-81). */
int glrm_hip_objective(glrm_handle* h, const double* X, const double* Y, int include_reg, double* out);

/* ---- step-level API (multi-GPU hosts: one process per GPU, collectives in the host) ---- */
/*
 * Device-resident factors use leading dimension ld = glrm_hip_factor_ld(h) >= k
 * (zero padded).  A distributed host allocates dX (ld*m), dY (ld*n), dObjCol (n) and
 * dObjRow (m) doubles on the handle's device, binds them, and between the calls below
 * all-gathers the slices the shard wrote:
 *   step_x     -> dX[ld*row_begin : ld*row_end]
 *   step_y     -> dY[ld*col_begin : ld*col_end], dObjCol[col_begin:col_end]
 *   col_losses -> dObjCol[col_begin:col_end]       (loss only, no ry)
 *   row_penalties -> dObjRow[row_begin:row_end]    (rx_e(x_e))
 *   col_penalties -> dObjCol[col_begin:col_end]    (ry_f(y_f))
 * All launches are asynchronous on the handle's stream.
 */
int glrm_hip_factor_ld(glrm_handle* h);
int glrm_hip_bind_buffers(glrm_handle* h, void* dX, void* dY, void* dObjCol, void* dObjRow);
int glrm_hip_set_factors(glrm_handle* h, const double* X, const double* Y); /* host k x m, k x n -> device */
int glrm_hip_get_factors(glrm_handle* h, double* X, double* Y);             /* device -> host (synchronises) */
int glrm_hip_reset_stepsizes(glrm_handle* h, double stepsize);              /* alpharow, alphacol (:69-70,:112-115) */
int glrm_hip_step_x(glrm_handle* h, double min_stepsize);                   /* one inner X sweep (:118-156) */
int glrm_hip_step_y(glrm_handle* h, double min_stepsize);                   /* one inner Y sweep (:162-201) */
/* The Y half-step while the updated X is still ARRIVING from the other shards.  `blocks` tile the rows [0, m) of X: block i is complete
 * on this device once `event` (a hipEvent_t the host recorded behind whatever fills the range -- a peer copy, a collective; NULL = the
 * range is already there, e.g. the shard's own rows) has fired.  The order of the array is the order in which the host expects the
 * blocks.  The phase-aligned column passes (csrc/glrm_blocked.hip) walk X one super-tile per launch and keep one partial sum per (column,
 * super-tile) that col_reduce adds in super-tile order whatever order the launches ran in: they launch each super-tile behind the events
 * of the blocks it touches, own rows first, so the exchange overlaps the half-step that consumes it.  The LDS-tiled column passes and their
 * lane-per-segment form launch their gradient pass in runs of super-tiles, each behind the blocks it reads, in the announced order.  Every
 * other family waits for all events and then runs glrm_hip_step_y.  Results are those of glrm_hip_step_y bit for bit.  With glrm_options.profile the time the
 * launch stream spent in those waits is accounted in glrm_kernel_stats.ms_wait_y (in true arrival order: plus the time the calling
 * thread polled while no super-tile was ready).
 * TRUE arrival order (default; GLRM_HIP_ARRIVAL_DYNAMIC=0 restores the announced order): a super-tile is enqueued once the events of all
 * its blocks have fired (hipEventQuery), the ready ones in the announced order, and the calling thread polls for the rest -- behind a
 * lagging peer the launch stream no longer stands in front of its block while other super-tiles are ready.  The call therefore returns
 * only when every block has arrived (the line search that follows reads its counters back anyway).  An event that cannot be queried, or
 * 5 s without progress, hands the remaining super-tiles to in-stream waits in the announced order. */
typedef struct glrm_arrival {
  int64_t begin, end; /* rows [begin, end) of X */
  void* event;        /* hipEvent_t or NULL */
} glrm_arrival;
int glrm_hip_step_y_arrival(glrm_handle* h, double min_stepsize, const glrm_arrival* blocks, int32_t n_blocks);
/* One prox-gradient step of the shard's rows / columns with a global step size and no line search
 * (src/algorithms/sparse_proxgrad.jl:59-77 / :81-99). */
int glrm_hip_gradstep_x(glrm_handle* h, double alpha);
int glrm_hip_gradstep_y(glrm_handle* h, double alpha);
/* The X sweep restricted to the shard's local rows [seg_begin, seg_end): lets a multi-GPU host pipeline the
 * all-gather of finished row chunks behind the sweep of the next chunk (rows are independent, :118). */
int glrm_hip_step_x_range(glrm_handle* h, int64_t seg_begin, int64_t seg_end, double min_stepsize);
int glrm_hip_col_losses(glrm_handle* h);
int glrm_hip_row_penalties(glrm_handle* h);
int glrm_hip_col_penalties(glrm_handle* h);
/* Replace the regularizer descriptors of an existing handle (same counts as at create: 1 or one per local row /
 * column).  Omega, A and the losses stay on the device: this is what `regularization_path` / `scale_regularizer!`
 * (src/cross_validate.jl:228-231, src/glrm.jl:85-89) need between warm-started fits. */
int glrm_hip_set_regularizers(glrm_handle* h, const glrm_reg* rx, int64_t n_rx, const glrm_reg* ry, int64_t n_ry);
/* A new handle over a SUBSET of the parent's observations, built on the device from the parent's resident Omega views and
 * values (nothing but the tags crosses PCIe: 1 byte per observation and view instead of 12).  Entry t of the parent's row
 * view is kept iff (row_tags[t] == match) != invert, entry t of its column view iff (col_tags[t] == match) != invert; order
 * and duplicates inside a row / column are preserved.  Losses, regularizers, rank, shard ranges and options are the parent's.
 * This is the train / test split of cross_validate, cv_by_iter and regularization_path (getfolds / get_train_and_test,
 * src/cross_validate.jl:54-105): fold f's training model is subset(tags = fold ids, match = f, invert = 1), its test model
 * subset(..., invert = 0).  The parent must be a list (not dense), finalized handle; it is not modified and may be destroyed first.
 * Children of SHARDS (a parent whose row / column ranges are not the whole problem): the child is one shard of the subset problem and
 * must, like its parent, choose its kernels from the signature of the WHOLE subset problem -- it is therefore returned in the
 * GLRM_PROBLEM_DEFER_SETUP state.  The host calls glrm_hip_signature on every shard's child, combines them (sum the counts, max the
 * rest) and calls glrm_hip_finalize on every child with the result; until then every step-level call on such a child fails with
 * GLRM_ERR_INVALID.  A glrm_hip_finalize that fails half way (e.g. GLRM_ERR_OOM in a family's buffers) leaves the child unusable: a
 * second finalize is refused and the handle can only be destroyed.  The child of a single-shard parent is ready on return. */
int glrm_hip_subset(glrm_handle* parent, const uint8_t* row_tags, const uint8_t* col_tags, int32_t match, int32_t invert,
                    glrm_handle** out);
/* init_svd!(glrm) (src/initialize.jl:35-132) on the resident lists of a single-shard list handle: the observed entries are
 * expanded to reals (categorical / ordinal multi-dimensional columns become +-1 indicators), centred by the per-column mean of
 * the observed entries, scaled by m*n/|Omega|, and X = sqrt(S) U', Y = sqrt(S) V' diag(std) of the top-k singular triplets are
 * written to the caller's k x m and k x d buffers (unobserved entries count as 0, like in the reference).  The reference calls
 * Arpack's svds; here a randomized subspace iteration (block size k + 8) runs until the k leading singular values move by
 * less than tol * s_1 (tol <= 0: 1e-12; max_iter <= 0: 60).  X and Y are determined up to the sign of each component; X'Y is
 * unique.  The two Omega views must list the same entries without duplicates (the reference works on the expanded MATRIX).
 * `offset` / `scale` of the reference are dead code there (typeof(glrm.rx) == lastentry1 is never true for a Vector) and not
 * reproduced.  singular_values (k) and iters_done may be NULL. */
int glrm_hip_init_svd(glrm_handle* h, double* X, double* Y, int32_t max_iter, double tol, uint64_t seed, double* singular_values,
                      int32_t* iters_done);
/* error_metric(glrm, X, Y, domains; standardize) (src/evaluate_fit.jl:107-153): over observed_examples, the squared error
 * (Real / Ordinal / Count / Periodic domains) or 0-1 misclassification (Bool / Categorical) between A[i,j] and
 * impute(domains[j], losses[j], (X'Y)[i, yidxs[j]]) (src/impute_and_err.jl:37-130); with standardize each column's sum is
 * divided by the mean of A[i,j]^2 over its observed entries (when that is not 0).  domains: n descriptors.  Single-shard
 * list handle.  Domain / loss pairs for which the reference throws (RealDomain + LogisticLoss) return GLRM_ERR_UNSUPPORTED. */
int glrm_hip_error_metric(glrm_handle* h, const double* X, const double* Y, const glrm_domain* domains, int32_t standardize, double* out);
/* impute(domains, losses, X'Y) (src/impute_and_err.jl:149-165, impute(glrm) src/evaluate_fit.jl:156): the full m x n matrix of
 * imputed values, column-major (Ahat[i + j*m]), Bool columns as 1.0 / 0.0. */
int glrm_hip_impute(glrm_handle* h, const double* X, const double* Y, const glrm_domain* domains, double* Ahat);
/* Fixed-order sum of n device doubles (independent of the number of shards); synchronises. */
int glrm_hip_sum(glrm_handle* h, const void* dvec, int64_t n, double* out);
int glrm_hip_synchronize(glrm_handle* h);

/* ---- multi-GPU whole-fit API: ONE host process drives N devices (what a Julia `fit!` can ccall) ------------------------
 *
 * The reference's own parallel contract is "rows independent, then columns independent" inside one address space
 * (src/algorithms/proxgrad_multithread.jl:118,163).  glrm_hip_multi_create takes the SAME single-shard problem description as
 * glrm_hip_create (host arrays: the two Omega views or dense_A), cuts rows and columns into n_shards contiguous blocks balanced
 * by observation count, uploads block s to device_ids[s] and keeps a replica of X and Y on every device.  glrm_hip_multi_fit is
 * fit!(glrm, ProxGradParams) (src/algorithms/proxgrad.jl:34-220) on that layout: every shard runs the X half-step of its rows on
 * its own device from its own host thread and stream, the updated row blocks are exchanged (all-gather), then the same for the
 * columns; the recorded objective is a fixed-order sum of the gathered per-column values and the kernel families are chosen from the whole
 * problem (glrm_signature), so objective[1:], X and Y are bit-identical to glrm_hip_fit on one device for every n_shards;
 * objective[0] (initial loss + penalties, summed per shard block) agrees to rounding.
 *
 * Exchange (glrm_multi_options.exchange): 0 = direct -- every device pushes its block to each peer with hipMemcpyPeerAsync on a
 * copy stream per (source, destination) pair, i.e. all 7 xGMI links of a GPU carry one 1/N slice at once; 1 = RCCL ncclAllGather
 * in place (librccl is loaded at run time; needs distinct devices and equal blocks, otherwise the direct path is used).
 * device_ids may repeat (several shards on one device: the layout of the multi-GPU fit on a box with fewer GPUs).
 */
typedef struct glrm_multi glrm_multi;

typedef struct glrm_multi_options {
  int32_t n_shards;          /* >= 1 */
  int32_t exchange;          /* 0 direct peer copies (default), 1 RCCL all-gather */
  const int32_t* device_ids; /* n_shards HIP device ordinals; NULL = 0, 1, ..., n_shards-1 */
  int32_t x_chunks;          /* >= 2: the X half-step runs in row chunks whose exchange overlaps the sweep of the next chunk; 0/1 off */
  int32_t arrival;           /* direct exchange only.  0 (default) / 1: the Y half-step consumes the peers' row chunks of X in arrival order
                                (glrm_hip_step_y_arrival: own rows first, then chunk j of every peer as its copy event fires) instead of
                                waiting for the whole exchange between the half-steps; 2: off (wait, then glrm_hip_step_y) */
} glrm_multi_options;

int glrm_hip_multi_create(glrm_multi** out, const glrm_problem* p, const glrm_options* o, const glrm_multi_options* mo);
/* Same arguments and results as glrm_hip_fit. */
int glrm_hip_multi_fit(glrm_multi* mh, const glrm_params* prm, double* X, double* Y, double* objective, double* seconds,
                       int64_t cap, int64_t* n_recorded);
/* Replace the regularizer descriptors (n_rx = 1 or m, n_ry = 1 or n, like at create): regularization_path / scale_regularizer!
 * between warm-started fits, Omega and A stay on the devices. */
int glrm_hip_multi_set_regularizers(glrm_multi* mh, const glrm_reg* rx, int64_t n_rx, const glrm_reg* ry, int64_t n_ry);
/* row_bounds / col_bounds: n_shards+1 entries each (may be NULL); exchange_used: 0 direct, 1 RCCL;
 * exchange_ms: EXPOSED exchange time of the last fit (needs glrm_options.profile) -- the summed wall time the compute streams spent
 * between the end of a half-step's sweeps and the arrival of the last block they waited for there, plus (arrival order) the time the
 * Y half-steps' launch streams stood in front of a block that had not arrived yet; max over shards.
 * Link emulation (a box with fewer GPUs than shards): GLRM_EXCHANGE_EMULATE_GBPS=<GB/s per direction and link> makes every direct
 * push take at least bytes / rate from the moment its source range was complete (GLRM_EXCHANGE_EMULATE_DILATE=<d>, default the
 * number of shards that share the source device, divides the rate: d shards time-share one device, so a transfer must take d times as
 * long to keep its proportion to the compute).  See csrc/glrm_multigpu.hip. */
int glrm_hip_multi_info(glrm_multi* mh, int64_t* row_bounds, int64_t* col_bounds, int32_t* exchange_used, double* exchange_ms);
void glrm_hip_multi_destroy(glrm_multi* mh); /* NULL is a no-op */

/* ---- introspection ------------------------------------------------------------------- */

typedef struct glrm_kernel_stats {
  int64_t launches_x, launches_y;
  double ms_x, ms_y;           /* summed HIP-event durations of the row / column sweeps (profile=1) */
  int64_t trials_x, trials_y;  /* line-search trials taken (sum over segments and sweeps) */
  int64_t accepts_x, accepts_y;
  int64_t nnz_rows, nnz_cols;  /* |Omega| of the local CSR / CSC */
  int32_t waves_row, waves_col, ld;
  int32_t tiled;               /* bit0: LDS-tiled row sweep in use, bit1: LDS-tiled column sweep in use,
                                  bit2: dense MFMA path in use, bit3: general sweeps (multi-dimensional losses),
                                  bit4 / bit5: phase-aligned gather passes (L2-blocked) for the row / column sweep,
                                  bit6: cached gather row sweep (the short rows' opposing vectors fetched once per half-step
                                  and kept in registers / LDS for every pass), bit7: reference-order validation sweeps
                                  (glrm_options.sum_order = 1), bit8 / bit9: the LDS-tiled row / column passes run in their
                                  lane-per-segment form (csrc/glrm_lane.hpp) */
  double ms_wait_y;            /* glrm_hip_step_y_arrival with profile=1: time the launch stream waited for blocks of X to arrive; in TRUE
                                  arrival order also the time the calling thread polled with NO super-tile ready (an upper bound on the
                                  device's idle time: earlier super-tiles may still have been running) */
} glrm_kernel_stats;

int glrm_hip_kernel_stats(glrm_handle* h, glrm_kernel_stats* out, int reset);

/*
 * The ORDER in which a handle adds the terms of a segment's sums -- the k-term dot product <x_e, y_f>, the loss sum of
 * row_objective / col_objective (src/evaluate_fit.jl:24-55) and the gradient axpys (src/algorithms/proxgrad.jl:122-132,165-175).
 * The reference adds them in list order; the engine's sweep families add the same fp64 terms in orders fixed by their lane
 * layouts.  The accept test of the line search is a strict `<` between two such sums (proxgrad.jl:143,187), so a trajectory can
 * fork on the last bit: this query is what lets a checker attribute a deviation to summation order (SURVEY.md section 7.3 item 1;
 * section 8(b)'s `line_search_sum_order`).  The CPU oracle adopts a reported order (glrm_cpu_set_sum_order) and then reproduces
 * the engine's factors BIT FOR BIT (tests/test_gpu_sum_order.py), while oracle(reference order) vs oracle(engine order) shows
 * what the order alone does to a trajectory.  Additive in ABI 2 (a host that never calls it is unaffected).
 *
 * Common to every family: lane j of the `lanes` lanes that share an observation holds the components {2 lanes i + 2 j, + 1},
 * i = 0 .. comps/2 - 1, of the zero-padded kp = lanes x comps vector; its partial dot product is one fma chain over them in that
 * order (with `rotate`: over the chunks i ^ ((global segment id & 7) >> 1)); the lanes' partials are added by an xor butterfly
 * (pairs at distance 1, then 2, 4, 8).  The regularizer's sum of squares / absolute values is formed the same way.
 */
#define GLRM_ORDER_REFERENCE 0 /* list order, one accumulator per sum: the oracle's default, and the engine's with glrm_options.sum_order = 1 */
#define GLRM_ORDER_STRIDED 1   /* gather sweeps / cached row sweep: a segment is shared by W waves of 64 / lanes lane groups; group q
                                  of the T = W x 64 / lanes groups takes the observations q, q + T, q + 2T, ... in ascending order into
                                  its own loss and gradient accumulators; the groups of a wave are added by an xor butterfly (distance
                                  1, 2, 4, ... in group index), the waves in ascending order starting from 0.  batch = 4: the group's
                                  loss accumulator is four partials (observation u of each trip of 4, u = 0..3) added as a butterfly */
#define GLRM_ORDER_WINDOWED 2  /* LDS-tiled sweeps / phase-aligned passes: ONE lane group per segment walks the list in list order;
                                  gradient terms are added in list order; the loss terms go to `batch` partial sums by the entry's
                                  position inside its window modulo batch (a window = the entries whose opposing index lies in
                                  [w window, (w + 1) window)), added as a butterfly at the end of a super-tile; with
                                  windows_per_sup > 0 both sums start from 0 in every super-tile of that many windows and the
                                  super-tiles' partial sums are added in ascending order starting from 0 */
#define GLRM_ORDER_OTHER 3     /* dense MFMA path, general sweeps: not restated by the oracle */

typedef struct glrm_sum_order {
  int32_t family;          /* GLRM_ORDER_* */
  int32_t lanes, comps;    /* G, R: kp = lanes x comps */
  int32_t waves;           /* STRIDED: waves per segment; 0 = from the segment's own length: 1 below waves4_from, 4 below waves8_from, else 8 */
  int64_t waves4_from, waves8_from;
  int64_t cached_maxlen;   /* STRIDED, rows: segments of at most this many observations are shared by cached_waves waves instead (-1: none) */
  int32_t cached_waves;
  int32_t batch;           /* loss partial sums per lane group: STRIDED 1 or 4 (4 applies to one-wave segments only when batch_one_wave_only), WINDOWED 2 or lanes */
  int32_t batch_one_wave_only;
  int32_t rotate;          /* WINDOWED: the lane's fma chain walks its 16-byte chunks in the order i ^ r: 1 = r = (global segment id & 7) >> 1
                              (four-lane groups, conflict-free column passes); 2 = r = (global segment id >> 1) & 7 (the lane-per-segment
                              passes, csrc/glrm_lane.hpp: lanes = 2, comps = kp / 2 -- two chains over the even / odd chunks) */
  int64_t window;          /* WINDOWED: opposing vectors per window */
  int64_t windows_per_sup; /* WINDOWED: windows per super-tile; 0 = the whole list is one super-tile and nothing is re-added */
  int32_t private_order;   /* the engine walks a private copy of the list in another order than the caller's; the order above applies to
                              THAT copy.  1 = the copy was tile-sorted (the caller's list was not ordered by window): not reproducible
                              from the caller's lists.  2 = WINDOWED rows of a model with several loss kinds, lists already ordered by
                              window: inside every window the entries are grouped by ascending glrm_loss.kind of their column, stably
                              (list order inside a kind) -- a function of the caller's list alone, which the oracle restates */
  int32_t long_from;       /* WINDOWED (phase-aligned column passes): > 0 = segments of at least this many observations are swept by the
                              8-wave gather sweep instead and add in the STRIDED order (same lanes / comps, 8 waves, batch 1); 0 = none */
} glrm_sum_order; /* 80 bytes */

/* which: 0 = the row view (X half-step), 1 = the column view (Y half-step).  Needs a finalized handle. */
int glrm_hip_sum_order(glrm_handle* h, int32_t which, glrm_sum_order* out);

#ifdef __cplusplus
}
#endif
#endif /* GLRM_HIP_H */
