// glrm_engine.hpp -- host-side declarations shared by the translation units of libglrm_hip.so
// (glrm_hip.hip: C ABI + gather sweeps; glrm_tiled.hip: LDS-tiled sweeps; glrm_dense.hip: MFMA path;
// glrm_multi.hip: multi-dimensional losses / block regularizers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/glrm_hip.h"

// kernel variants by loss model.  The *_NOTRIG variants are compiled without PeriodicLoss (its sin / cos with full range reduction
// costs ~60 VGPRs of pressure in every kernel that merely CONTAINS the case); models without a PeriodicLoss column use them.
enum { LOSS_QUAD_UNIFORM = 0, LOSS_SEGMENT = 1, LOSS_PER_OBS = 2, LOSS_SEGMENT_NOTRIG = 3, LOSS_PER_OBS_NOTRIG = 4 };
constexpr int loss_mode(int loss) { return loss >= 3 ? loss - 2 : loss; }
constexpr bool loss_trig(int loss) { return loss < 3; }

// conflict-free tile reads of the LDS-tiled column passes (glrm_tiled.hpp: tile_rot); also read by glrm_hip_sum_order
#ifndef GLRM_TILE_ROT
#define GLRM_TILE_ROT 1
#endif

extern thread_local char g_err[768];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCK(expr)                                                                                   \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail(e_ == hipErrorOutOfMemory ? GLRM_ERR_OOM : GLRM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                         \
  } while (0)

struct glrm_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int64_t m = 0, n = 0;
  int k = 0, kp = 0, G = 4, R = 2;
  int unroll_row = 1, unroll_col = 1, unroll_long = 8;
  // LDS-tiled sweeps (glrm_tiled.hpp)
  bool rows_sorted = false, cols_sorted = false;
  double fixed_alpha = 0.0;           // > 0 while a SparseProxGradParams step (no line search) is being launched
  int64_t rng_b = 0, rng_e = -1;      // row sub-range of the next X sweep (glrm_hip_step_x_range); rng_e < 0 = all
  int tiled_opt = 0;                  // glrm_options.tiled
  int tiled_row = 0, tiled_col = 0;   // 0 = gather sweep, 1 = tiled
  int tG = 4, tR = 2;                 // lane layout of the tiled kernels (kp = tG*tR)
  int tile_cfg = 0;                   // 0: 8 waves + ~64 KB tile, 1: 16 waves + ~128 KB tile
  int tile_lw = 0, tile_lw_sides = 1; // loader waves per workgroup of the double-buffered tiled sweeps; which sides use them
  bool tile_cfg12 = false;            // GLRM_HIP_TILE_CFG=2: heterogeneous row sweep on 12 waves
  int nsup = 0, tiles_per_sup = 0;
  int blocked_row = 0, blocked_col = 0; // phase-aligned gather passes (glrm_blocked.hip) instead of the one-kernel gather sweep
  int row_split = 0, tiles_per_sup_r = 0; // row sweep in super-tile passes (nsup_r super-tiles; buffers part_r ... ntrial_r below)
  int tile_rounds = 0;                // LDS-tiled sweeps: line-search rounds over the still-searching segments only (bit0 rows, bit1 columns)
  int32_t* actlist = nullptr;         // two lists of actlist_cap segment ids (glrm_tiled.hpp: TiledArgs::actlist_out / actlist_in)
  int64_t actlist_cap = 0;
  double *part = nullptr, *gsum = nullptr, *trialbuf = nullptr, *joldbuf = nullptr;
  int32_t *activebuf = nullptr, *ntrialbuf = nullptr;
  unsigned int* nactive = nullptr;
  // lockstep form of the phase-aligned column passes (glrm_blocked.hip: lockstep_col_pass_kernel): 0 off, 1 on; per-XCD window counters
  int lockstep = 0;
  unsigned int* lock_ctr = nullptr;
  int* dflag = nullptr;
  uint8_t* rowdescid = nullptr;       // heterogeneous tiled row sweep: id of the loss descriptor of every entry of the row view
  glrm_loss* udesc = nullptr;         // the model's distinct loss descriptors (<= 256), device
  int n_udesc = 0;
  int32_t *colperm = nullptr, *rowperm = nullptr; // tiled sweeps: segments sorted by (loss kind, length) / by length
  // lane-per-segment LDS-tiled passes (glrm_lane.hpp), [0] rows, [1] columns: family flag and the SELL layout of the view
  int lane[2] = {0, 0};
  int64_t* lane_bptr[2] = {nullptr, nullptr};
  int32_t* lane_off[2] = {nullptr, nullptr};
  double* lane_val[2] = {nullptr, nullptr};
  int64_t lane_nwb[2] = {0, 0}, lane_steps[2] = {0, 0};
  int lane_ntiles[2] = {0, 0};
  uint16_t* lane_off16[2] = {nullptr, nullptr}; // the compact form of the stream (glrm_lane.hpp: LaneArgs::off16 / vptr) of the sides that run it
  int64_t* lane_vptr[2] = {nullptr, nullptr};
  int32_t* lane_sval[2] = {nullptr, nullptr};
  int lane_dealt[2] = {0, 0};                    // the side's sorted slot permutation was dealt out by class (make_segperm): slot s holds a segment of class s & 15
  int32_t* lane_inv[2] = {nullptr, nullptr};   // local segment -> slot of the layout (sides whose slots are permuted); -1 = not in the layout
  // trial rounds read out of the SELL layout (lane_pass_kernel FORM 2): the still-searching segments class by class, rebuilt every round
  int32_t *lane_gcnt = nullptr, *lane_gbase = nullptr, *lane_gtotal = nullptr, *lane_glist = nullptr;
  int64_t lane_gcap = 0, lane_gchunks = 0;
  // general sweeps: multi-dimensional losses / wrapped regularizers (glrm_multi.hip)
  bool multi = false;
  int64_t d = 0;                      // vectors of Y = sum of embedding dimensions (= n for scalar losses)
  int dmax = 1;
  int multi_kmask = 0;                // loss kinds of the model: bit `kind` per multi-dimensional kind, bit 0 = some scalar loss (glrm_multi.hpp: MULTI_KM_*)
  int64_t* ystart = nullptr;          // device, n+1
  int64_t ys_cb = 0;                  // ystart[cb]: first vector of the shard's column block
  int col_nsplit = 1;                 // > 1: the Y half-step runs as split passes + decide rounds
  int64_t col_chunk = 0;
  double *mtrial = nullptr, *mpart_loss = nullptr, *mpart_G = nullptr, *mgtot = nullptr, *mobjold = nullptr;
  int32_t* mactive = nullptr;
  unsigned int* mnactive = nullptr;
  // dense MFMA path (glrm_dense.hip)
  bool dense = false;
  double *Arow = nullptr, *Acol = nullptr; // packed, zero padded: [ml_pad][lda_r], [nl_pad][lda_c]
  int64_t lda_r = 0, lda_c = 0;
  double dense_scale = 1.0;
  int nsup_r = 1, nsup_c = 1;
  int64_t vps_r = 0, vps_c = 0;           // opposing vectors per super-tile
  double *part_r = nullptr, *gsum_r = nullptr, *trial_r = nullptr, *jold_r = nullptr;
  int32_t *active_r = nullptr, *ntrial_r = nullptr;
  int cached_row = 0, cached_cap = 0; // cached gather row sweep (glrm_cached.hip): 0 off, 1 LDS, 2 registers; trips / vectors a row may have
  int cached_want = 0;                // the whole problem runs its short rows on the cached sweep (decided from glrm_signature, never from the shard)
  // glrm_options.quad_gram: trials from the quadratic form (glrm_dense.hpp: dense_gram_*)
  bool dense_gram = false;
  double *gramH = nullptr, *gram_part = nullptr; // [kp*kp], [GRAM_BLOCKS][kp*kp]
  double *jloss_r = nullptr, *jloss_c = nullptr; // loss sum at the current point, per local segment
  int64_t rb = 0, re = 0, cb = 0, ce = 0, ml = 0, nl = 0, nnz_r = 0, nnz_c = 0;
  int64_t *rowptr = nullptr, *colptr = nullptr;
  int32_t *colidx = nullptr, *rowidx = nullptr;
  double *rowvals = nullptr, *colvals = nullptr;
  // GLRM_PROBLEM_BORROW_DEVICE_ARRAYS: false = the array above is the caller's (never freed, never written by the engine)
  bool own_ptrs = true, own_rowview = true, own_colview = true;
  glrm_loss* losses = nullptr;
  int64_t n_losses = 0;
  bool loss_quad_uniform = false;
  bool has_trig = false; // some column carries a PeriodicLoss
  glrm_reg *rx = nullptr, *ry = nullptr;
  int64_t n_rx = 0, n_ry = 0;
  double *alpharow = nullptr, *alphacol = nullptr;
  double *X = nullptr, *Y = nullptr, *objcol = nullptr, *objrow = nullptr;             // in use (bound or owned)
  double *oX = nullptr, *oY = nullptr, *oobjcol = nullptr, *oobjrow = nullptr;         // owned
  double *partials = nullptr, *dscalar = nullptr;
  unsigned long long* dcount = nullptr;
  int32_t *trials_r = nullptr, *accepts_r = nullptr, *trials_c = nullptr, *accepts_c = nullptr;
  // Gather sweeps: the waves that share a segment are a function of the segment's OWN length (1 below 1536 observations, 4 below
  // 98304, else 8; glrm_options.waves_* pins one count for all), so the order of a segment's sums never depends on which shard
  // holds it.  Class 0 = rows the cached sweep holds on chip.  ncls_* counts the local segments per class; when more than one class
  // is populated seglist_* lists the segments class by class (ascending ids inside a class) and each class gets its own launch.
  int waves_row = 1, waves_col = 4;   // the class most local segments fall into (reported by glrm_hip_kernel_stats)
  int32_t *seglist_r = nullptr, *seglist_c = nullptr;
  int64_t ncls_r[4] = {0, 0, 0, 0}, ncls_c[4] = {0, 0, 0, 0};
  bool finalized = false;             // false between a GLRM_PROBLEM_DEFER_SETUP create and glrm_hip_finalize
  bool finalize_failed = false;       // glrm_hip_finalize ran and failed half way: the handle can only be destroyed
  // launch geometry of the persistent / sliced kernels, per handle (device and fill percentage at finalize; a process may drive devices
  // with different CU counts, and the knobs are read per handle): 0 = not computed yet
  // phase-aligned column passes on skewed data: slot -> column for the passes (columns below blk_long_from observations, longest
  // first), the columns at or above it (8-wave gather sweep on the side stream), and the threshold (from the WHOLE problem's mean)
  int32_t *blk_perm_c = nullptr, *blk_long_c = nullptr;
  int64_t blk_nshort_c = 0, blk_nlong_c = 0, blk_long_from = 0;
  int64_t blocked_cap[2][2] = {{0, 0}, {0, 0}}; // phase-aligned passes: segments per launch slice, [row / column view][gradient / trial instantiation]
  int cached_grid[2] = {0, 0};        // persistent cached row sweep: resident workgroups, MAXT = 7 / 4 instantiation
  glrm_signature sig_local{}, sig{};  // this shard's contribution / the whole problem's
  int order_unit = 0;                 // opposing vectors per unit of the tile-order check (glrm_tiled.hip)
  hipStream_t side_stream = nullptr;  // the launches of the minority classes run beside the main launch
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int profile = 0;
  // glrm_hip_step_y_arrival: the blocks of X that are still arriving (valid during that call only), which of them the launch stream has
  // already been made to wait for, and -- phase-aligned column passes -- the order in which the gradient pass walks the super-tiles
  const glrm_arrival* arrival = nullptr;
  int n_arrival = 0;
  std::vector<char> arrival_waited;
  bool arrival_static = false; // true arrival order gave up once (an event that cannot be queried, 5 s without progress): in-stream waits from then on
  std::vector<int> sup_order;
  double ms_wait = 0;                 // profile: time the launch stream stood in those waits
  int sum_order_opt = 0;              // glrm_options.sum_order (1: reference-order validation sweeps, glrm_reforder.hip)
  // hipGraph of one outer iteration (gather sweeps on a private stream): small fits are launch bound
  hipGraph_t iter_graph = nullptr;
  hipGraphExec_t iter_exec = nullptr;
  double* pinned_obj = nullptr;       // host-pinned scalar the graph copies sum(obj_by_col) into
  int64_t graph_ix = 0, graph_iy = 0; // inner iteration counts the graph was captured for
  double graph_min = 0.0, graph_step = 0.0;
  // what glrm_hip_subset needs to build a child handle: the options and host copies of the descriptors
  glrm_options opts{};
  int wr_opt = 0, wc_opt = 0;
  std::vector<glrm_loss> losses_h;
  std::vector<glrm_reg> rx_h, ry_h;
  struct Ev { hipEvent_t a, b; int which; };
  std::vector<Ev> pending, pool;
  int64_t launches_x = 0, launches_y = 0;
  double ms_x = 0, ms_y = 0;
};


inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}


// per-segment class of the gather sweeps (see glrm_handle::seglist_r)
constexpr int64_t GLRM_WAVES4_FROM = 1536, GLRM_WAVES8_FROM = 98304;
__host__ __device__ inline int glrm_wave_class(int64_t len) { return len < GLRM_WAVES4_FROM ? 1 : (len < GLRM_WAVES8_FROM ? 2 : 3); }
constexpr int glrm_class_waves(int cls) { return cls <= 1 ? 1 : (cls == 2 ? 4 : 8); }
// rows the register-cached sweep takes: at most 13 trips of the lane layout, 64 / G observations each
inline int64_t glrm_cached_reg_maxlen(int G) { return (int64_t)13 * (64 / G); }

// LDS-tiled sweeps (glrm_tiled.hip).  prepare: tile configuration + the tile-order check of this shard's lists (create);
// setup: family choice from h->sig and buffers (finalize)
int glrm_prepare_tiled(glrm_handle* h);
int glrm_setup_tiled(glrm_handle* h);
int glrm_run_tiled(glrm_handle* h, bool rows, int loss, int loss_by_segment, double min_stepsize, int eval_only);

// lane-per-segment LDS-tiled passes (glrm_lane.hip): setup decides lane[] from the whole problem's signature and builds the SELL layouts;
// run takes the TiledArgs glrm_run_tiled prepared (glrm_tiled.hpp)
namespace glrm { struct TiledArgs; }
bool glrm_lane_wants(const glrm_handle* h, bool rows);   // the whole problem's shape / losses / options admit the family on that side (given that the side runs the LDS tiles)
int glrm_setup_lane(glrm_handle* h);
bool glrm_lane_loss_ok(const glrm_handle* h, int loss);
bool glrm_lane_few_descriptors(const glrm_handle* h);
int glrm_run_lane(glrm_handle* h, bool rows, int loss, const glrm::TiledArgs& a, double min_stepsize, int eval_only);

// phase-aligned gather passes (glrm_blocked.hip): setup decides blocked_row / blocked_col and allocates the pass buffers
int glrm_setup_blocked(glrm_handle* h);
int glrm_run_blocked(glrm_handle* h, bool rows, int loss, int loss_by_segment, double min_stepsize, int eval_only);

// glrm_hip_step_y_arrival (glrm_hip.hip): make h->stream wait for every block of h->arrival that intersects rows [lo, hi) of X and has
// not been waited for yet
int glrm_arrival_wait(glrm_handle* h, int64_t lo, int64_t hi);
// LDS-tiled / lane column passes under glrm_hip_step_y_arrival (round 6): runs of super-tiles in the order their rows of X are announced --
// launch(sup_lo, sup_hi) is called once per run, behind the in-stream waits for the blocks the run touches; without arrival blocks: one call
// over all super-tiles.  Partial sums are per (column, super-tile) and are reduced in super-tile order: which run went first changes no bit.
#include <functional>
int glrm_for_sup_runs_in_arrival_order(glrm_handle* h, int nsup, int64_t rows_per_sup, const std::function<int(int, int)>& launch);

// stable segmented sort of a view by tile index (glrm_tilesort.hip)
int glrm_tile_sort_view(hipStream_t st, const int64_t* ptr, int64_t nseg, int64_t nnz, int tile, int64_t n_other, int32_t** idx, double** vals, bool free_old);

// reference-order validation sweeps (glrm_reforder.hip; glrm_options.sum_order = 1)
// test hooks (csrc/glrm_testhooks.hip): constant in the product library, environment-driven in the test build (-DGLRM_HIP_TESTING)
int glrm_test_fail_finalize();                        // 1 = glrm_hip_finalize fails half way (the set-up-failed latch cannot be reached otherwise)
int glrm_setup_reforder(glrm_handle* h);               // finalize: refuses what the mode does not cover
int glrm_run_reforder(glrm_handle* h, bool rows, double min_stepsize, int eval_only);
int glrm_reforder_sum(glrm_handle* h, const void* dvec, int64_t n, double* out); // Julia's pairwise sum(::Vector{Float64})
int glrm_reforder_objective(glrm_handle* h, int include_reg, double* out);       // objective(): ONE accumulator over all observations, then the penalties

// GLRM_PROBLEM_ROWS_FROM_COLS (glrm_transpose.hip): the row view derived on the device from the uploaded column view
int glrm_rows_from_cols(glrm_handle* h);

// cached gather row sweep (glrm_cached.hip)
int glrm_setup_cached(glrm_handle* h);                 // finalize: cached_want / cached_row from h->sig
int64_t glrm_cached_maxlen(const glrm_handle* h);      // longest row the cached sweep takes (a function of k and the variant)
void glrm_cached_set_cap(glrm_handle* h, int64_t maxlen);
int glrm_run_cached(glrm_handle* h, int loss, double min_stepsize, const int32_t* seglist, int64_t nlist, hipStream_t st);
// dense MFMA path (glrm_dense.hip)
int glrm_setup_dense(glrm_handle* h, const glrm_problem* p);
int glrm_run_dense(glrm_handle* h, bool rows, double min_stepsize, int eval_only);

// general sweeps (glrm_multi.hip)
int glrm_setup_multi(glrm_handle* h, const glrm_problem* p);
int glrm_run_multi(glrm_handle* h, bool rows, double min_stepsize, int eval_only);
int glrm_run_multi_penalty(glrm_handle* h, bool rows);

// per-segment reduce (which = 0) / decide (which = 1) kernels of glrm_tiled.hpp for a kp-wide factor
namespace glrm { struct TiledArgs; }
void glrm_launch_col_small(int kp, int which, const glrm::TiledArgs& a, hipStream_t st);
