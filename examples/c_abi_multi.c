/*
 * c_abi_multi.c -- the multi-GPU entry points of the drop-in boundary from plain C: ONE process, several shards.
 * This is what a single Julia process does through `ccall` when `HipProxGradParams(ngpus = N)` (julia/HipGLRM.jl).
 *
 *   gcc -O2 -I include examples/c_abi_multi.c -o build/c_abi_multi -ldl
 *   ./build/c_abi_multi lowrankmodels.jl_amd/libglrm_hip.so glrm_hip_ 2 0,0   # two shards, both on device 0 (one-GPU box)
 *   ./build/c_abi_multi lowrankmodels.jl_amd/libglrm_hip.so glrm_hip_ 8       # eight shards on devices 0..7
 *   ./build/c_abi_multi oracle/libglrm_oracle.so glrm_cpu_ 3                  # the CPU oracle exports the same entry points
 *
 * Builds a GLRM (400 x 90, rank 8, 40 % observed, QuadLoss, NonNegConstraint on X, QuadReg on Y), fits it once through
 * glrm_*_create / glrm_*_fit (one shard) and once through glrm_*_multi_create / glrm_*_multi_fit (N shards: the library cuts rows
 * and columns, replicates X and Y per device and exchanges the updated blocks after every half-step), and checks that the two
 * trajectories and factors are bit-identical -- sharding only relabels which device runs an independent row or column
 * (src/algorithms/proxgrad_multithread.jl:118,163).
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "glrm_hip.h"

typedef int (*create_fn)(glrm_handle**, const glrm_problem*, const glrm_options*);
typedef int (*fit_fn)(glrm_handle*, const glrm_params*, double*, double*, double*, double*, int64_t, int64_t*);
typedef void (*destroy_fn)(glrm_handle*);
typedef int (*mcreate_fn)(glrm_multi**, const glrm_problem*, const glrm_options*, const glrm_multi_options*);
typedef int (*mfit_fn)(glrm_multi*, const glrm_params*, double*, double*, double*, double*, int64_t, int64_t*);
typedef int (*minfo_fn)(glrm_multi*, int64_t*, int64_t*, int32_t*, double*);
typedef void (*mdestroy_fn)(glrm_multi*);
typedef const char* (*last_error_fn)(void);

static unsigned long long rng_state = 1234567891011ull;
static double unif(void) {
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

static void* sym(void* lib, const char* prefix, const char* name) {
  char buf[128];
  snprintf(buf, sizeof buf, "%s%s", prefix, name);
  void* p = dlsym(lib, buf);
  if (!p) { fprintf(stderr, "missing symbol %s\n", buf); exit(2); }
  return p;
}

enum { M = 400, N = 90, K = 8, MAXSH = 16 };

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "lowrankmodels.jl_amd/libglrm_hip.so";
  const char* prefix = argc > 2 ? argv[2] : "glrm_hip_";
  int nsh = argc > 3 ? atoi(argv[3]) : 2;
  if (nsh < 1 || nsh > MAXSH) { fprintf(stderr, "shards must be in 1..%d\n", MAXSH); return 2; }
  int32_t devs[MAXSH];
  int have_devs = 0;
  if (argc > 4) { /* comma-separated device ordinals, one per shard */
    char* s = argv[4];
    for (int i = 0; i < nsh; ++i) { devs[i] = (int32_t)strtol(s, &s, 10); if (*s == ',') ++s; }
    have_devs = 1;
  }
  void* lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { fprintf(stderr, "dlopen(%s): %s\n", path, dlerror()); return 2; }
  create_fn create = (create_fn)sym(lib, prefix, "create");
  fit_fn fit = (fit_fn)sym(lib, prefix, "fit");
  destroy_fn destroy = (destroy_fn)sym(lib, prefix, "destroy");
  mcreate_fn mcreate = (mcreate_fn)sym(lib, prefix, "multi_create");
  mfit_fn mfit = (mfit_fn)sym(lib, prefix, "multi_fit");
  minfo_fn minfo = (minfo_fn)sym(lib, prefix, "multi_info");
  mdestroy_fn mdestroy = (mdestroy_fn)sym(lib, prefix, "multi_destroy");
  last_error_fn last_error = (last_error_fn)sym(lib, prefix, "last_error");
  /* the structs below are laid out as THIS header declares them: refuse a library built from another revision */
  int (*version)(void) = (int (*)(void))sym(lib, prefix, "version");
  if (version() != GLRM_HIP_ABI_VERSION) { fprintf(stderr, "%s speaks ABI %d, this caller was compiled against ABI %d\n", path, version(), GLRM_HIP_ABI_VERSION); return 2; }

  static double A[M][N], Xs[M][K], Ys[N][K];
  static unsigned char obs[M][N];
  for (int i = 0; i < M; ++i) for (int c = 0; c < K; ++c) Xs[i][c] = unif();
  for (int j = 0; j < N; ++j) for (int c = 0; c < K; ++c) Ys[j][c] = 2 * unif() - 1;
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
    double d = 0; for (int c = 0; c < K; ++c) d += Xs[i][c] * Ys[j][c];
    A[i][j] = d + 0.05 * (2 * unif() - 1);
    obs[i][j] = unif() < 0.4;
  }
  static int64_t rowptr[M + 1], colptr[N + 1];
  static int32_t colidx[M * N], rowidx[M * N];
  static double rowvals[M * N], colvals[M * N];
  int64_t t = 0;
  for (int i = 0; i < M; ++i) { rowptr[i] = t; for (int j = 0; j < N; ++j) if (obs[i][j]) { colidx[t] = j; rowvals[t] = A[i][j]; ++t; } }
  rowptr[M] = t; t = 0;
  for (int j = 0; j < N; ++j) { colptr[j] = t; for (int i = 0; i < M; ++i) if (obs[i][j]) { rowidx[t] = i; colvals[t] = A[i][j]; ++t; } }
  colptr[N] = t;

  glrm_loss loss = {GLRM_LOSS_QUAD, 0, 1.0, 0.0, 0.0};
  glrm_reg rx = {GLRM_REG_NONNEG, 0, 1.0}, ry = {GLRM_REG_QUAD, 0, 0.1};
  glrm_problem p;
  memset(&p, 0, sizeof p);
  p.m = M; p.n = N; p.k = K; p.row_end = M; p.col_end = N;
  p.rowptr = rowptr; p.colidx = colidx; p.rowvals = rowvals;
  p.colptr = colptr; p.rowidx = rowidx; p.colvals = colvals;
  p.losses = &loss; p.n_losses = 1; p.rx = &rx; p.n_rx = 1; p.ry = &ry; p.n_ry = 1;
  glrm_options o;
  memset(&o, 0, sizeof o);
  o.device_id = have_devs ? devs[0] : -1;

  static double X0[M * K], Y0[N * K], X1[M * K], Y1[N * K], X2[M * K], Y2[N * K];
  for (int i = 0; i < M * K; ++i) X0[i] = unif();
  for (int i = 0; i < N * K; ++i) Y0[i] = 2 * unif() - 1;
  glrm_params prm = {1.0, 60, 1, 1, 1e-5, 1e-4, 0.01};
  double o1[61], s1[61], o2[61], s2[61];
  int64_t n1 = 0, n2 = 0;

  glrm_handle* h = NULL;
  if (create(&h, &p, &o) != GLRM_OK) { fprintf(stderr, "create: %s\n", last_error()); return 1; }
  memcpy(X1, X0, sizeof X0); memcpy(Y1, Y0, sizeof Y0);
  if (fit(h, &prm, X1, Y1, o1, s1, 61, &n1) != GLRM_OK) { fprintf(stderr, "fit: %s\n", last_error()); return 1; }
  destroy(h);

  glrm_multi_options mo;
  memset(&mo, 0, sizeof mo);
  mo.n_shards = nsh; mo.device_ids = have_devs ? devs : NULL; mo.x_chunks = 2;
  glrm_multi* mh = NULL;
  if (mcreate(&mh, &p, &o, &mo) != GLRM_OK) { fprintf(stderr, "multi_create: %s\n", last_error()); return 1; }
  memcpy(X2, X0, sizeof X0); memcpy(Y2, Y0, sizeof Y0);
  if (mfit(mh, &prm, X2, Y2, o2, s2, 61, &n2) != GLRM_OK) { fprintf(stderr, "multi_fit: %s\n", last_error()); return 1; }
  int64_t rb[MAXSH + 1], cb[MAXSH + 1];
  int32_t ex = -1;
  double exms = 0;
  if (minfo(mh, rb, cb, &ex, &exms) != GLRM_OK) { fprintf(stderr, "multi_info: %s\n", last_error()); return 1; }
  mdestroy(mh);

  int same = n1 == n2;
  for (int64_t i = 1; same && i < n1; ++i) same = o1[i] == o2[i]; /* objective[0] is summed per column first on the sharded path */
  same = same && memcmp(X1, X2, sizeof X1) == 0 && memcmp(Y1, Y2, sizeof Y1) == 0;
  printf("%s: 1 shard %lld iterations, objective %.6f -> %.9f; %d shards (rows", prefix, (long long)n1 - 1, o1[0], o1[n1 - 1], nsh);
  for (int s = 0; s <= nsh; ++s) printf(" %lld", (long long)rb[s]);
  printf("; cols");
  for (int s = 0; s <= nsh; ++s) printf(" %lld", (long long)cb[s]);
  printf("; exchange %s) %lld iterations -> %.9f: %s\n", ex == 1 ? "rccl" : "direct", (long long)n2 - 1, o2[n2 - 1],
         same ? "BIT-IDENTICAL" : "MISMATCH");
  return same && o1[n1 - 1] < o1[0] ? 0 : 1;
}
