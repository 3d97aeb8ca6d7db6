/*
 * c_abi_example.c -- the drop-in boundary used from plain C (what a Julia `ccall` does, minus Julia).
 *
 *   gcc -O2 -I include examples/c_abi_example.c -o build/c_abi_example -ldl
 *   ./build/c_abi_example lowrankmodels.jl_amd/libglrm_hip.so      # MI355X engine
 *   ./build/c_abi_example oracle/libglrm_oracle.so glrm_cpu_       # the CPU oracle exports the same entry points
 *
 * Builds a small GLRM (60 x 40, rank 4, 50 % observed, QuadLoss + QuadReg(0.1)), runs
 * fit!(glrm, ProxGradParams()) through glrm_*_create / glrm_*_fit / glrm_*_objective / glrm_*_destroy and prints the
 * objective trajectory.  The symbols are resolved with dlsym so that the same binary can drive either library.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "glrm_hip.h"

typedef int (*create_fn)(glrm_handle**, const glrm_problem*, const glrm_options*);
typedef int (*fit_fn)(glrm_handle*, const glrm_params*, double*, double*, double*, double*, int64_t, int64_t*);
typedef int (*objective_fn)(glrm_handle*, const double*, const double*, int, double*);
typedef void (*destroy_fn)(glrm_handle*);
typedef const char* (*last_error_fn)(void);

static unsigned long long rng_state = 88172645463325252ull;
static double unif(void) { /* xorshift64*, enough for an example */
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

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "lowrankmodels.jl_amd/libglrm_hip.so";
  const char* prefix = argc > 2 ? argv[2] : "glrm_hip_";
  void* lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { fprintf(stderr, "dlopen(%s): %s\n", path, dlerror()); return 2; }
  create_fn create = (create_fn)sym(lib, prefix, "create");
  fit_fn fit = (fit_fn)sym(lib, prefix, "fit");
  objective_fn objective = (objective_fn)sym(lib, prefix, "objective");
  destroy_fn destroy = (destroy_fn)sym(lib, prefix, "destroy");
  last_error_fn last_error = (last_error_fn)sym(lib, prefix, "last_error");
  /* the structs below are laid out as THIS header declares them: refuse a library built from another revision */
  int (*version)(void) = (int (*)(void))sym(lib, prefix, "version");
  if (version() != GLRM_HIP_ABI_VERSION) { fprintf(stderr, "%s speaks ABI %d, this caller was compiled against ABI %d\n", path, version(), GLRM_HIP_ABI_VERSION); return 2; }

  enum { M = 60, N = 40, K = 4 };
  static double A[M][N], Xs[M][K], Ys[N][K];
  for (int i = 0; i < M; ++i) for (int c = 0; c < K; ++c) Xs[i][c] = 2 * unif() - 1;
  for (int j = 0; j < N; ++j) for (int c = 0; c < K; ++c) Ys[j][c] = 2 * unif() - 1;
  static unsigned char obs[M][N];
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
    double d = 0; for (int c = 0; c < K; ++c) d += Xs[i][c] * Ys[j][c];
    A[i][j] = d + 0.05 * (2 * unif() - 1);
    obs[i][j] = unif() < 0.5;
  }
  /* observed_features -> CSR, observed_examples -> CSC (each in its own order) */
  static int64_t rowptr[M + 1], colptr[N + 1];
  static int32_t colidx[M * N], rowidx[M * N];
  static double rowvals[M * N], colvals[M * N];
  int64_t t = 0;
  for (int i = 0; i < M; ++i) { rowptr[i] = t; for (int j = 0; j < N; ++j) if (obs[i][j]) { colidx[t] = j; rowvals[t] = A[i][j]; ++t; } }
  rowptr[M] = t; t = 0;
  for (int j = 0; j < N; ++j) { colptr[j] = t; for (int i = 0; i < M; ++i) if (obs[i][j]) { rowidx[t] = i; colvals[t] = A[i][j]; ++t; } }
  colptr[N] = t;

  glrm_loss loss = {GLRM_LOSS_QUAD, 0, 1.0, 0.0, 0.0};
  glrm_reg reg = {GLRM_REG_QUAD, 0, 0.1};
  glrm_problem p;
  memset(&p, 0, sizeof p);
  p.m = M; p.n = N; p.k = K; p.row_end = M; p.col_end = N;
  p.rowptr = rowptr; p.colidx = colidx; p.rowvals = rowvals;
  p.colptr = colptr; p.rowidx = rowidx; p.colvals = colvals;
  p.losses = &loss; p.n_losses = 1; p.rx = &reg; p.n_rx = 1; p.ry = &reg; p.n_ry = 1;
  glrm_options o;
  memset(&o, 0, sizeof o);
  o.device_id = -1;
  glrm_handle* h = NULL;
  if (create(&h, &p, &o) != GLRM_OK) { fprintf(stderr, "create: %s\n", last_error()); return 1; }

  static double X[M * K], Y[N * K]; /* k x m and k x n, column-major: x_e and y_f are K contiguous doubles */
  for (int i = 0; i < M * K; ++i) X[i] = 2 * unif() - 1;
  for (int i = 0; i < N * K; ++i) Y[i] = 2 * unif() - 1;
  glrm_params prm = {1.0, 100, 1, 1, 1e-5, 1e-4, 0.01}; /* ProxGradParams() defaults */
  double objs[101], secs[101];
  int64_t nrec = 0;
  if (fit(h, &prm, X, Y, objs, secs, 101, &nrec) != GLRM_OK) { fprintf(stderr, "fit: %s\n", last_error()); return 1; }
  double final_obj = 0;
  if (objective(h, X, Y, 1, &final_obj) != GLRM_OK) { fprintf(stderr, "objective: %s\n", last_error()); return 1; }
  printf("%s: %lld iterations, objective %.6f -> %.6f (loss+ry), full objective %.6f, %.3f ms\n", prefix, (long long)nrec - 1,
         objs[0], objs[nrec - 1], final_obj, 1e3 * secs[nrec - 1]);
  destroy(h);
  return objs[nrec - 1] < objs[0] ? 0 : 1;
}
