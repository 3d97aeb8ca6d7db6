// glrm_svd.hip -- glrm_hip_init_svd: init_svd!(glrm) (src/initialize.jl:35-132) on the resident observation lists.
//
// The reference expands A to an m x d real matrix (categorical columns -> +-1 indicators per level, multi-dimensional ordinal
// columns -> +-1 threshold indicators, :47-80), subtracts the per-column mean of the observed entries (:83-98), leaves unobserved
// entries at 0, scales by m*n/|Omega| (:113) and takes the top-k singular triplets with Arpack (:121):  X = sqrt(S) U',
// Y = sqrt(S) V' diag(std) (:129-130).  Here the standardized matrix B is never materialised: both products of a randomized
// subspace iteration, U <- B V (row view) and V <- B' U (column view), expand the observations on the fly; the tall-skinny
// blocks are orthonormalized through their l x l Gram matrix (eigen-decomposition on the host, l = k + oversampling <= 128),
// and the Ritz values of B'U give the singular values.  Everything is fp64 and every reduction has a fixed order.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "glrm_engine.hpp"

namespace {

constexpr int LMAX = 128;

struct SvdArgs {
  int64_t nseg;
  const int64_t* ptr;
  const int32_t* idx;
  const double* vals;
  const glrm_loss* losses;
  int loss_single;
  const int64_t* ystart;
  const double* means;  // [d]
  double cscale;        // m*n/|Omega_rows|
  const double* in;     // rows: V [d][l];  cols: U [m][l]
  double* out;          // rows: U [m][l];  cols: V [d][l] (or partials)
  int l;
  int nsplit;
  int64_t chunk;
  double* partial;      // cols with nsplit > 1: [nseg][nsplit][dmax][l]
  int dmax;
};

// entry (a, j) of the real-valued expansion of one observed value (src/initialize.jl:52-77)
__device__ __forceinline__ double areal(int kind, int d, double a, int j) {
  if (d <= 1) return a;
  if (kind == GLRM_LOSS_MULTINOMIAL || kind == GLRM_LOSS_OVA) return a == (double)(j + 1) ? 1.0 : -1.0; // CategoricalDomain
  const int nlev = kind == GLRM_LOSS_ORDISTIC ? d : d + 1;                                                // OrdinalDomain: 1..max
  return j < nlev - 1 ? (a > (double)(j + 1) ? 1.0 : -1.0) : 0.0;
}

// per expanded column: mean and corrected standard deviation of the observed entries (:83-95); one workgroup per column
__global__ void __launch_bounds__(256) svd_stats_kernel(const int64_t* colptr, const double* colvals, const glrm_loss* losses, int loss_single,
                                                        const int64_t* ystart, int64_t nl, double tol, double* means, double* stds) {
  const int64_t f = blockIdx.x;
  const glrm_loss lo = losses[loss_single ? 0 : f];
  const int d = lo.dim > 1 ? lo.dim : 1;
  const int64_t b = colptr[f], e = colptr[f + 1], ys = ystart[f];
  __shared__ double sh[256];
  for (int j = 0; j < d; ++j) {
    double s = 0.0;
    for (int64_t t = b + threadIdx.x; t < e; t += 256) s += areal(lo.kind, d, colvals[t], j);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    double mean = sh[0] / (double)(e - b); // 0/0 = NaN for an empty column
    __syncthreads();
    double q = 0.0;
    for (int64_t t = b + threadIdx.x; t < e; t += 256) { const double x = areal(lo.kind, d, colvals[t], j) - mean; q = fma(x, x, q); }
    sh[threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) {
      double sd = sqrt(sh[0] / (double)(e - b - 1));
      if (mean != mean) mean = 1.0;             // isnan(means[j]) -> 1 (:88-90)
      if (sd < tol || sd != sd) sd = 1.0;       // :92-94
      means[ys + j] = mean;
      stds[ys + j] = sd;
    }
    __syncthreads();
  }
}

// U[e, :] = cscale * sum_{(f,a) in Omega_e} sum_j (areal(a,j) - mean_j) V[ys_f + j, :]      one wave per row
__global__ void __launch_bounds__(256) svd_rows_kernel(const SvdArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= a.nseg) return;
  const int l = a.l;
  double acc0 = 0.0, acc1 = 0.0;
  for (int64_t t = a.ptr[e]; t < a.ptr[e + 1]; ++t) {
    const int32_t f = a.idx[t];
    const double av = a.vals[t];
    const glrm_loss lo = a.losses[a.loss_single ? 0 : f];
    const int d = lo.dim > 1 ? lo.dim : 1;
    const int64_t ys = a.ystart[f];
    for (int j = 0; j < d; ++j) {
      const double v = areal(lo.kind, d, av, j) - a.means[ys + j];
      const double* vr = a.in + (ys + j) * l;
      if (lane < l) acc0 = fma(v, vr[lane], acc0);
      if (lane + 64 < l) acc1 = fma(v, vr[lane + 64], acc1);
    }
  }
  if (lane < l) a.out[e * l + lane] = a.cscale * acc0;
  if (lane + 64 < l) a.out[e * l + lane + 64] = a.cscale * acc1;
}

// V[ys_f + j, :] = cscale * sum_{(e,a) in Omega^f} (areal(a,j) - mean_j) U[e, :]     workgroup (4 waves) per (column, chunk)
__global__ void __launch_bounds__(256) svd_cols_kernel(const SvdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t f = blockIdx.x;
  const int y = blockIdx.y;
  const int l = a.l;
  const glrm_loss lo = a.losses[a.loss_single ? 0 : f];
  const int d = lo.dim > 1 ? lo.dim : 1;
  const int64_t ys = a.ystart[f];
  int64_t b = a.ptr[f] + (int64_t)y * a.chunk, e = b + a.chunk;
  const int64_t e0 = a.ptr[f + 1];
  b = b < e0 ? b : e0;
  e = e < e0 ? e : e0;
  __shared__ double sh[4][LMAX];
  for (int j = 0; j < d; ++j) {
    const double mean = a.means[ys + j];
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t t = b + wave; t < e; t += 4) {
      const double v = areal(lo.kind, d, a.vals[t], j) - mean;
      const double* ur = a.in + (int64_t)a.idx[t] * l;
      if (lane < l) acc0 = fma(v, ur[lane], acc0);
      if (lane + 64 < l) acc1 = fma(v, ur[lane + 64], acc1);
    }
    if (lane < l) sh[wave][lane] = acc0;
    if (lane + 64 < l) sh[wave][lane + 64] = acc1;
    __syncthreads();
    for (int c = threadIdx.x; c < l; c += 256) {
      const double s = ((sh[0][c] + sh[1][c]) + sh[2][c]) + sh[3][c];
      if (a.nsplit > 1) a.partial[(((size_t)f * a.nsplit + y) * a.dmax + j) * l + c] = s;
      else a.out[(ys + j) * l + c] = a.cscale * s;
    }
    __syncthreads();
  }
}

__global__ void svd_cols_reduce_kernel(const SvdArgs a) { // chunk partials in chunk order
  const int64_t f = blockIdx.x;
  const int l = a.l;
  const glrm_loss lo = a.losses[a.loss_single ? 0 : f];
  const int d = lo.dim > 1 ? lo.dim : 1;
  const int64_t ys = a.ystart[f];
  for (int i = threadIdx.x; i < d * l; i += blockDim.x) {
    const int j = i / l, c = i - j * l;
    double s = 0.0;
    for (int y = 0; y < a.nsplit; ++y) s += a.partial[(((size_t)f * a.nsplit + y) * a.dmax + j) * l + c];
    a.out[(ys + j) * l + c] = a.cscale * s;
  }
}

// partial Gram matrices Z'Z of row blocks: thread t owns entries t, t+256, ... of the l x l matrix
constexpr int GROWS = 32;
__global__ void __launch_bounds__(256) gram_kernel(const double* Z, int64_t nrows, int l, int64_t rows_per_block, double* part) {
  __shared__ double tile[GROWS][LMAX + 1];
  const int l2 = l * l;
  double acc[LMAX * LMAX / 256];
#pragma unroll
  for (int i = 0; i < LMAX * LMAX / 256; ++i) acc[i] = 0.0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < nrows ? r0 + rows_per_block : nrows;
  for (int64_t r = r0; r < r1; r += GROWS) {
    __syncthreads();
    for (int i = threadIdx.x; i < GROWS * l; i += 256) {
      const int rr = i / l, c = i - rr * l;
      tile[rr][c] = r + rr < r1 ? Z[(r + rr) * l + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LMAX * LMAX / 256; ++i) {
      const int ent = i * 256 + threadIdx.x;
      if (ent < l2) {
        const int p = ent / l, q = ent - p * l;
        double s = acc[i];
        for (int rr = 0; rr < GROWS; ++rr) s = fma(tile[rr][p], tile[rr][q], s);
        acc[i] = s;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LMAX * LMAX / 256; ++i) {
    const int ent = i * 256 + threadIdx.x;
    if (ent < l2) part[(size_t)blockIdx.x * l2 + ent] = acc[i];
  }
}

__global__ void gram_reduce_kernel(const double* part, int nblk, int l2, double* G) {
  for (int ent = blockIdx.x * blockDim.x + threadIdx.x; ent < l2; ent += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * l2 + ent];
    G[ent] = s;
  }
}

// Z <- Z * M (l x l, row-major), one wave per row
__global__ void __launch_bounds__(256) rightmul_kernel(double* Z, int64_t nrows, int l, const double* M) {
  __shared__ double zr[4][LMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r < nrows) {
    if (lane < l) zr[wave][lane] = Z[r * l + lane];
    if (lane + 64 < l) zr[wave][lane + 64] = Z[r * l + lane + 64];
  }
  __syncthreads();
  if (r >= nrows) return;
  double o0 = 0.0, o1 = 0.0;
  for (int p = 0; p < l; ++p) {
    const double z = zr[wave][p];
    if (lane < l) o0 = fma(z, M[p * l + lane], o0);
    if (lane + 64 < l) o1 = fma(z, M[p * l + lane + 64], o1);
  }
  if (lane < l) Z[r * l + lane] = o0;
  if (lane + 64 < l) Z[r * l + lane + 64] = o1;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void randn_kernel(double* V, int64_t count, uint64_t seed) { // Box-Muller on a counter-based hash
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const double u1 = ((double)(mix64(seed ^ mix64((uint64_t)i * 2 + 1)) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(mix64(seed ^ mix64((uint64_t)i * 2 + 2)) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    V[i] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
  }
}

// X[c, e] = sqrt(s_c) U[e, c];  Y[c, j] = sqrt(s_c) V[j, c] std_j   (k x count, column-major like glrm.X / glrm.Y)
__global__ void factors_kernel(const double* Z, int64_t count, int l, int k, const double* sqrt_s, const double* stds, double* F) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count * k; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / k;
    const int c = (int)(i - r * k);
    F[i] = sqrt_s[c] * Z[r * l + c] * (stds ? stds[r] : 1.0);
  }
}

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major); eigenvalues descending, eigenvectors in the
// columns of E.  n <= 128: a few hundred microseconds on the host.
void jacobi_eigh(std::vector<double>& A, int n, std::vector<double>& w, std::vector<double>& E) {
  E.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) E[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) (i == j ? diag : off) += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int r = 0; r < n; ++r) { // columns p, q
          const double arp = A[(size_t)r * n + p], arq = A[(size_t)r * n + q];
          A[(size_t)r * n + p] = c * arp - s * arq;
          A[(size_t)r * n + q] = s * arp + c * arq;
        }
        for (int r = 0; r < n; ++r) { // rows p, q
          const double apr = A[(size_t)p * n + r], aqr = A[(size_t)q * n + r];
          A[(size_t)p * n + r] = c * apr - s * aqr;
          A[(size_t)q * n + r] = s * apr + c * aqr;
        }
        for (int r = 0; r < n; ++r) {
          const double erp = E[(size_t)r * n + p], erq = E[(size_t)r * n + q];
          E[(size_t)r * n + p] = c * erp - s * erq;
          E[(size_t)r * n + q] = s * erp + c * erq;
        }
      }
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return A[(size_t)x * n + x] > A[(size_t)y * n + y]; });
  w.resize(n);
  std::vector<double> Es((size_t)n * n);
  for (int c = 0; c < n; ++c) {
    w[c] = A[(size_t)order[c] * n + order[c]];
    for (int r = 0; r < n; ++r) Es[(size_t)r * n + c] = E[(size_t)r * n + order[c]];
  }
  E.swap(Es);
}

struct Work {
  glrm_handle* h;
  hipStream_t st;
  int l;
  double *V = nullptr, *U = nullptr, *means = nullptr, *stds = nullptr, *gpart = nullptr, *G = nullptr, *M = nullptr, *cpart = nullptr,
         *sq = nullptr, *dX = nullptr, *dY = nullptr;
  int nblk_max = 1024;
  ~Work() {
    for (double* p : {V, U, means, stds, gpart, G, M, cpart, sq, dX, dY})
      if (p) (void)hipFree(p);
  }
};

// G = Z'Z -> eigenpairs on the host.  Returns descending eigenvalues w and eigenvectors E (row-major, columns).
int gram_eig(Work& w, const double* Z, int64_t nrows, std::vector<double>& ev, std::vector<double>& E) {
  const int l = w.l, l2 = l * l;
  int nblk = (int)std::min<int64_t>(w.nblk_max, (nrows + 255) / 256);
  if (nblk < 1) nblk = 1;
  const int64_t rpb = ((nrows + nblk - 1) / nblk + GROWS - 1) / GROWS * GROWS;
  nblk = (int)((nrows + rpb - 1) / rpb);
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(gram_kernel, dim3(nblk), dim3(256), 0, w.st, Z, nrows, l, rpb, w.gpart);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(16), dim3(256), 0, w.st, w.gpart, nblk, l2, w.G);
  std::vector<double> Gh((size_t)l2);
  HIPCK(hipMemcpyAsync(Gh.data(), w.G, (size_t)l2 * 8, hipMemcpyDeviceToHost, w.st));
  HIPCK(hipStreamSynchronize(w.st));
  for (int i = 0; i < l; ++i) // symmetrise against rounding
    for (int j = i + 1; j < l; ++j) { const double s = 0.5 * (Gh[(size_t)i * l + j] + Gh[(size_t)j * l + i]); Gh[(size_t)i * l + j] = Gh[(size_t)j * l + i] = s; }
  jacobi_eigh(Gh, l, ev, E);
  return GLRM_OK;
}

int rightmul(Work& w, double* Z, int64_t nrows, const std::vector<double>& Mh) {
  HIPCK(hipMemcpyAsync(w.M, Mh.data(), (size_t)w.l * w.l * 8, hipMemcpyHostToDevice, w.st));
  hipLaunchKernelGGL(rightmul_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, w.st, Z, nrows, w.l, w.M);
  HIPCK(hipGetLastError());
  HIPCK(hipStreamSynchronize(w.st)); // Mh may go out of scope
  return GLRM_OK;
}

// Z <- orthonormal basis of its column space (directions with a vanishing Gram eigenvalue become zero columns).
// Pass 1 maps Z to Z E L^-1/2: the columns come out along the principal directions of Z, by descending Gram eigenvalue
// (`ritz`, `Eout`).  Pass 2 removes the rounding left by pass 1 with the symmetric inverse square root E L^-1/2 E', which is
// the identity up to rounding and therefore keeps that order.
int orthonormalize(Work& w, double* Z, int64_t nrows, std::vector<double>* ritz = nullptr, std::vector<double>* Eout = nullptr) {
  std::vector<double> ev, E;
  const int l = w.l;
  for (int pass = 0; pass < 2; ++pass) {
    int rc = gram_eig(w, Z, nrows, ev, E);
    if (rc) return rc;
    if (pass == 0 && ritz) { *ritz = ev; if (Eout) *Eout = E; }
    std::vector<double> M((size_t)l * l, 0.0);
    const double thr = ev[0] * 1e-28;
    for (int c = 0; c < l; ++c) {
      const double s = ev[c] > thr && ev[c] > 0 ? 1.0 / std::sqrt(ev[c]) : 0.0;
      if (pass == 0) {
        for (int r = 0; r < l; ++r) M[(size_t)r * l + c] = E[(size_t)r * l + c] * s;
      } else {
        for (int r = 0; r < l; ++r)
          for (int q = 0; q < l; ++q) M[(size_t)r * l + q] += E[(size_t)r * l + c] * s * E[(size_t)q * l + c];
      }
    }
    if ((rc = rightmul(w, Z, nrows, M))) return rc;
  }
  return GLRM_OK;
}

} // namespace

extern "C" int glrm_hip_init_svd(glrm_handle* h, double* X, double* Y, int32_t max_iter, double tol, uint64_t seed, double* singular_values,
                                 int32_t* iters_done) {
  if (!h || !X || !Y) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (!(h->rb == 0 && h->re == h->m && h->cb == 0 && h->ce == h->n)) return fail(GLRM_ERR_INVALID, "glrm_hip_init_svd needs a single-shard handle");
  if (h->dense) return fail(GLRM_ERR_UNSUPPORTED, "glrm_hip_init_svd works on the observation lists (create the handle without dense_A)");
  if (!h->finalized) return fail(GLRM_ERR_INVALID, "the handle was created with GLRM_PROBLEM_DEFER_SETUP: call glrm_hip_finalize first");
  const int k = h->k;
  const int64_t m = h->m, d = h->d;
  if (k > m || k > d) return fail(GLRM_ERR_INVALID, "k = %d exceeds min(m, d): no k singular triplets", k);
  if (hipSetDevice(h->device) != hipSuccess) return fail(GLRM_ERR_HIP, "cannot select device %d", h->device);
  if (max_iter <= 0) max_iter = 60;
  if (!(tol > 0)) tol = 1e-12;
  int l = k + 8;
  if (l > LMAX) l = LMAX;
  if (l > m) l = (int)m;
  if (l > d) l = (int)d;
  if (l < k) return fail(GLRM_ERR_UNSUPPORTED, "glrm_hip_init_svd supports k <= %d", LMAX);
  Work w;
  w.h = h; w.st = h->stream; w.l = l;
  hipStream_t st = w.st;
  HIPCK(hipMalloc((void**)&w.V, (size_t)d * l * 8));
  HIPCK(hipMalloc((void**)&w.U, (size_t)m * l * 8));
  HIPCK(hipMalloc((void**)&w.means, (size_t)d * 8));
  HIPCK(hipMalloc((void**)&w.stds, (size_t)d * 8));
  HIPCK(hipMalloc((void**)&w.gpart, (size_t)w.nblk_max * l * l * 8));
  HIPCK(hipMalloc((void**)&w.G, (size_t)l * l * 8));
  HIPCK(hipMalloc((void**)&w.M, (size_t)l * l * 8));
  HIPCK(hipMalloc((void**)&w.sq, (size_t)l * 8));
  // column statistics
  hipLaunchKernelGGL(svd_stats_kernel, dim3((unsigned)h->n), dim3(256), 0, st, h->colptr, h->colvals, h->losses, h->n_losses == 1 ? 1 : 0,
                     h->ystart, h->n, 1e-10, w.means, w.stds);
  HIPCK(hipGetLastError());
  SvdArgs ra{}, ca{};
  ra.nseg = m; ra.ptr = h->rowptr; ra.idx = h->colidx; ra.vals = h->rowvals;
  ra.losses = h->losses; ra.loss_single = h->n_losses == 1 ? 1 : 0; ra.ystart = h->ystart; ra.means = w.means;
  ra.cscale = h->nnz_r > 0 ? (double)m * (double)h->n / (double)h->nnz_r : 0.0; // Astd *= m*n/sum(map(length, observed_features)) (:113)
  ra.in = w.V; ra.out = w.U; ra.l = l; ra.nsplit = 1; ra.dmax = h->dmax;
  ca = ra;
  ca.nseg = h->n; ca.ptr = h->colptr; ca.idx = h->rowidx; ca.vals = h->colvals; ca.in = w.U; ca.out = w.V;
  { // long columns: several workgroups per column, partials added in chunk order
    std::vector<int64_t> cp((size_t)h->n + 1);
    HIPCK(hipMemcpyAsync(cp.data(), h->colptr, ((size_t)h->n + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    int64_t longest = 0;
    for (int64_t f = 0; f < h->n; ++f) longest = std::max(longest, cp[f + 1] - cp[f]);
    ca.chunk = 16384;
    while ((longest + ca.chunk - 1) / ca.chunk > 65535) ca.chunk *= 2; // gridDim.y
    ca.nsplit = (int)std::max<int64_t>(1, (longest + ca.chunk - 1) / ca.chunk);
    if (ca.nsplit > 1) {
      HIPCK(hipMalloc((void**)&w.cpart, (size_t)h->n * ca.nsplit * h->dmax * l * 8));
      ca.partial = w.cpart;
    } else ca.chunk = longest > 0 ? longest : 1;
  }
  auto BV = [&]() { // U = B V
    hipLaunchKernelGGL(svd_rows_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, ra);
  };
  auto BtU = [&]() { // V = B' U
    hipLaunchKernelGGL(svd_cols_kernel, dim3((unsigned)h->n, (unsigned)ca.nsplit), dim3(256), 0, st, ca);
    if (ca.nsplit > 1) hipLaunchKernelGGL(svd_cols_reduce_kernel, dim3((unsigned)h->n), dim3(256), 0, st, ca);
  };
  hipLaunchKernelGGL(randn_kernel, dim3(1024), dim3(256), 0, st, w.V, d * l, seed);
  int rc = orthonormalize(w, w.V, d);
  if (rc) return rc;
  std::vector<double> ritz, prev, E;
  int it = 0;
  for (it = 1; it <= max_iter; ++it) {
    BV();
    if ((rc = orthonormalize(w, w.U, m))) return rc;
    BtU();
    HIPCK(hipGetLastError());
    if ((rc = orthonormalize(w, w.V, d, &ritz, &E))) return rc; // eigenvalues of (B'U)'(B'U) = squared singular values
    bool done = !prev.empty();
    for (int c = 0; c < k && done; ++c) {
      const double s = std::sqrt(std::max(ritz[c], 0.0)), sp = std::sqrt(std::max(prev[c], 0.0)), s0 = std::sqrt(std::max(ritz[0], 0.0));
      if (std::fabs(s - sp) > tol * (s0 > 0 ? s0 : 1.0)) done = false;
    }
    prev = ritz;
    if (done) break;
  }
  if (it > max_iter) it = max_iter;
  // After the last pass: V = orth(B'U) = right singular vectors (columns by descending value); left ones = U E.
  if ((rc = rightmul(w, w.U, m, E))) return rc;
  std::vector<double> sq((size_t)l);
  for (int c = 0; c < l; ++c) sq[c] = std::sqrt(std::sqrt(std::max(ritz[c], 0.0))); // sqrt of the singular value
  HIPCK(hipMemcpyAsync(w.sq, sq.data(), (size_t)l * 8, hipMemcpyHostToDevice, st));
  HIPCK(hipMalloc((void**)&w.dX, (size_t)m * k * 8));
  HIPCK(hipMalloc((void**)&w.dY, (size_t)d * k * 8));
  hipLaunchKernelGGL(factors_kernel, dim3(1024), dim3(256), 0, st, w.U, m, l, k, w.sq, (const double*)nullptr, w.dX);
  hipLaunchKernelGGL(factors_kernel, dim3(1024), dim3(256), 0, st, w.V, d, l, k, w.sq, w.stds, w.dY);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(X, w.dX, (size_t)m * k * 8, hipMemcpyDeviceToHost, st));
  HIPCK(hipMemcpyAsync(Y, w.dY, (size_t)d * k * 8, hipMemcpyDeviceToHost, st));
  HIPCK(hipStreamSynchronize(st));
  if (singular_values) for (int c = 0; c < k; ++c) singular_values[c] = std::sqrt(std::max(ritz[c], 0.0));
  if (iters_done) *iters_done = it;
  return GLRM_OK;
}
