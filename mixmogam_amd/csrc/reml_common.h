// reml_common.h -- what reml_chol.hip (one Cholesky factorisation per delta) and reml_band.hip (one band reduction for
// all deltas) share: the workspace behind mmg_reml, error macros, the q x q host algebra.
#pragma once
#include <rocblas/rocblas.h>

#include <cmath>
#include <string>
#include <vector>

#include "mmg_internal.h"

using mmg::set_err;

struct mmg_reml {
  int32_t N = 0, q = 0;
  double* dK = nullptr;     // [N x N] symmetric
  double* dL = nullptr;     // [N x N] work: H -> L -> L^-1
  double linv_delta = NAN;  // dL holds L^-1 of K + linv_delta I = L L' as a dense lower-triangular matrix (upper triangle zero); NaN: it does not
  double* dB = nullptr;     // [N x (q+1)] = [X y] (column-major: column c contiguous)
  double* dZ = nullptr;     // [N x (q+1)]
  double* dG = nullptr;     // [N x (q+1)]
  double* dsc = nullptr;    // scalars / per-column partials [N + 8]
  std::vector<double> X, y; // host copies (X row-major N x q)
  double logdet_xtx = 0.0, sum_sq_etas = 0.0;
  void* rocblas = nullptr;
  // band route (reml_band.hip): K reduced once to an orthogonally similar band matrix, every delta from that
  bool band_ready = false;
  double* dBand = nullptr;  // [N][BAND_LD]: column j of the band, B[j + d][j] at d = 0..64
  double* dZr = nullptr;    // [q+1][N]: Q'[X y]
  double band_s = 0.0;      // seconds the reduction took
  int64_t band_k0 = 0;      // first column the Householder-panel path still has to do (reml_band.hip:band_reduce)
  bool band_fallback = false;   // a Cholesky-QR panel was rank deficient: the reduction was redone with Householder panels
  std::vector<double> keep_deltas, keep_logdet;   // mmg_reml_band_factor: variance ratios whose banded factors lie in ctx->band_keep (while
                                                  // ctx->band_keep_owner == this), and log|B + delta I| of each
  void* dBandWs = nullptr;      // per-delta factors and right-hand sides of reml_band_sums: grows, freed with the workspace (a
  size_t band_ws_bytes = 0;     // hipMalloc + hipFree of 650 MB per call was 3 ms of an 11 ms call of 227 variance ratios)
};

#define RC_HIP(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) return set_err(ctx, MMG_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)
#define RC_RB(ctx, call)                                                                      \
  do {                                                                                        \
    rocblas_status s__ = (call);                                                              \
    if (s__ != rocblas_status_success)                                                        \
      return set_err(ctx, MMG_E_LIB, std::string(#call) + ": rocblas status " + std::to_string((int)s__)); \
  } while (0)

// small dense helpers on the host (q x q, q <= 16)
static inline bool chol_solve_small(int q, std::vector<double> a, std::vector<double>& b, int nrhs, double* logdet) {
  // a: q x q SPD row-major (destroyed); b: q x nrhs row-major, overwritten with a^-1 b
  double ld = 0.0;
  for (int j = 0; j < q; ++j) {
    double d = a[j * q + j];
    for (int k = 0; k < j; ++k) d -= a[j * q + k] * a[j * q + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    a[j * q + j] = d;
    ld += 2.0 * std::log(d);
    for (int i = j + 1; i < q; ++i) {
      double v = a[i * q + j];
      for (int k = 0; k < j; ++k) v -= a[i * q + k] * a[j * q + k];
      a[i * q + j] = v / d;
    }
  }
  for (int r = 0; r < nrhs; ++r) {
    for (int i = 0; i < q; ++i) {
      double v = b[i * nrhs + r];
      for (int k = 0; k < i; ++k) v -= a[i * q + k] * b[k * nrhs + r];
      b[i * nrhs + r] = v / a[i * q + i];
    }
    for (int i = q - 1; i >= 0; --i) {
      double v = b[i * nrhs + r];
      for (int k = i + 1; k < q; ++k) v -= a[k * q + i] * b[k * nrhs + r];
      b[i * nrhs + r] = v / a[i * q + i];
    }
  }
  if (logdet) *logdet = ld;
  return true;
}

static inline int reml_handle(mmg_ctx* ctx, rocblas_handle* h) {
  if (!ctx->rocblas) {
    rocblas_handle hh;
    RC_RB(ctx, rocblas_create_handle(&hh));
    RC_RB(ctx, rocblas_set_stream(hh, ctx->stream));
    ctx->rocblas = hh;
  }
  *h = (rocblas_handle)ctx->rocblas;
  return MMG_OK;
}


namespace mmg {
// reml_band.hip: the four sums for nd deltas through the band matrix (reduces K on first use)
// ldh / trh (optional): log|K + delta I| and tr (K + delta I)^-1 of every delta (what the ML likelihood adds, :634-649)
int reml_band_sums(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s2, double* s3, double* s4,
                   double* ldh = nullptr, double* trh = nullptr);
void reml_band_free(mmg_reml* r);
int reml_band_factor_keep(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas);   // mmg_reml_band_factor
// reml_chol.hip: makes r->dL hold L^-1 of K + delta I = L L' (column-major, upper triangle zero) and hands the pointer out
int reml_linv_device(mmg_ctx* ctx, mmg_reml* r, double delta, const double** dLinv);
// reml_band.hip: W [n x 64] (ld n) = A V on the fp64 matrix pipe for an n x n matrix A of which only the lower triangle is read
// (column-major, ld lda): symmetric (tri = false) or lower triangular (tri = true); V [n x 64] (ld n).  S > 1: the contraction
// range in S slices, slice s into W_or_Wp + s * n * 64 (the caller adds them up: launch_slice_sum_into).
void launch_tall_product(hipStream_t st, const double* A, int64_t lda, int64_t n, const double* V, double* W_or_Wp, int S, bool tri);
void launch_slice_sum_into(hipStream_t st, const double* Wp, int S, int64_t n, double* dst, int64_t ldd, double sign);
}
