// reml_chol.hip -- the EMMA restricted likelihood and the EMMAX scan model WITHOUT an eigendecomposition.
//
// The reference gets everything from eigh(K) and eigh(S(K+I)S) (linear_models.py:589-615, 771-927).  What it
// actually consumes are four sums per variance ratio delta (H = K + delta I, P = H^-1 - H^-1 X (X'H^-1 X)^-1 X'H^-1):
//     s1 = y'Py,   s2 = log|H| + log|X'H^-1 X| - log|X'X|,   s3 = |Py|^2,   s4 = tr P
// (the 51-point grid :796-810, the secant iterations :847, the final log-likelihood :882, vg :894) and, for the
// scan, A = Mp Mp' = P,  w = Mp r = Py,  h0_rss = y'Py  (:1290-1303 in closed form).  All of these follow from ONE
// Cholesky factorisation per delta:  H = LL',  Z = L^-1 [X y],  G = L^-T Z = H^-1 [X y],  tr H^-1 = |L^-1|_F^2.
// Cost per delta: N^3/3 (potrf) + N^3/3 (triangular inverse, recursive over trsm so that the zero half is never
// touched) against ~10 N^3 for the eigendecomposition -- and, unlike rocsolver_dsyevd, every routine used here has
// a 64-bit-index form (rocsolver_dpotrf_64, rocblas_dtrsm_64, rocblas_dsyrk_64, rocblas_dgemm_64), so it runs at
// N = 50,000 where the eigensolver has to fall back to block Jacobi (397 s, eigh_block.hip).  The grid points are
// independent: a multi-GPU run can deal them out (mixmogam_amd/linear_models.py:_SpectralSumsChol, `coll`).
// Roofline: fp64 MFMA through rocSOLVER / rocBLAS (library GEMMs, off the SNPs/s metric).
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dense64.h"
#include "mmg_internal.h"
#include "reml_common.h"

namespace mmg {

__global__ void add_diag_kernel(double* __restrict__ A, int64_t N, double delta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) A[i * N + i] += delta;
}

// sum of log(diag) and, of the lower triangle incl. diagonal (column-major view: rows >= cols), the squared Frobenius norm
// one block, fixed summation order (the likelihood must not depend on atomics' arrival order)
__global__ __launch_bounds__(256) void logdiag_kernel(const double* __restrict__ A, int64_t N, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < N; i += 256) s += log(A[i * N + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (w[0] + w[1]) + (w[2] + w[3]);
}

__global__ __launch_bounds__(256) void lower_sqnorm_kernel(const double* __restrict__ A, int64_t N, double* __restrict__ part) {
  // one block per column (column-major: column j holds rows j..N-1 of the lower triangle contiguously)
  const int64_t j = blockIdx.x;
  double s = 0.0;
  for (int64_t i = j + threadIdx.x; i < N; i += 256) { const double v = A[j * N + i]; s = fma(v, v, s); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[j] = w[0] + w[1] + w[2] + w[3];
}

// zero the strictly upper triangle (column-major: rows < cols) so that the triangular factor can be used as a dense matrix
__global__ void zero_upper_kernel(double* __restrict__ A, int64_t N) {
  const int64_t j = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < j) A[j * N + i] = 0.0;
}

// mirror the lower triangle into the upper one (column-major A[j*N + i], i > j  ->  A[i*N + j])
__global__ void mirror_lower_kernel(double* __restrict__ A, int64_t N) {
  const int64_t j = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N && i > j) A[i * N + j] = A[j * N + i];
}

// first failing block of the blocked Cholesky: acc = (row offset of the block) + (info of its potrf), once
__global__ void note_info_kernel(const rocblas_int* __restrict__ info, long long* __restrict__ acc, long long base) {
  if (*info != 0 && *acc == 0) *acc = base + (long long)*info;
}

}  // namespace mmg
using namespace mmg;

// in-place inverse of the lower-triangular n x n block at L (leading dimension ld, column-major), recursive:
// [[L11, 0], [L21, L22]]^-1 = [[X11, 0], [-X22 L21 X11, X22]] -- two trsm per level touch only the non-zero half.
static int tri_inv_lower(mmg_ctx* ctx, rocblas_handle h, double* L, int64_t n, int64_t ld, rocblas_int* dinfo) {
  const int64_t NB = 4096;
  if (n <= NB) {
    RC_RB(ctx, rocsolver_dtrtri(h, rocblas_fill_lower, rocblas_diagonal_non_unit, (rocblas_int)n, L, (rocblas_int)ld, dinfo));
    return MMG_OK;
  }
  const int64_t n1 = (n / 2 + 63) / 64 * 64, n2 = n - n1;
  double* L11 = L;
  double* L21 = L + n1;
  double* L22 = L + n1 + n1 * ld;
  const double one = 1.0, mone = -1.0;
  static const bool use_trmm = [] { const char* e = std::getenv("MMG_REML_TRTRI"); return !(e && std::string(e) == "trsm"); }();
  if (use_trmm) {
    // invert the diagonal blocks first, then L21 <- -X22 L21 X11 by two triangular multiplies (in place)
    int rc = tri_inv_lower(ctx, h, L11, n1, ld, dinfo);
    if (rc) return rc;
    rc = tri_inv_lower(ctx, h, L22, n2, ld, dinfo);
    if (rc) return rc;
    RC_RB(ctx, rocblas_dtrmm_64(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                n2, n1, &one, L11, ld, L21, ld, L21, ld));
    RC_RB(ctx, rocblas_dtrmm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                n2, n1, &mone, L22, ld, L21, ld, L21, ld));
    return MMG_OK;
  }
  // L21 <- L21 L11^-1   (X L11 = L21)
  RC_RB(ctx, rocblas_dtrsm_64(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                              n2, n1, &one, L11, ld, L21, ld));
  // L21 <- -L22^-1 L21
  RC_RB(ctx, rocblas_dtrsm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                              n2, n1, &mone, L22, ld, L21, ld));
  int rc = tri_inv_lower(ctx, h, L11, n1, ld, dinfo);
  if (rc) return rc;
  return tri_inv_lower(ctx, h, L22, n2, ld, dinfo);
}

// lower triangle of P = X'X for a LOWER-TRIANGULAR X (upper triangle stored as zeros), recursively so that the zero
// half is never multiplied: [[X11, 0], [X21, X22]]' [[X11, 0], [X21, X22]] = [[X11'X11 + X21'X21, .], [X22'X21, X22'X22]]
// -- N^3/3 flops where a dense syrk on the factor spends N^3 (1.9 of the 3.6 s of the scan model at N = 50,000).
static int lauum_lower(mmg_ctx* ctx, rocblas_handle h, const double* X, int64_t n, int64_t ld, double* P, int64_t ldp) {
  const double one = 1.0, zero = 0.0;
  if (n <= 2048) {
    RC_RB(ctx, rocblas_dsyrk_64(h, rocblas_fill_lower, rocblas_operation_transpose, n, n, &one, X, ld, &zero, P, ldp));
    return MMG_OK;
  }
  const int64_t n1 = (n / 2 + 63) / 64 * 64, n2 = n - n1;
  const double* X21 = X + n1;
  const double* X22 = X + n1 + n1 * ld;
  int rc = lauum_lower(ctx, h, X, n1, ld, P, ldp);
  if (rc) return rc;
  RC_RB(ctx, rocblas_dsyrk_64(h, rocblas_fill_lower, rocblas_operation_transpose, n1, n2, &one, X21, ld, &one, P, ldp));
  RC_RB(ctx, rocblas_dtrmm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, n2,
                              n1, &one, X22, ld, X21, ld, P + n1, ldp));
  return lauum_lower(ctx, h, X22, n2, ld, P + n1 + n1 * ldp, ldp);
}

static int potrf_own(mmg_ctx* ctx, double* A, int64_t N, int64_t lda, double* LinvT, long long* dacc, long long base, bool keep_all);

// Right-looking blocked Cholesky (lower, column-major, in place) over the 64-bit rocBLAS level-3 routines:
// diagonal block by rocsolver_dpotrf, panel by trsm, trailing update by syrk -- the trailing update carries
// N^3/3 of the flops at GEMM speed.  Selected with MMG_REML_POTRF=blocked (A/B against rocsolver_dpotrf_64).
static int potrf_blocked(mmg_ctx* ctx, rocblas_handle h, double* A, int64_t N, int64_t nb, rocblas_int* dinfo, long long* dacc,
                         double* LinvT = nullptr) {
  const double one = 1.0, mone = -1.0;
  for (int64_t k0 = 0; k0 < N; k0 += nb) {
    const int64_t kb = std::min(nb, N - k0), rest = N - k0 - kb;
    double* Akk = A + k0 + k0 * N;
    if (LinvT) {                                             // diagonal block on this library's kernels (round 4)
      int rc = potrf_own(ctx, Akk, kb, N, LinvT, dacc, (long long)k0, false);
      if (rc) return rc;
    } else {
      RC_RB(ctx, rocsolver_dpotrf(h, rocblas_fill_lower, (rocblas_int)kb, Akk, (rocblas_int)N, dinfo));
      hipLaunchKernelGGL(note_info_kernel, dim3(1), dim3(1), 0, ctx->stream, dinfo, dacc, (long long)k0);
    }
    if (rest > 0) {
      double* Apk = A + (k0 + kb) + k0 * N;                 // panel below the diagonal block
      RC_RB(ctx, rocblas_dtrsm_64(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose,
                                  rocblas_diagonal_non_unit, rest, kb, &one, Akk, N, Apk, N));
      double* Att = A + (k0 + kb) + (k0 + kb) * N;
      RC_RB(ctx, rocblas_dsyrk_64(h, rocblas_fill_lower, rocblas_operation_none, rest, kb, &mone, Apk, N, &one, Att, N));
    }
  }
  return MMG_OK;
}

// The same factorisation on this library's own kernels (dense64.h), 64 columns at a time: the diagonal block in one
// workgroup (with its inverse, so that the panel below is a product on the matrix pipe instead of a triangular solve), the
// trailing matrix by the rank-64 update of its lower 64 x 64 tiles.  Three launches per 64 columns and no library call:
// N = 5000 takes 79 x 3 launches where the blocked form above spent 17 ms in rocSOLVER's unblocked potf2 kernels and
// forward substitutions (profiles/r4_*).
// At large N the rank-64 trailing updates stream the whole trailing matrix 64 columns at a time (N = 50,000: 1.76 s
// against 0.80 s for 2048-column blocks over syrk_64), so beyond N = 8192 this factors the 2048-column diagonal blocks of
// potrf_blocked and rocBLAS keeps the big updates.
// keep_all: LinvT holds one 64 x 64 slot PER block column (the inverses of all diagonal blocks: tri_inv_own below)
static int potrf_own(mmg_ctx* ctx, double* A, int64_t N, int64_t lda, double* LinvT_base /*device, 64 x 64 (x blocks)*/,
                     long long* dacc, long long base, bool keep_all) {
  if (dense64_init()) return set_err(ctx, MMG_E_HIP, "hipFuncSetAttribute (dense64 kernels)");
  hipStream_t st = ctx->stream;
  for (int64_t k0 = 0; k0 < N; k0 += 64) {
    const int kb = (int)std::min<int64_t>(64, N - k0);
    const int64_t rest = N - k0 - kb;
    double* LinvT = LinvT_base + (keep_all ? (k0 / 64) * 4096 : 0);
    launch_potrf_head(st, A + k0 + k0 * lda, lda, kb, LinvT, dacc, base + (long long)k0);
    if (rest > 0) {
      double* panel = A + (k0 + kb) + k0 * lda;
      launch_rows_gemm(st, panel, lda, panel, lda, rest, LinvT);              // X L' = A  <=>  X = A L^-T
      launch_nt_update_lower(st, A + (k0 + kb) + (k0 + kb) * lda, lda, rest, panel, panel, nullptr, nullptr, lda, lda);
    }
  }
  RC_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

// In-place inverse of the Cholesky factor on this library's kernels (N <= 8192, where potrf_own kept the inverse of every
// diagonal block): block column k from the last to the first,
//     X[k][k] = L[k][k]^-1,      X[k+1:, k] = -X[k+1:, k+1:] (L[k+1:, k] X[k][k]),
// i.e. a tall x 64x64 product on the matrix pipe and one lower-triangular x tall product (sym_skinny_kernel<TRI>, contraction
// slices summed by wsum_into_kernel with the sign) per block column -- three launches, no library call.  N = 5000: 4 ms,
// against 3 ms for rocsolver_dtrtri + rocblas_dtrmm (see reml_point): kept behind MMG_REML_TRTRI=own, tested, not the default.
__global__ void put_diag_inverse_kernel(double* __restrict__ A, int64_t lda, int kb, const double* __restrict__ LinvT) {
  const int e = blockIdx.x * 256 + threadIdx.x;               // element (i, j) of the block: A[i][j] = (L^-1)[i][j] = LinvT[j][i]
  const int i = e & 63, j = e >> 6;
  if (i < kb && j < kb) A[i + (int64_t)j * lda] = i >= j ? LinvT[j + 64 * i] : 0.0;
}

static int tri_inv_own(mmg_ctx* ctx, double* L, int64_t N, const double* LinvT_all, Scratch& sc) {
  hipStream_t st = ctx->stream;
  const int64_t nb = (N + 63) / 64;
  double *T = nullptr, *Wp = nullptr;
  const int smax = 8;
  RC_HIP(ctx, sc.alloc(&T, (size_t)N * 64 * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&Wp, (size_t)smax * N * 64 * sizeof(double)));
  for (int64_t k = nb - 1; k >= 0; --k) {
    const int64_t k0 = k * 64, a0 = k0 + 64;
    const int kb = (int)std::min<int64_t>(64, N - k0);
    const int64_t n = N - a0;                                 // rows below the block
    double* panel = L + a0 + k0 * N;
    if (n > 0) {
      // T = L21 X11 (X11 = LinvT': the coefficient is read transposed), then the panel = -X22 T
      launch_rows_gemm(st, panel, N, T, n, n, LinvT_all + k * 4096, true);
      const int S = (int)std::max<int64_t>(1, std::min<int64_t>(smax, (n + 63) / 64 / 6));
      launch_tall_product(st, L + a0 + a0 * N, N, n, T, Wp, S, true);
      launch_slice_sum_into(st, Wp, S, n, panel, N, -1.0);
    }
    hipLaunchKernelGGL(put_diag_inverse_kernel, dim3(16), dim3(256), 0, st, L + k0 + k0 * N, N, kb, LinvT_all + k * 4096);
  }
  RC_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

struct RemlPoint {           // everything one delta yields
  double s1, s2, s3, s4;
  double ldh = 0.0, trh = 0.0;   // log|H|, tr H^-1
  std::vector<double> beta;  // GLS estimate (q)
  std::vector<double> Py;    // N (only when asked for)
  std::vector<double> GA;    // N x q row-major: H^-1 X (X'H^-1 X)^-1 (only when asked for)
};

// factor H = K + delta I in r->dL and fill `pt`; leaves L^-1 (lower) in r->dL when inverse is true
static int reml_point(mmg_ctx* ctx, mmg_reml* r, double delta, bool inverse, bool want_vectors, RemlPoint& pt) {
  rocblas_handle h;
  int rc = reml_handle(ctx, &h);
  if (rc) return rc;
  const int64_t N = r->N;
  const int q = r->q, q1 = q + 1;
  hipStream_t st = ctx->stream;
  rocblas_int* dinfo = (rocblas_int*)(r->dsc + N + 4);   // [N+4]: info word read back; [N+6]: first failing block; [N+8]: scratch
  const bool verbose = std::getenv("MMG_REML_VERBOSE") != nullptr;
  auto now = [&]() { (void)hipStreamSynchronize(st); return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = verbose ? now() : 0.0, t1 = 0.0, t2 = 0.0;
  r->linv_delta = NAN;                                        // dL is about to be overwritten
  RC_HIP(ctx, hipMemcpyAsync(r->dL, r->dK, (size_t)N * N * sizeof(double), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, r->dL, N, delta);
  Scratch sc_inv;                                             // the diagonal blocks' inverses live until the inverse is done
  double* LinvT_all = nullptr;
  bool own_inverse = false;
  {
    // measured at N = 50,000: rocsolver_dpotrf_64 1.31 s (32 TF); the blocked form over syrk_64 0.80 s (52 TF) at
    // nb = 2048 (0.83 / 0.89 s at 4096 / 8192).  Default since round 4: potrf_own.  MMG_REML_POTRF=rocsolver | blocked:<nb> | own.
    const char* pe = std::getenv("MMG_REML_POTRF");
    const std::string ps = pe ? pe : "";
    long long* dacc = (long long*)(r->dsc + N + 6);
    RC_HIP(ctx, hipMemsetAsync(dinfo, 0, 4 * sizeof(int64_t), st));
    if (ps.empty() || ps == "own") {
      // MMG_REML_TRTRI=own: tri_inv_own.  Measured and NOT the default: emmax()'s scan phase 48.9 ms against 45.9 ms with
      // rocsolver_dtrtri + trmm at N = 5000 (9.0 / 7.9 at 2000) -- the 9 ms a kernel trace attributes to the library's
      // inverse are profiler overhead on its many small launches; its wall time is 3 ms
      own_inverse = inverse && N <= 8192 && std::getenv("MMG_REML_TRTRI") && std::string(std::getenv("MMG_REML_TRTRI")) == "own";
      RC_HIP(ctx, sc_inv.alloc(&LinvT_all, (own_inverse ? (size_t)((N + 63) / 64) : 1) * 4096 * sizeof(double)));
      int rcb = N <= 8192 ? potrf_own(ctx, r->dL, N, N, LinvT_all, dacc, 0, own_inverse)
                          : potrf_blocked(ctx, h, r->dL, N, 2048, (rocblas_int*)(r->dsc + N + 8), dacc, LinvT_all);
      if (rcb) return rcb;
      RC_HIP(ctx, hipMemcpyAsync(dinfo, dacc, sizeof(long long), hipMemcpyDeviceToDevice, st));
    } else if (ps != "rocsolver" && (N >= 4096 || ps.rfind("blocked", 0) == 0)) {
      int64_t nb = 2048;
      if (ps.size() > 8) nb = std::max<int64_t>(256, std::atoll(ps.c_str() + 8));    // "blocked:<nb>"
      int rcb = potrf_blocked(ctx, h, r->dL, N, nb, (rocblas_int*)(r->dsc + N + 8), dacc);
      if (rcb) return rcb;
      RC_HIP(ctx, hipMemcpyAsync(dinfo, dacc, sizeof(long long), hipMemcpyDeviceToDevice, st));
    } else {
      RC_RB(ctx, rocsolver_dpotrf_64(h, rocblas_fill_lower, N, r->dL, N, (int64_t*)dinfo));
    }
  }
  if (verbose) t1 = now();
  RC_HIP(ctx, hipMemsetAsync(r->dsc, 0, 4 * sizeof(double), st));
  hipLaunchKernelGGL(logdiag_kernel, dim3(1), dim3(256), 0, st, r->dL, N, r->dsc);
  // Z = L^-1 [X y];  G = L^-T Z
  const double one = 1.0;
  RC_HIP(ctx, hipMemcpyAsync(r->dZ, r->dB, (size_t)N * q1 * sizeof(double), hipMemcpyDeviceToDevice, st));
  RC_RB(ctx, rocblas_dtrsm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, N,
                              q1, &one, r->dL, N, r->dZ, N));
  RC_HIP(ctx, hipMemcpyAsync(r->dG, r->dZ, (size_t)N * q1 * sizeof(double), hipMemcpyDeviceToDevice, st));
  RC_RB(ctx, rocblas_dtrsm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                              N, q1, &one, r->dL, N, r->dG, N));
  std::vector<double> Z((size_t)N * q1), G((size_t)N * q1);
  double sc[4] = {0, 0, 0, 0};
  int64_t info64 = 0;
  RC_HIP(ctx, hipMemcpyAsync(Z.data(), r->dZ, Z.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  RC_HIP(ctx, hipMemcpyAsync(G.data(), r->dG, G.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  RC_HIP(ctx, hipMemcpyAsync(sc, r->dsc, sizeof(sc), hipMemcpyDeviceToHost, st));
  RC_HIP(ctx, hipMemcpyAsync(&info64, dinfo, sizeof(info64), hipMemcpyDeviceToHost, st));
  RC_HIP(ctx, hipStreamSynchronize(st));
  if (info64 != 0) return set_err(ctx, MMG_E_LIB, "K + delta I is not positive definite (dpotrf info " + std::to_string((long long)info64) + ")");
  const double logdetH = 2.0 * sc[0];
  double trHinv = 0.0;
  if (verbose) t2 = now();
  if (inverse) {
    rc = own_inverse ? tri_inv_own(ctx, r->dL, N, LinvT_all, sc_inv) : tri_inv_lower(ctx, h, r->dL, N, N, dinfo);
    if (rc) return rc;
    hipLaunchKernelGGL(lower_sqnorm_kernel, dim3((unsigned)N), dim3(256), 0, st, r->dL, N, r->dsc + 4);
    std::vector<double> part((size_t)N);
    RC_HIP(ctx, hipMemcpyAsync(part.data(), r->dsc + 4, N * sizeof(double), hipMemcpyDeviceToHost, st));
    RC_HIP(ctx, hipStreamSynchronize(st));
    for (int64_t j = 0; j < N; ++j) trHinv += part[j];
  }
  if (verbose)
    fprintf(stderr, "[reml] N=%lld delta=%.4g: copy+potrf %.3f s, solves+download %.3f s, triangular inverse %.3f s\n",
            (long long)N, delta, t1 - t0, t2 - t1, now() - t2);
  // ---- q x q algebra on the host (columns of Z / G are contiguous: column-major N x q1)
  auto colZ = [&](int c) { return Z.data() + (size_t)c * N; };
  auto colG = [&](int c) { return G.data() + (size_t)c * N; };
  auto dot = [&](const double* a, const double* b) { double s = 0; for (int64_t i = 0; i < N; ++i) s += a[i] * b[i]; return s; };
  std::vector<double> a((size_t)q * q), b((size_t)q), B2((size_t)q * q);
  for (int i = 0; i < q; ++i) {
    for (int j = 0; j <= i; ++j) {
      a[i * q + j] = a[j * q + i] = dot(colZ(i), colZ(j));                  // X'H^-1 X
      B2[i * q + j] = B2[j * q + i] = dot(colG(i), colG(j));               // X'H^-2 X
    }
    b[i] = dot(colZ(i), colZ(q));                                          // X'H^-1 y
  }
  const double c = dot(colZ(q), colZ(q));                                  // y'H^-1 y
  std::vector<double> beta = b;
  double logdet_a = 0.0;
  if (!chol_solve_small(q, a, beta, 1, &logdet_a)) return set_err(ctx, MMG_E_LIB, "X'H^-1 X is not positive definite");
  std::vector<double> aB2 = B2;
  chol_solve_small(q, a, aB2, q, nullptr);                                 // a^-1 B2
  double tr_aB2 = 0.0, bb = 0.0;
  for (int i = 0; i < q; ++i) { tr_aB2 += aB2[i * q + i]; bb += b[i] * beta[i]; }
  std::vector<double> Py((size_t)N);
  double s3 = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    double v = colG(q)[i];
    for (int k = 0; k < q; ++k) v -= colG(k)[i] * beta[k];
    Py[i] = v;
    s3 += v * v;
  }
  pt.s1 = c - bb;
  pt.s2 = logdetH + logdet_a - r->logdet_xtx;
  pt.s3 = s3;
  pt.s4 = trHinv - tr_aB2;
  pt.ldh = logdetH;
  pt.trh = trHinv;
  pt.beta = beta;
  if (want_vectors) {
    pt.Py = Py;
    // GA = Gx a^-1  (N x q row-major)
    std::vector<double> ainv((size_t)q * q, 0.0);
    for (int i = 0; i < q; ++i) ainv[i * q + i] = 1.0;
    chol_solve_small(q, a, ainv, q, nullptr);
    pt.GA.assign((size_t)N * q, 0.0);
    for (int64_t i = 0; i < N; ++i)
      for (int j = 0; j < q; ++j) {
        double v = 0.0;
        for (int k = 0; k < q; ++k) v += colG(k)[i] * ainv[k * q + j];
        pt.GA[(size_t)i * q + j] = v;
      }
  }
  return MMG_OK;
}

namespace mmg {
int model_from_device_public(mmg_ctx* ctx, int32_t N, const double* dA, const double* dw, int ndigits, bool adaptive);

// L^-1 of K + delta I = L L' in r->dL as a dense lower-triangular matrix (column-major, upper triangle zero): what the scan
// model has just left there for the same delta, or one factorisation + triangular inverse (N = 5000: ~10 ms).
// Any H with H'H = (K + delta I)^-1 serves as the reference's H_sqrt_inv (linear_models.py:898 takes diag((lambda +
// delta)^-1/2) U', fixed only up to LAPACK's eigenvector signs); L^-1 is one that needs no eigendecomposition.
int reml_linv_device(mmg_ctx* ctx, mmg_reml* r, double delta, const double** dLinv) {
  if (!(r->linv_delta == delta)) {
    RemlPoint pt;
    int rc = reml_point(ctx, r, delta, true, false, pt);
    if (rc) return rc;
    const int64_t N = r->N;
    hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, ctx->stream, r->dL, N);
    RC_HIP(ctx, hipGetLastError());
    r->linv_delta = delta;
  }
  *dLinv = r->dL;
  return MMG_OK;
}
int reml_linv_device_opaque(mmg_ctx* ctx, mmg_reml* r, double delta, const double** dLinv, int32_t* N) {   // for api.hip
  *N = r->N;
  return reml_linv_device(ctx, r, delta, dLinv);
}
}

extern "C" {

// K: host matrix, or (K_on_device) a device pointer -- the kinship a streamed pass left in HBM (mmg_reml_create_from_acc)
static int reml_create(mmg_ctx* ctx, int32_t N, int32_t q, const double* K, bool K_on_device, const double* X, const double* y, mmg_reml** out) {
  if (!ctx) return MMG_E_ARG;
  MMG_NOTE_ENTRY();
  RC_HIP(ctx, hipSetDevice(ctx->device));
  if (!(out && K && X && y && N > 0 && q >= 1 && q <= 16 && q < N)) return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_create");
  *out = nullptr;
  mmg_reml* r = new mmg_reml();
  r->N = N; r->q = q;
  const size_t nn = (size_t)N * N * sizeof(double), nq = (size_t)N * (q + 1) * sizeof(double);
  hipError_t e = hipMalloc(&r->dK, nn);
  if (e == hipSuccess) e = hipMalloc(&r->dL, nn);
  if (e == hipSuccess) e = hipMalloc(&r->dB, nq);
  if (e == hipSuccess) e = hipMalloc(&r->dZ, nq);
  if (e == hipSuccess) e = hipMalloc(&r->dG, nq);
  if (e == hipSuccess) e = hipMalloc(&r->dsc, ((size_t)N + 16) * sizeof(double));
  if (e != hipSuccess) {
    hipFree(r->dK); hipFree(r->dL); hipFree(r->dB); hipFree(r->dZ); hipFree(r->dG); hipFree(r->dsc); delete r;
    return set_err(ctx, MMG_E_NOMEM, std::string("hipMalloc REML workspace: ") + hipGetErrorString(e));
  }
  r->X.assign(X, X + (size_t)N * q);
  r->y.assign(y, y + N);
  std::vector<double> B((size_t)N * (q + 1));                              // column-major [X y]
  for (int64_t i = 0; i < N; ++i) {
    for (int c = 0; c < q; ++c) B[(size_t)c * N + i] = X[(size_t)i * q + c];
    B[(size_t)q * N + i] = y[i];
  }
  RC_HIP(ctx, hipMemcpyAsync(r->dK, K, nn, K_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  RC_HIP(ctx, hipMemcpyAsync(r->dB, B.data(), nq, hipMemcpyHostToDevice, ctx->stream));
  RC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // log|X'X| and |Sy|^2 = y'y - y'X (X'X)^-1 X'y
  std::vector<double> xtx((size_t)q * q, 0.0), xty((size_t)q, 0.0);
  double yy = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    for (int a = 0; a < q; ++a) {
      for (int b = 0; b < q; ++b) xtx[a * q + b] += X[(size_t)i * q + a] * X[(size_t)i * q + b];
      xty[a] += X[(size_t)i * q + a] * y[i];
    }
    yy += y[i] * y[i];
  }
  std::vector<double> sol = xty;
  if (!chol_solve_small(q, xtx, sol, 1, &r->logdet_xtx)) { mmg_reml_destroy(ctx, r); return set_err(ctx, MMG_E_ARG, "X'X is singular"); }
  double t = 0.0;
  for (int a = 0; a < q; ++a) t += xty[a] * sol[a];
  r->sum_sq_etas = yy - t;
  *out = r;
  return MMG_OK;
}

int mmg_reml_create(mmg_ctx* ctx, int32_t N, int32_t q, const double* K, const double* X, const double* y, mmg_reml** out) {
  return reml_create(ctx, N, q, K, false, X, y, out);
}

int mmg_reml_create_dev(mmg_ctx* ctx, int32_t N, int32_t q, const double* dK, const double* X, const double* y, mmg_reml** out) {
  return reml_create(ctx, N, q, dK, true, X, y, out);
}

int mmg_reml_destroy(mmg_ctx* ctx, mmg_reml* r) {
  if (!r) return MMG_OK;
  MMG_NOTE_ENTRY();
  if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
  hipFree(r->dK); hipFree(r->dL); hipFree(r->dB); hipFree(r->dZ); hipFree(r->dG); hipFree(r->dsc);
  reml_band_free(r);
  delete r;
  return MMG_OK;
}

int mmg_reml_sums_ex(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s2, double* s3,
                     double* s4, double* sum_sq_etas, int32_t route) {
  if (!ctx) return MMG_E_ARG;
  MMG_NOTE_ENTRY();
  RC_HIP(ctx, hipSetDevice(ctx->device));
  if (!(r && deltas && s1 && s2 && s3 && s4 && nd >= 0 && route >= 0 && route <= 2))
    return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_sums");
  if (sum_sq_etas) *sum_sq_etas = r->sum_sq_etas;
  if (route == MMG_REML_ROUTE_AUTO) {
    // one band reduction costs about two factorisations + inverses and serves every later delta of this workspace
    const char* e = std::getenv("MMG_REML_ROUTE");
    const std::string es = e ? e : "";
    route = es == "chol" ? MMG_REML_ROUTE_CHOL : (es == "band" || r->band_ready || r->N >= 256) ? MMG_REML_ROUTE_BAND : MMG_REML_ROUTE_CHOL;
  }
  if (route == MMG_REML_ROUTE_BAND) return reml_band_sums(ctx, r, nd, deltas, s1, s2, s3, s4);
  for (int k = 0; k < nd; ++k) {
    RemlPoint pt;
    int rc = reml_point(ctx, r, deltas[k], true, false, pt);
    if (rc) return rc;
    s1[k] = pt.s1; s2[k] = pt.s2; s3[k] = pt.s3; s4[k] = pt.s4;
  }
  return MMG_OK;
}

int mmg_reml_band_info(mmg_ctx* ctx, mmg_reml* r, int32_t* band_ready, int32_t* householder_fallback, double* seconds) {
  if (!ctx) return MMG_E_ARG;
  if (!r) return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_band_info");
  if (band_ready) *band_ready = r->band_ready ? 1 : 0;
  if (householder_fallback) *householder_fallback = r->band_fallback ? 1 : 0;
  if (seconds) *seconds = r->band_s;
  return MMG_OK;
}

int mmg_reml_sums_ml(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s3, double* logdet_h,
                     double* tr_hinv, int32_t route) {
  if (!ctx) return MMG_E_ARG;
  MMG_NOTE_ENTRY();
  RC_HIP(ctx, hipSetDevice(ctx->device));
  if (!(r && deltas && s1 && s3 && logdet_h && tr_hinv && nd >= 0 && route >= 0 && route <= 2))
    return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_sums_ml");
  if (route == MMG_REML_ROUTE_AUTO) route = (r->band_ready || r->N >= 256) ? MMG_REML_ROUTE_BAND : MMG_REML_ROUTE_CHOL;
  if (route == MMG_REML_ROUTE_BAND) {
    std::vector<double> s2((size_t)nd), s4((size_t)nd);
    return reml_band_sums(ctx, r, nd, deltas, s1, s2.data(), s3, s4.data(), logdet_h, tr_hinv);
  }
  for (int k = 0; k < nd; ++k) {
    RemlPoint pt;
    int rc = reml_point(ctx, r, deltas[k], true, false, pt);
    if (rc) return rc;
    s1[k] = pt.s1; s3[k] = pt.s3; logdet_h[k] = pt.ldh; tr_hinv[k] = pt.trh;
  }
  return MMG_OK;
}

int mmg_reml_sums(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s2, double* s3,
                  double* s4, double* sum_sq_etas) {
  return mmg_reml_sums_ex(ctx, r, nd, deltas, s1, s2, s3, s4, sum_sq_etas, MMG_REML_ROUTE_AUTO);
}

int mmg_reml_scan_model_c(mmg_ctx* ctx, mmg_reml* r, double delta, int ndigits, double* h0_rss, double* beta,
                          double* mahalanobis_rss, double* C_out);

int mmg_reml_scan_model(mmg_ctx* ctx, mmg_reml* r, double delta, int ndigits, double* h0_rss, double* beta,
                        double* mahalanobis_rss) {
  return mmg_reml_scan_model_c(ctx, r, delta, ndigits, h0_rss, beta, mahalanobis_rss, nullptr);
}

// C_out (q x N, row-major; NULL: not wanted): (X'V^-1 X)^-1 X'V^-1, V = K + delta I -- what _emmax_f_test_(with_betas=True)
// multiplies every SNP with for the covariates' coefficients (linear_models.py:1300-1303,1323: R^-1 Q'H of the QR of H X)
int mmg_reml_scan_model_c(mmg_ctx* ctx, mmg_reml* r, double delta, int ndigits, double* h0_rss, double* beta,
                          double* mahalanobis_rss, double* C_out) {
  if (!ctx) return MMG_E_ARG;
  MMG_NOTE_ENTRY();
  RC_HIP(ctx, hipSetDevice(ctx->device));
  if (!r) return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_scan_model");
  rocblas_handle h;
  int rc = reml_handle(ctx, &h);
  if (rc) return rc;
  RemlPoint pt;
  rc = reml_point(ctx, r, delta, true, true, pt);          // leaves L^-1 (lower) in dL
  if (rc) return rc;
  const int64_t N = r->N;
  const int q = r->q;
  hipStream_t st = ctx->stream;
  Scratch sc;
  double *dP = nullptr, *dw = nullptr, *dGA = nullptr, *dGx = nullptr;
  RC_HIP(ctx, sc.alloc(&dP, (size_t)N * N * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&dw, N * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&dGA, (size_t)N * q * sizeof(double)));
  // H^-1 = L^-T L^-1: syrk on the triangular factor used as a dense matrix (upper triangle zeroed first)
  hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, st, r->dL, N);
  r->linv_delta = delta;                                      // (reml_linv_device: the permutation test of the same delta reuses it)
  const double one = 1.0, zero = 0.0, mone = -1.0;
  {
    static const bool dense = [] { const char* e = std::getenv("MMG_REML_LAUUM"); return e && e[0] == '0'; }();   // A/B: dense syrk on the factor
    if (dense) RC_RB(ctx, rocblas_dsyrk_64(h, rocblas_fill_lower, rocblas_operation_transpose, N, N, &one, r->dL, N, &zero, dP, N));
    else if ((rc = lauum_lower(ctx, h, r->dL, N, N, dP, N))) return rc;
  }
  hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, st, dP, N);
  // P = H^-1 - (Gx a^-1) Gx'   (GA row-major N x q == column-major q x N;  dG column-major N x (q+1): Gx = first q columns)
  RC_HIP(ctx, hipMemcpyAsync(dGA, pt.GA.data(), (size_t)N * q * sizeof(double), hipMemcpyHostToDevice, st));
  dGx = r->dG;
  // column-major: P (N x N) -= Gx (N x q) * GA' (q x N) where GA' in column-major is the row-major GA buffer read as q x N
  RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, N, N, q, &mone, dGx, N, dGA, q, &one, dP, N));
  RC_HIP(ctx, hipMemcpyAsync(dw, pt.Py.data(), N * sizeof(double), hipMemcpyHostToDevice, st));
  RC_HIP(ctx, hipGetLastError());
  bool adaptive = false;
  if (ndigits == 0) {
    ndigits = 4;
    const char* e = std::getenv("MMG_SCAN_ADAPTIVE");
    adaptive = !(e && e[0] == '0');
  }
  rc = model_from_device_public(ctx, (int32_t)N, dP, dw, ndigits, adaptive);
  if (rc) return rc;
  RC_HIP(ctx, hipStreamSynchronize(st));
  if (h0_rss) *h0_rss = pt.s1;
  if (mahalanobis_rss) *mahalanobis_rss = pt.s1;
  if (beta) std::memcpy(beta, pt.beta.data(), q * sizeof(double));
  if (C_out)                                                  // GA (N x q, row-major) = V^-1 X (X'V^-1 X)^-1 = C'
    for (int64_t i = 0; i < N; ++i)
      for (int c = 0; c < q; ++c) C_out[(size_t)c * N + i] = pt.GA[(size_t)i * q + c];
  return MMG_OK;
}


// out [N x k] = L^-1 V (trans = 0) or L^-T V (trans = 1) for K + delta I = L L'; V, out: column-major N x k on the host.
// With V = [X y]: H X and H y of linear_models.py:1290-1291 for H = L^-1.
int mmg_reml_linv_apply(mmg_ctx* ctx, mmg_reml* r, double delta, int32_t trans, const double* V, int32_t k, double* out) {
  if (!ctx) return MMG_E_ARG;
  MMG_NOTE_ENTRY();
  RC_HIP(ctx, hipSetDevice(ctx->device));
  if (!(r && V && out && k >= 1 && (trans == 0 || trans == 1))) return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_linv_apply");
  rocblas_handle h;
  int rc = reml_handle(ctx, &h);
  if (rc) return rc;
  const double* dLinv = nullptr;
  if ((rc = reml_linv_device(ctx, r, delta, &dLinv))) return rc;
  const int64_t N = r->N;
  Scratch sc;
  double *dV = nullptr, *dO = nullptr;
  RC_HIP(ctx, sc.alloc(&dV, (size_t)N * k * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&dO, (size_t)N * k * sizeof(double)));
  RC_HIP(ctx, hipMemcpyAsync(dV, V, (size_t)N * k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const double one = 1.0, zero = 0.0;
  RC_RB(ctx, rocblas_dgemm_64(h, trans ? rocblas_operation_transpose : rocblas_operation_none, rocblas_operation_none, N, k, N, &one,
                              dLinv, N, dV, N, &zero, dO, N));
  RC_HIP(ctx, hipMemcpyAsync(out, dO, (size_t)N * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

// H_out [N x N] row-major = L^-1 (lower triangular): a square root of (K + delta I)^-1 for callers that want the matrix itself
int mmg_reml_linv_fetch(mmg_ctx* ctx, mmg_reml* r, double delta, double* H_out) {
  if (!ctx) return MMG_E_ARG;
  MMG_NOTE_ENTRY();
  RC_HIP(ctx, hipSetDevice(ctx->device));
  if (!(r && H_out)) return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_linv_fetch");
  rocblas_handle h;
  int rc = reml_handle(ctx, &h);
  if (rc) return rc;
  const double* dLinv = nullptr;
  if ((rc = reml_linv_device(ctx, r, delta, &dLinv))) return rc;
  const int64_t N = r->N;
  Scratch sc;
  double* dT = nullptr;
  RC_HIP(ctx, sc.alloc(&dT, (size_t)N * N * sizeof(double)));
  const double one = 1.0, zero = 0.0;                          // out-of-place transpose: column-major L^-1 -> row-major L^-1
  RC_RB(ctx, rocblas_dgeam(h, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)N, (rocblas_int)N, &one, dLinv,
                           (rocblas_int)N, &zero, dT, (rocblas_int)N, dT, (rocblas_int)N));
  RC_HIP(ctx, hipMemcpyAsync(H_out, dT, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

}  // extern "C"
