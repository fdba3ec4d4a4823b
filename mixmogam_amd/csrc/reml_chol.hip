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


// ---- own kernels of the library-free scan model (round 5) -------------------------------------------------------------
// out [N x k] = T V (T lower triangular N x N, column-major, only rows >= cols read; V, out column-major N x k, k <= 17):
// thread = row i, loop over the columns j <= i of T (a wave reads 64 consecutive rows of column j: coalesced), V[j][.] broadcast.
// The columns are dealt over blockIdx.y slices, partial sums to part[slice][k][N] (fixed order afterwards: deterministic).
template <int K>
__global__ __launch_bounds__(256) void tri_apply_n_kernel(const double* __restrict__ T, int64_t N, const double* __restrict__ V,
                                                          double* __restrict__ part, int nslice) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i_hi = min(N, ((int64_t)blockIdx.x + 1) * 256);            // columns beyond the block's last row contribute nothing
  const int64_t per = (i_hi + nslice - 1) / nslice;
  const int64_t j0 = (int64_t)blockIdx.y * per, j1 = min(i_hi, j0 + per);
  double acc[K];
#pragma unroll
  for (int c = 0; c < K; ++c) acc[c] = 0.0;
  if (i < N) {
    for (int64_t j = j0; j < j1; ++j) {
      if (j > i) break;
      const double t = T[i + j * N];
#pragma unroll
      for (int c = 0; c < K; ++c) acc[c] = fma(t, V[j + (int64_t)c * N], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < K; ++c) part[((int64_t)blockIdx.y * K + c) * N + i] = acc[c];
  }
}

// out [N x k] = T'V: one wave per column j of T (rows j .. N-1, contiguous), lanes stride the rows, K sums per lane, wave reduce
template <int K>
__global__ __launch_bounds__(256) void tri_apply_t_kernel(const double* __restrict__ T, int64_t N, const double* __restrict__ V,
                                                          double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  if (j >= N) return;
  double acc[K];
#pragma unroll
  for (int c = 0; c < K; ++c) acc[c] = 0.0;
  const double* __restrict__ col = T + j * N;
  for (int64_t i = j + l; i < N; i += 64) {
    const double t = col[i];
#pragma unroll
    for (int c = 0; c < K; ++c) acc[c] = fma(t, V[i + (int64_t)c * N], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < K; ++c) {
    double v = acc[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (l == 0) out[j + (int64_t)c * N] = v;
  }
}

__global__ void slice_sum_kernel(const double* __restrict__ part, int nslice, int64_t n, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double s = 0.0;
  for (int k = 0; k < nslice; ++k) s += part[(int64_t)k * n + e];
  out[e] = s;
}

// P (lower 64 x 64 tiles, column-major ld ldp) = X'X for a lower-triangular X (column-major ld ldx; the strictly upper part of its
// diagonal blocks must be zero, blocks above them are never read): tile (I, J), I >= J, = X[64 I :, I]' X[64 I :, J] -- the rows
// above 64 I of block column I are zero.  One workgroup per tile, the Gram inner loop of dense64.hip (v_mfma_f64_16x16x4_f64,
// operands straight from global memory); replaces rocBLAS dsyrk + dtrmm recursion (lauum_lower).
typedef double v4d_rc __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void lauum_tiles_kernel(const double* __restrict__ X, int64_t ldx, int64_t n, double* __restrict__ P,
                                                          int64_t ldp) {
  const int64_t t = blockIdx.x;
  int64_t I = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int64_t J = t - I * (I + 1) / 2;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  // wave w owns rows 16 w .. 16 w + 15 of the tile (columns 64 I + .. of X), all four 16-column groups of block column J
  const int64_t ca = I * 64 + 16 * w + lr;                   // X column of the a operand
  const bool ain = ca < n;
  const double* __restrict__ ap = X + (ain ? ca : 0) * ldx;
  const double* bp[4];
  bool bin[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int64_t cb = J * 64 + 16 * ct + lr;
    bin[ct] = cb < n;
    bp[ct] = X + (bin[ct] ? cb : 0) * ldx;
  }
  v4d_rc acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = v4d_rc{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int64_t r0 = I * 64; r0 < n; r0 += 4) {
    const int64_t r = r0 + lk;
    const bool in = r < n;
    const double a = (in && ain) ? ap[r] : 0.0;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const double bv = (in && bin[ct]) ? bp[ct][r] : 0.0;
      acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[ct], 0, 0, 0);
    }
  }
  // accumulator layout of v_mfma_f64_16x16x4_f64 (as dense64.hip:gram_slices_kernel stores it): register e of lane l =
  // C[row 4 e + l / 16][column l % 16], rows = the a operand's index (tile row 16 w + ..), columns = the b operand's (16 ct + ..)
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t pi = I * 64 + 16 * w + 4 * e + lk, pj = J * 64 + 16 * ct + lr;
      if (pi < n && pj < n) P[pi + pj * ldp] = acc[ct][e];
    }
}

// The same tiles with the operand panels staged through LDS (round 6).  The contraction of X'X runs over the ROWS of X, its
// contiguous dimension, while a 16x16x4 MFMA operand wants 16 different columns per 4 rows: read straight from global memory
// (kernel above) every wave load touches 16 cache lines for 32 bytes each.  Here a chunk of 32 rows x 64 columns is loaded with
// 64 contiguous bytes per lane (lane = column, wave = 8-row group), written to LDS as [row][column] and read from there in the
// operand layout; the next chunk travels in registers while the current one is multiplied.  2.5 -> ~1.3 ms at N = 5000.
constexpr int LAU_LD = 80;                                    // doubles per staged row: 64 columns + 16 (row stride = 32 banks mod 64: the two
                                                              // k rows a 32-lane half reads fall into disjoint bank halves)
__global__ __launch_bounds__(256) void lauum_tiles_lds_kernel(const double* __restrict__ X, int64_t ldx, int64_t n,
                                                              double* __restrict__ P, int64_t ldp) {
  const int64_t t = blockIdx.x;
  int64_t I = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int64_t J = t - I * (I + 1) / 2;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  __shared__ __attribute__((aligned(16))) double As[2][32 * LAU_LD], Bs[2][32 * LAU_LD];
  const bool same = I == J;
  const int64_t ca = I * 64 + l, cb = J * 64 + l;             // this lane's column of the two panels
  const bool ain = ca < n, bin = cb < n;
  const double* __restrict__ ap = X + (ain ? ca : 0) * ldx;
  const double* __restrict__ bp = X + (bin ? cb : 0) * ldx;
  double ra[8], rb[8];
  auto fetch = [&](int64_t r0) {                              // rows r0 + 8 w .. + 7 of this lane's columns
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t r = r0 + 8 * w + k;
      ra[k] = (ain && r < n) ? ap[r] : 0.0;
      rb[k] = (!same && bin && r < n) ? bp[r] : 0.0;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      As[buf][(8 * w + k) * LAU_LD + l] = ra[k];
      if (!same) Bs[buf][(8 * w + k) * LAU_LD + l] = rb[k];
    }
  };
  v4d_rc acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = v4d_rc{0.0, 0.0, 0.0, 0.0};
  const int64_t rbeg = I * 64;
  fetch(rbeg);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = rbeg; r0 < n; r0 += 32) {
    const bool more = r0 + 32 < n;
    if (more) fetch(r0 + 32);
    const double* __restrict__ a_s = As[buf];
    const double* __restrict__ b_s = same ? As[buf] : Bs[buf];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const double a = a_s[(4 * ks + lk) * LAU_LD + 16 * w + lr];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b_s[(4 * ks + lk) * LAU_LD + 16 * ct + lr], acc[ct], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);                                 // (the other buffer's readers passed the barrier below one chunk ago)
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t pi = I * 64 + 16 * w + 4 * e + lk, pj = J * 64 + 16 * ct + lr;
      if (pi < n && pj < n) P[pi + pj * ldp] = acc[ct][e];
    }
}

// P (n x n column-major, lower tiles valid) -= Gx GA' on the lower tiles and the result mirrored into the upper ones:
// Gx [n x q] column-major (ld ldg), GA [n x q] ROW-major.  One workgroup per 64 x 64 lower tile, the mirrored tile through LDS.
__global__ __launch_bounds__(256) void rankq_mirror_kernel(double* __restrict__ P, int64_t n, const double* __restrict__ Gx, int64_t ldg,
                                                           const double* __restrict__ GA, int q) {
  const int64_t t = blockIdx.x;
  int64_t I = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int64_t J = t - I * (I + 1) / 2;
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t i = I * 64 + tx;                             // row (fast index: column-major)
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int64_t j = J * 64 + ty + 4 * r;
    double v = 0.0;
    if (i < n && j < n && (I > J || i >= j)) {
      v = P[i + j * n];
      for (int c = 0; c < q; ++c) v -= Gx[i + (int64_t)c * ldg] * GA[j * q + c];
      P[i + j * n] = v;
    }
    tile[ty + 4 * r][tx] = v;                                // tile[col][row]
  }
  __syncthreads();
  // the mirrored element P[j][i] for j > i ... i.e. rows of the upper tile (J, I): row index j2 in block J, column i2 in block I
  const int64_t j2 = J * 64 + tx;                            // row of the upper tile (fast index)
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int64_t i2 = I * 64 + ty + 4 * r;                  // its column
    if (j2 < n && i2 < n && (I > J ? true : j2 < i2)) P[j2 + i2 * n] = tile[tx][ty + 4 * r];
  }
}

// B [n x n] column-major = A' (out of place), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose64_kernel(const double* __restrict__ A, int64_t n, double* __restrict__ B) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int64_t i = i0 + tx, j = j0 + ty + 4 * r;
    tile[ty + 4 * r][tx] = (i < n && j < n) ? A[i + j * n] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int64_t j = j0 + tx, i = i0 + ty + 4 * r;          // B[j][i] = A[i][j]
    if (i < n && j < n) B[j + i * n] = tile[tx][ty + 4 * r];
  }
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
// (one launch for ALL diagonal blocks -- blockIdx.y = block column -- in front of the loop below: nothing in it reads a diagonal
// block of L before it has been replaced; round 5 launched this once per block column, 79 x 5 us at N = 5000)
__global__ void put_diag_inverse_kernel(double* __restrict__ L, int64_t lda, int64_t N, const double* __restrict__ LinvT_all) {
  const int64_t k0 = (int64_t)blockIdx.y * 64;
  const int kb = (int)min((int64_t)64, N - k0);
  double* A = L + k0 + k0 * lda;
  const double* LinvT = LinvT_all + (size_t)blockIdx.y * 4096;
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
  hipLaunchKernelGGL(put_diag_inverse_kernel, dim3(16, (unsigned)nb), dim3(256), 0, st, L, N, N, LinvT_all);
  for (int64_t k = nb - 1; k >= 0; --k) {
    const int64_t k0 = k * 64, a0 = k0 + 64;
    const int64_t n = N - a0;                                 // rows below the block
    double* panel = L + a0 + k0 * N;
    if (n > 0) {
      // T = L21 X11 (X11 = LinvT': the coefficient is read transposed), then the panel = -X22 T
      launch_rows_gemm(st, panel, N, T, n, n, LinvT_all + k * 4096, true);
      const int S = (int)std::max<int64_t>(1, std::min<int64_t>(smax, (n + 63) / 64 / 6));
      launch_tall_product(st, L + a0 + a0 * N, N, n, T, Wp, S, true);
      launch_slice_sum_into(st, Wp, S, n, panel, N, -1.0);
    }
  }
  RC_HIP(ctx, hipGetLastError());
  return MMG_OK;
}


// out [N x k] = T V / T'V for the lower-triangular T (see the kernels); V, out device column-major (ld N), k <= 17
static int tri_apply_own(mmg_ctx* ctx, const double* T, int64_t N, const double* V, int k, double* out, bool trans, Scratch& sc) {
  hipStream_t st = ctx->stream;
  if (k < 1 || k > 17) return set_err(ctx, MMG_E_ARG, "tri_apply_own: 1..17 columns");
#define MMG_TA(K_)                                                                                                       \
  case K_:                                                                                                              \
    if (trans) hipLaunchKernelGGL(tri_apply_t_kernel<K_>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, T, N, V, out); \
    else hipLaunchKernelGGL(tri_apply_n_kernel<K_>, dim3((unsigned)((N + 255) / 256), nslice), dim3(256), 0, st, T, N, V, part, nslice); \
    break;
  const int nslice = (int)std::max<int64_t>(1, std::min<int64_t>(16, N / 256));
  double* part = nullptr;
  if (!trans) RC_HIP(ctx, sc.alloc(&part, (size_t)nslice * k * N * sizeof(double)));
  switch (k) {
    MMG_TA(1) MMG_TA(2) MMG_TA(3) MMG_TA(4) MMG_TA(5) MMG_TA(6) MMG_TA(7) MMG_TA(8) MMG_TA(9) MMG_TA(10) MMG_TA(11) MMG_TA(12)
    MMG_TA(13) MMG_TA(14) MMG_TA(15) MMG_TA(16) MMG_TA(17)
  }
#undef MMG_TA
  if (!trans) {
    const int64_t tot = (int64_t)k * N;
    hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, part, nslice, tot, out);
  }
  RC_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

// whether a call of size N runs on this library's kernels alone (no rocBLAS / rocSOLVER handle is even created): the own
// Cholesky keeps every diagonal block's inverse up to N = 8192; MMG_REML_LIB=1 forces the library route (A/B)
static bool reml_own_route(int64_t N) {
  static const bool lib = [] { const char* e = std::getenv("MMG_REML_LIB"); return e && e[0] == '1'; }();
  const char* pe = std::getenv("MMG_REML_POTRF");
  return !lib && N <= 8192 && (!pe || std::string(pe) == "own");
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
  const int64_t N = r->N;
  // Round 5: with the inverse asked for (every caller's case) and N <= 8192 the whole point -- factorisation, inverse, the
  // products with [X y] -- runs on this library's kernels; no rocBLAS / rocSOLVER handle is created (a process that only
  // calls emmax() never loads either library: first call 0.25 -> ~0.1 s at N = 5000, seconds at N = 199 on a cold box)
  const bool own_all = inverse && reml_own_route(N);
  rocblas_handle h = nullptr;
  int rc = MMG_OK;
  if (!own_all && (rc = reml_handle(ctx, &h))) return rc;
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
      own_inverse = own_all || (inverse && N <= 8192 && std::getenv("MMG_REML_TRTRI") && std::string(std::getenv("MMG_REML_TRTRI")) == "own");
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
  bool inverted = false;
  if (own_all) {
    // the inverse first (three launches per block column, dense64.h), then both as products with it
    Scratch sc_ta;
    if ((rc = tri_inv_own(ctx, r->dL, N, LinvT_all, sc_inv))) return rc;
    inverted = true;
    if ((rc = tri_apply_own(ctx, r->dL, N, r->dB, q1, r->dZ, false, sc_ta))) return rc;
    if ((rc = tri_apply_own(ctx, r->dL, N, r->dZ, q1, r->dG, true, sc_ta))) return rc;
    RC_HIP(ctx, hipStreamSynchronize(st));                     // sc_ta is released here
  } else {
    RC_HIP(ctx, hipMemcpyAsync(r->dZ, r->dB, (size_t)N * q1 * sizeof(double), hipMemcpyDeviceToDevice, st));
    RC_RB(ctx, rocblas_dtrsm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, N,
                                q1, &one, r->dL, N, r->dZ, N));
    RC_HIP(ctx, hipMemcpyAsync(r->dG, r->dZ, (size_t)N * q1 * sizeof(double), hipMemcpyDeviceToDevice, st));
    RC_RB(ctx, rocblas_dtrsm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                N, q1, &one, r->dL, N, r->dG, N));
  }
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
    if (!inverted) {
      rc = own_inverse ? tri_inv_own(ctx, r->dL, N, LinvT_all, sc_inv) : tri_inv_lower(ctx, h, r->dL, N, N, dinfo);
      if (rc) return rc;
    }
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
  if (ctx && ctx->band_keep_owner == r) ctx->band_keep_owner = nullptr;     // (the next workspace may get this address)
  hipFree(r->dK); hipFree(r->dL); hipFree(r->dB); hipFree(r->dZ); hipFree(r->dG); hipFree(r->dsc);
  reml_band_free(r);
  delete r;
  return MMG_OK;
}

int mmg_reml_band_factor(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas) {
  if (!ctx) return MMG_E_ARG;
  MMG_NOTE_ENTRY();
  RC_HIP(ctx, hipSetDevice(ctx->device));
  if (!(r && deltas && nd >= 0)) return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_band_factor");
  for (int k = 0; k < nd; ++k)
    if (!(deltas[k] > 0.0)) return set_err(ctx, MMG_E_ARG, "bad argument: mmg_reml_band_factor wants positive variance ratios");
  return reml_band_factor_keep(ctx, r, nd, deltas);
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
  const bool own_all = reml_own_route(r->N);
  rocblas_handle h = nullptr;
  int rc = MMG_OK;
  if (!own_all && (rc = reml_handle(ctx, &h))) return rc;
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
  RC_HIP(ctx, hipMemcpyAsync(dGA, pt.GA.data(), (size_t)N * q * sizeof(double), hipMemcpyHostToDevice, st));
  dGx = r->dG;
  if (own_all) {
    // lower tiles of H^-1 = X'X (lauum_tiles_kernel), then P = H^-1 - Gx GA' on them and the mirror image, one pass
    const int64_t nt = (N + 63) / 64;
    static const bool lauum_direct = [] { const char* e = std::getenv("MMG_LAUUM_TILES"); return e && std::string(e) == "direct"; }();   // round 5's kernel (A/B)
    if (lauum_direct) hipLaunchKernelGGL(lauum_tiles_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, r->dL, N, N, dP, N);
    else hipLaunchKernelGGL(lauum_tiles_lds_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, r->dL, N, N, dP, N);
    hipLaunchKernelGGL(rankq_mirror_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, dP, N, dGx, N, dGA, q);
    RC_HIP(ctx, hipGetLastError());
  } else {
    static const bool dense = [] { const char* e = std::getenv("MMG_REML_LAUUM"); return e && e[0] == '0'; }();   // A/B: dense syrk on the factor
    if (dense) RC_RB(ctx, rocblas_dsyrk_64(h, rocblas_fill_lower, rocblas_operation_transpose, N, N, &one, r->dL, N, &zero, dP, N));
    else if ((rc = lauum_lower(ctx, h, r->dL, N, N, dP, N))) return rc;
    hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, st, dP, N);
    // P = H^-1 - (Gx a^-1) Gx'   (GA row-major N x q == column-major q x N;  dG column-major N x (q+1): Gx = first q columns)
    // column-major: P (N x N) -= Gx (N x q) * GA' (q x N) where GA' in column-major is the row-major GA buffer read as q x N
    RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, N, N, q, &mone, dGx, N, dGA, q, &one, dP, N));
  }
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
  int rc = MMG_OK;
  const double* dLinv = nullptr;
  if ((rc = reml_linv_device(ctx, r, delta, &dLinv))) return rc;
  const int64_t N = r->N;
  Scratch sc;
  double *dV = nullptr, *dO = nullptr;
  RC_HIP(ctx, sc.alloc(&dV, (size_t)N * k * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&dO, (size_t)N * k * sizeof(double)));
  RC_HIP(ctx, hipMemcpyAsync(dV, V, (size_t)N * k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (k <= 17) {                                               // a handful of vectors: the bandwidth-bound own kernels
    if ((rc = tri_apply_own(ctx, dLinv, N, dV, k, dO, trans != 0, sc))) return rc;
  } else {
    rocblas_handle h;
    if ((rc = reml_handle(ctx, &h))) return rc;
    const double one = 1.0, zero = 0.0;
    RC_RB(ctx, rocblas_dgemm_64(h, trans ? rocblas_operation_transpose : rocblas_operation_none, rocblas_operation_none, N, k, N, &one,
                                dLinv, N, dV, N, &zero, dO, N));
  }
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
  int rc = MMG_OK;
  const double* dLinv = nullptr;
  if ((rc = reml_linv_device(ctx, r, delta, &dLinv))) return rc;
  const int64_t N = r->N;
  Scratch sc;
  double* dT = nullptr;
  RC_HIP(ctx, sc.alloc(&dT, (size_t)N * N * sizeof(double)));
  // out-of-place transpose: column-major L^-1 -> row-major L^-1
  hipLaunchKernelGGL(transpose64_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, ctx->stream, dLinv, N, dT);
  RC_HIP(ctx, hipGetLastError());
  RC_HIP(ctx, hipMemcpyAsync(H_out, dT, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

}  // extern "C"
