// eigh_block.hip -- symmetric eigendecomposition beyond rocSOLVER's index range.
//
// rocsolver_dsyevd (ROCm 7.2) addresses the matrix with 32-bit element offsets and faults on the device once
// N*N >= 2^31 (N > 46340; seen at the C5 shape N = 50000).  This file adds a two-sided BLOCK JACOBI solver on
// top of what does work at any size: rocsolver_dsyevd on sub-problems of at most 46336 rows and
// rocblas_dgemm_64 panels.
//
//   partition the rows into p blocks (p = 3 at N = 50000); cyclic sweeps over the block pairs (I, J):
//     G = [[A_II, A_IJ], [A_JI, A_JJ]]  ->  dsyevd  ->  G = W diag(lam) W'
//     A[:, IJ] <- A[:, IJ] W   (dgemm_64, N x |IJ| x |IJ|),  rows IJ = transpose of the new columns (symmetry),
//     A[IJ, IJ] <- diag(lam) exactly;   V[:, IJ] <- V[:, IJ] W
//   until the off-diagonal Frobenius norm is <= 1e-12 ||A||_F (quadratic convergence; 4-7 sweeps), then sort.
//
// Every step is deterministic (fixed pair order, fixed-order norm reductions), so results are reproducible.
// Everything stays in HBM: A, V (N^2 doubles each), G and one N x |IJ| panel -- 85 GB at N = 50000.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>
#include <vector>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include "mmg_internal.h"

namespace mmg {

// rows [o, o+n) of column-major A (ld = N) <- transpose of columns [o, o+n), for k outside the two skip ranges
__global__ __launch_bounds__(256) void sym_rows_from_cols_kernel(double* __restrict__ A, int64_t N, int64_t o, int64_t n,
                                                                 int64_t s0, int64_t n0, int64_t s1, int64_t n1) {
  __shared__ double tile[32][33];
  const int64_t k0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t c = c0 + ty + 8 * r, k = k0 + tx;
    if (c < n && k < N) tile[ty + 8 * r][tx] = A[k + (o + c) * N];   // element (k, o+c): coalesced along k
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t k = k0 + ty + 8 * r, c = c0 + tx;
    if (c < n && k < N && !(k >= s0 && k < s0 + n0) && !(k >= s1 && k < s1 + n1))
      A[(o + c) + k * N] = tile[tx][ty + 8 * r];                     // element (o+c, k): coalesced along c
  }
}

// A[IJ, IJ] <- diag(lam): pair index q in [0, nI + nJ) maps to row oI + q or oJ + (q - nI)
__global__ void set_pair_block_kernel(double* __restrict__ A, int64_t N, int64_t oI, int64_t nI, int64_t oJ, int64_t nJ,
                                      const double* __restrict__ lam) {
  const int64_t n2 = nI + nJ;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n2 * n2) return;
  const int64_t r = gid % n2, c = gid / n2;
  const int64_t gr = r < nI ? oI + r : oJ + (r - nI), gc = c < nI ? oI + c : oJ + (c - nI);
  A[gr + gc * N] = (r == c) ? lam[r] : 0.0;
}

// partial[b] = sum of a^2 over the block's slice (all elements, or off-diagonal only); fixed order
__global__ __launch_bounds__(256) void sq_norm_partial_kernel(const double* __restrict__ A, int64_t N, int offdiag_only,
                                                              double* __restrict__ partial) {
  __shared__ double sh[256];
  const int64_t total = N * N;
  const int64_t per = (total + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < total ? b0 + per : total;
  double s = 0.0;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
    const double v = A[i];
    if (!offdiag_only || (i % N) != (i / N)) s += v * v;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void identity_kernel(double* __restrict__ V, int64_t N) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < N * N) V[gid] = (gid % N) == (gid / N) ? 1.0 : 0.0;
}

__global__ void take_diag_kernel(const double* __restrict__ A, int64_t N, double* __restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) d[i] = A[i + i * N];
}

// out[:, i] = V[:, perm[i]]
__global__ void gather_cols_kernel(const double* __restrict__ V, int64_t N, const int64_t* __restrict__ perm,
                                   double* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * N) return;
  const int64_t r = gid % N, c = gid / N;
  out[gid] = V[r + perm[c] * N];
}

static double sq_norm(mmg_ctx* ctx, const double* A, int64_t N, bool offdiag, double* dpart, std::vector<double>& hpart) {
  const int nb = (int)hpart.size();
  hipLaunchKernelGGL(sq_norm_partial_kernel, dim3(nb), dim3(256), 0, ctx->stream, A, N, offdiag ? 1 : 0, dpart);
  (void)hipMemcpyAsync(hpart.data(), dpart, nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  double s = 0.0;
  for (double v : hpart) s += v;
  return s;
}

// dA: column-major symmetric N x N on the device (destroyed).  On return the sorted eigenvalues are in `evals`
// (host) and, if evecs != nullptr, the eigenvectors as the columns of a column-major matrix (= rows of the
// row-major array the ABI returns) in `evecs` (host).  `block` = maximum rows per block.
int eigh_block_jacobi(mmg_ctx* ctx, rocblas_handle h, double* dA, int32_t N32, int32_t block, double* evals,
                      double* evecs, std::string& err) {
  Scratch sc;
  const int64_t N = N32;
  const int p = (int)((N + block - 1) / block);
  std::vector<int64_t> off(p + 1, 0);
  {
    const int64_t b = ((N + p - 1) / p + 63) / 64 * 64;        // equal blocks, multiples of 64 rows
    for (int i = 0; i <= p; ++i) off[i] = std::min<int64_t>(N, i * b);
  }
  int64_t maxpair = 0;
  for (int i = 0; i < p; ++i)
    for (int j = i + 1; j < p; ++j) maxpair = std::max(maxpair, (off[i + 1] - off[i]) + (off[j + 1] - off[j]));
  if (maxpair * maxpair >= (int64_t)1 << 31) { err = "eigh_block_jacobi: block pair exceeds the dsyevd range"; return MMG_E_ARG; }

  double *dV = nullptr, *dG = nullptr, *dT = nullptr, *dLam = nullptr, *dE = nullptr, *dpart = nullptr;
  rocblas_int* dinfo = nullptr;
  const int NB = 1024;
  std::vector<double> hpart(NB);
#define EB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x ": ") + hipGetErrorString(e_); return MMG_E_HIP; } } while (0)
#define EB_RB(x) do { rocblas_status s_ = (x); if (s_ != rocblas_status_success) { err = std::string(#x ": rocblas status ") + std::to_string((int)s_); return MMG_E_LIB; } } while (0)
  if (evecs) EB_HIP(sc.alloc(&dV, (size_t)N * N * sizeof(double)));
  EB_HIP(sc.alloc(&dG, (size_t)maxpair * maxpair * sizeof(double)));
  EB_HIP(sc.alloc(&dT, (size_t)N * maxpair * sizeof(double)));
  EB_HIP(sc.alloc(&dLam, maxpair * sizeof(double)));
  EB_HIP(sc.alloc(&dE, maxpair * sizeof(double)));
  EB_HIP(sc.alloc(&dpart, NB * sizeof(double)));
  EB_HIP(sc.alloc(&dinfo, sizeof(rocblas_int)));
  hipStream_t st = ctx->stream;
  const unsigned nbNN = (unsigned)((N * N + 255) / 256);
  if (evecs) hipLaunchKernelGGL(identity_kernel, dim3(nbNN), dim3(256), 0, st, dV, N);

  const double normA2 = sq_norm(ctx, dA, N, false, dpart, hpart);
  const double tol2 = 1e-24 * normA2;                          // off <= 1e-12 ||A||_F
  const double one = 1.0, zero = 0.0;
  auto copy_block = [&](double* dst, int64_t ldd, const double* src, int64_t lds_, int64_t rows, int64_t cols) {
    return hipMemcpy2DAsync(dst, ldd * sizeof(double), src, lds_ * sizeof(double), rows * sizeof(double), cols,
                            hipMemcpyDeviceToDevice, st);
  };
  // X[:, IJ] <- X[:, IJ] W  through the panel dT
  auto rotate_cols = [&](double* X, int64_t oI, int64_t nI, int64_t oJ, int64_t nJ) -> int {
    const int64_t n2 = nI + nJ;
    EB_RB(rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, N, n2, nI, &one, X + oI * N, N, dG, n2,
                           &zero, dT, N));
    EB_RB(rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, N, n2, nJ, &one, X + oJ * N, N, dG + nI,
                           n2, &one, dT, N));
    EB_HIP(hipMemcpyAsync(X + oI * N, dT, (size_t)N * nI * sizeof(double), hipMemcpyDeviceToDevice, st));
    EB_HIP(hipMemcpyAsync(X + oJ * N, dT + N * nI, (size_t)N * nJ * sizeof(double), hipMemcpyDeviceToDevice, st));
    return MMG_OK;
  };

  int sweeps = 0;
  double off2 = sq_norm(ctx, dA, N, true, dpart, hpart);
  while (off2 > tol2 && p > 1) {
    if (++sweeps > 20) { err = "eigh_block_jacobi: no convergence in 20 sweeps"; return MMG_E_LIB; }
    for (int I = 0; I < p; ++I)
      for (int J = I + 1; J < p; ++J) {
        const int64_t oI = off[I], nI = off[I + 1] - off[I], oJ = off[J], nJ = off[J + 1] - off[J], n2 = nI + nJ;
        if (nI == 0 || nJ == 0) continue;
        EB_HIP(copy_block(dG, n2, dA + oI + oI * N, N, nI, nI));
        EB_HIP(copy_block(dG + nI, n2, dA + oJ + oI * N, N, nJ, nI));
        EB_HIP(copy_block(dG + nI * n2, n2, dA + oI + oJ * N, N, nI, nJ));
        EB_HIP(copy_block(dG + nI + nI * n2, n2, dA + oJ + oJ * N, N, nJ, nJ));
        EB_RB(rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, (rocblas_int)n2, dG, (rocblas_int)n2, dLam,
                               dE, dinfo));
        rocblas_int info = 0;
        EB_HIP(hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, st));
        EB_HIP(hipStreamSynchronize(st));
        if (info != 0) { err = "rocsolver_dsyevd (block pair) did not converge: info " + std::to_string((int)info); return MMG_E_LIB; }
        int rc = rotate_cols(dA, oI, nI, oJ, nJ);
        if (rc) return rc;
        hipLaunchKernelGGL(sym_rows_from_cols_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)((nI + 31) / 32)), dim3(256), 0,
                           st, dA, N, oI, nI, oI, nI, oJ, nJ);
        hipLaunchKernelGGL(sym_rows_from_cols_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)((nJ + 31) / 32)), dim3(256), 0,
                           st, dA, N, oJ, nJ, oI, nI, oJ, nJ);
        hipLaunchKernelGGL(set_pair_block_kernel, dim3((unsigned)((n2 * n2 + 255) / 256)), dim3(256), 0, st, dA, N, oI, nI,
                           oJ, nJ, dLam);
        EB_HIP(hipGetLastError());
        if (evecs) {
          rc = rotate_cols(dV, oI, nI, oJ, nJ);
          if (rc) return rc;
        }
      }
    off2 = sq_norm(ctx, dA, N, true, dpart, hpart);
    if (std::getenv("MMG_EIGH_VERBOSE"))
      fprintf(stderr, "[eigh_block] N=%lld blocks=%d sweep %d: off/||A|| = %.3e\n", (long long)N, p, sweeps,
              std::sqrt(off2 / normA2));
  }
  if (p == 1) {                                               // degenerate call: a single block is a plain dsyevd
    EB_RB(rocsolver_dsyevd(h, evecs ? rocblas_evect_original : rocblas_evect_none, rocblas_fill_lower, N32, dA, N32, dLam,
                           dE, dinfo));
    EB_HIP(hipMemcpyAsync(evals, dLam, N * sizeof(double), hipMemcpyDeviceToHost, st));
    if (evecs) EB_HIP(hipMemcpyAsync(evecs, dA, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, st));
    EB_HIP(hipStreamSynchronize(st));
    return MMG_OK;
  }
  // ---- sort ascending (stable: ties keep their block order), gather the eigenvector columns
  double* dd = nullptr;
  EB_HIP(sc.alloc(&dd, N * sizeof(double)));
  hipLaunchKernelGGL(take_diag_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dA, N, dd);
  std::vector<double> d(N);
  EB_HIP(hipMemcpyAsync(d.data(), dd, N * sizeof(double), hipMemcpyDeviceToHost, st));
  EB_HIP(hipStreamSynchronize(st));
  std::vector<int64_t> perm(N);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return d[a] < d[b]; });
  for (int64_t i = 0; i < N; ++i) evals[i] = d[perm[i]];
  if (evecs) {
    int64_t* dperm = nullptr;
    EB_HIP(sc.alloc(&dperm, N * sizeof(int64_t)));
    EB_HIP(hipMemcpyAsync(dperm, perm.data(), N * sizeof(int64_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_cols_kernel, dim3(nbNN), dim3(256), 0, st, dV, N, dperm, dA);   // A is free now
    EB_HIP(hipGetLastError());
    EB_HIP(hipMemcpyAsync(evecs, dA, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, st));
    EB_HIP(hipStreamSynchronize(st));
  }
#undef EB_HIP
#undef EB_RB
  return MMG_OK;
}

}  // namespace mmg
