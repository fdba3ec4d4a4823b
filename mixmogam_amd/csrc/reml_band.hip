// reml_band.hip -- the REML likelihood sums of EVERY variance ratio from ONE orthogonal reduction of K.
//
// reml_chol.hip pays a Cholesky factorisation and a triangular inverse (2/3 N^3 flop) per delta: 57 of them for the
// 51-point grid + secant steps of get_estimates (linear_models.py:796-847) -- 91 s at N = 50,000, the largest stage of
// config 5 on one GPU.  The sums depend on K only through functions of K + delta I, and those commute with any
// orthogonal similarity: with K = Q B Q', B symmetric of bandwidth 64,
//     log|K + dI| = log|B + dI|,  tr (K + dI)^-1 = tr (B + dI)^-1,  z'(K + dI)^-k z = (Q'z)'(B + dI)^-k (Q'z),
// so K is reduced ONCE (blocked Householder band reduction, 4/3 N^3 flop of matrix-matrix work: one QR of the block column
// below the band + one symmetric rank-2b update of the trailing matrix per 64 columns) and every delta afterwards costs
// O(N b^2): a banded Cholesky factorisation, banded triangular solves for the q+1 rotated columns of [X y], and the
// band of the inverse (Takahashi's recurrence) for the trace.  All deltas of a call run side by side, one wavefront
// each; nothing per delta touches an N x N matrix.  The scan model at the chosen delta still comes from reml_chol.hip
// (it needs P itself).
//
// Reduction kernels: panel_qr_step_kernel (Householder QR of a tall 64-column panel, one launch per column),
// sym_skinny_kernel (A22 V from the lower triangle on the fp64 matrix pipe), tsmm_tn_kernel (tall-skinny products);
// the rank-128 update of the trailing matrix and the small products are rocBLAS GEMMs.
// Per-delta kernels (one workgroup per delta; lane t of a wave owns the columns / unknowns whose index is t mod 64, so
// the 64-column window that a step touches is spread over the lanes and never moves between registers):
//   band_factor_kernel  right-looking Cholesky of B + dI on four waves; the pivot column goes through LDS to the others
//   band_solve_kernel   forward and backward substitution per column of Q'[X y] as column sweeps, L streamed through LDS
//   band_trace_kernel   Z = (B + dI)^-1 inside the band, from the last column up; the 64 x 64 window of Z lives in LDS
// Roofline: the reduction is fp64 matrix work (own kernel at 32 TFLOP/s + rocBLAS); the per-delta kernels are latency
// chains of N steps (N = 50,000: ~0.1 s for the whole grid) -- off the SNPs/s metric either way.
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
#include "dense64_dev.h"
#include "reml_common.h"

namespace mmg {

constexpr int BAND_B = 64;
constexpr int BAND_LD = 72;        // doubles per stored band column: d = 0..64 the band, 65 = 1 / l_jj (factor only)

// V [n x nr] (ld n) <- the Householder vectors below the diagonal of the factored panel P (ld lda), unit diagonal,
// zeros above; the panel keeps only R
__global__ void band_build_v_kernel(double* __restrict__ P, int64_t lda, int n, int nr, double* __restrict__ V) {
  const int col = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  if (i == col) v = 1.0;
  else if (i > col) { v = P[i + (int64_t)col * lda]; P[i + (int64_t)col * lda] = 0.0; }
  V[i + (int64_t)col * n] = v;
}

// Bc[j][d] = A[j + d][j] (column-major A, lower triangle), zero beyond the matrix
__global__ __launch_bounds__(64) void band_extract_kernel(const double* __restrict__ A, int64_t N, double* __restrict__ Bc) {
  const int64_t j = blockIdx.x;
  for (int d = threadIdx.x; d < BAND_LD; d += 64)
    Bc[j * BAND_LD + d] = (d <= BAND_B && j + d < N) ? A[j * N + j + d] : 0.0;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- per-delta kernels -------------------------------------------------------------------------------------------
// value of `v` in lane `lane` (wave-uniform lane index): two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double lane_value(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// L (same layout as Bc) <- Cholesky factor of B + delta I; logdet = sum log(pivot); fail = 1 + first non-positive pivot.
// Four waves per delta: lane t of every wave owns column i = t (mod 64) of the 64-column window, wave w the rows
// d = 16 w .. 16 w + 15 of it (wave 3 also d = 64), so a step is 16-17 multiply-adds per lane after one barrier.
// The columns the lanes take over next (index + 64) wait in LDS (`pre`), staged a whole 64-step block ahead: all threads
// fetch the block after next at the start of a block and store it at its end -- no global load sits inside a step (a
// one-step-ahead fetch made every step as long as an HBM access: 1.2 us).
__global__ __launch_bounds__(256) void band_factor_kernel(const double* __restrict__ Bc, int N, const double* __restrict__ deltas,
                                                          double* __restrict__ Lall, double* __restrict__ logdet,
                                                          int* __restrict__ fail) {
  const int t = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d0 = 16 * w, nd = w == 3 ? 17 : 16;
  const double delta = deltas[blockIdx.x];
  double* Lc = Lall + (size_t)blockIdx.x * N * BAND_LD;
  __shared__ __attribute__((aligned(16))) double lbuf[2][2 * BAND_B + 8];   // [0..64] the raw pivot column, zeros above
  constexpr int PS = BAND_B + 2;
  constexpr int NPRE = (64 * PS + 255) / 256;                 // elements of a staged block per thread
  __shared__ __attribute__((aligned(16))) double pre[2][64 * PS];
  for (int i = threadIdx.x; i < 2 * (2 * BAND_B + 8); i += 256) (&lbuf[0][0])[i] = 0.0;
  auto band_at = [&](int i, int d) { return i < N ? Bc[(size_t)i * BAND_LD + d] + (d == 0 ? delta : 0.0) : 0.0; };
  // element e of the staged image of columns [c0, c0 + 64): slot e / PS, row e % PS (rows 65 are padding)
  auto stage_at = [&](int c0, int e) { const int sl = e / PS, d = e - sl * PS; return (e < 64 * PS && d <= BAND_B) ? band_at(c0 + sl, d) : 0.0; };
  double col[17], hold[NPRE];
#pragma unroll
  for (int k = 0; k < 17; ++k) col[k] = k < nd ? band_at(t, d0 + k) : 0.0;
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int e = threadIdx.x + 256 * k;
    if (e < 64 * PS) pre[0][e] = stage_at(BAND_B, e);          // columns 64..127: taken over during steps 0..63
  }
  double mant = 1.0;                                          // prod of pivots = mant 2^expo
  int expo = 0, bad = 0;
  __syncthreads();
  for (int J0 = 0; J0 < N && !bad; J0 += 64) {
    const int pb = (J0 >> 6) & 1;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) hold[k] = stage_at(J0 + 2 * BAND_B, threadIdx.x + 256 * k);   // for the NEXT block of steps
    const int jend = min(N, J0 + 64);
    for (int j = J0; j < jend; ++j) {
      const int p = j & 63, buf = j & 1;
      if (t == p) {
#pragma unroll
        for (int k = 0; k < 17; ++k)
          if (k < nd) lbuf[buf][d0 + k] = col[k];
      }
      __syncthreads();
      const double piv = lbuf[buf][0];
      if (!(piv > 0.0)) { bad = j + 1; break; }             // uniform: every lane reads the same pivot
      const double rinv = 1.0 / sqrt(piv);
      const int c = (t - j) & 63;
      if (w == 0) {
        int e;
        mant = frexp(mant * piv, &e);
        expo += e;
        // column j of L, one element per lane (lane 0 also the 65th), and 1 / l_jj beside it
        Lc[(size_t)j * BAND_LD + t] = lbuf[buf][t] * rinv;
        if (t == 0) {
          Lc[(size_t)j * BAND_LD + BAND_B] = lbuf[buf][BAND_B] * rinv;
          Lc[(size_t)j * BAND_LD + BAND_B + 1] = rinv;
        }
      }
      if (c == 0) {
        // the pivot's lanes move on to column j + 64: only its diagonal has met column j
        const double* src = &pre[pb][p * PS + d0];
#pragma unroll
        for (int k = 0; k < 17; ++k)
          if (k < nd) col[k] = src[k];
        if (w == 0) { const double x = lbuf[buf][BAND_B] * rinv; col[0] -= x * x; }
      } else {
        // column i = j + c:  A[i + d][i] -= l[c + d] l[c]   (lbuf is zero beyond 64: no bound to test)
        const double lc = lbuf[buf][c] * (rinv * rinv);
        const double* lb = &lbuf[buf][c + d0];
#pragma unroll
        for (int k = 0; k < 17; ++k)
          if (k < nd) col[k] = fma(-lb[k], lc, col[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = threadIdx.x + 256 * k;
      if (e < 64 * PS) pre[pb ^ 1][e] = hold[k];              // read from the first step of the next block on (a barrier later)
    }
  }
  if (threadIdx.x == 0) { logdet[blockIdx.x] = log(mant) + (double)expo * 0.6931471805599453094; fail[blockIdx.x] = bad; }
}

// The same factor in 64-column BLOCKS (round 6).  With bandwidth 64 the matrix is block tridiagonal in 64 x 64 blocks whose
// sub-diagonal blocks C_k are upper triangular, so the factorisation is
//     L_kk = chol(S_k),   L_{k+1,k} = C_k L_kk^-T  (upper triangular again),   S_{k+1} = A_{k+1} + delta I - L_{k+1,k} L_{k+1,k}',
// i.e. per 64 columns one blocked 64 x 64 Cholesky with its inverse (dense64_dev.h:chol64_lds: 8 barriers) and two
// triangular 64^3 products as 16 x 16 block products on the matrix pipe -- against 64 steps of LDS write -> barrier -> read.
// Same output layout (band of L per column, 1 / l_jj at d = 65, log-determinant, first bad pivot).
constexpr int BFAC_LDS = (4 * 64 * LD + 64 + 8) * (int)sizeof(double);
__global__ __launch_bounds__(256) void band_factor_blk_kernel(const double* __restrict__ Bc, int N, const double* __restrict__ deltas,
                                                              double* __restrict__ Lall, double* __restrict__ logdet,
                                                              int* __restrict__ fail) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *As = sm, *Fs = sm + 64 * LD, *Xs = sm + 2 * 64 * LD, *T1 = sm + 3 * 64 * LD, *rs = sm + 4 * 64 * LD;
  int* sh = (int*)(rs + 64);
  const int tid = threadIdx.x, i = tid & 63, kq = tid >> 6;
  const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  const double delta = deltas[blockIdx.x];
  double* Lc = Lall + (size_t)blockIdx.x * N * BAND_LD;
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  // thread (i, kq): entries (row i, columns 16 kq + m) of a block.  Diagonal block k: A[64k + i][64k + c] for i >= c (identity
  // beyond N); sub-diagonal block k: C[i][c] = A[64(k+1) + i][64k + c] for c >= i.
  auto diag_at = [&](int k, int c) -> double {
    const int col = 64 * k + c, d = i - c;
    if (d < 0) return 0.0;
    if (64 * k + i >= N) return d == 0 ? 1.0 : 0.0;
    return Bc[(size_t)col * BAND_LD + d] + (d == 0 ? delta : 0.0);
  };
  auto sub_at = [&](int k, int c) -> double {
    const int col = 64 * k + c, d = 64 + i - c;
    return (c >= i && 64 * (k + 1) + i < N) ? Bc[(size_t)col * BAND_LD + d] : 0.0;
  };
  const int nblk = (N + 63) / 64;
#pragma unroll
  for (int m = 0; m < 16; ++m) As[i * LD + 16 * kq + m] = diag_at(0, 16 * kq + m);
  double lsum = 0.0;                                           // threads < 64: sum of log pivots of their column index
  int bad = 0;
  __syncthreads();
  for (int k = 0; k < nblk; ++k) {
    double cpre[16], apre[16];                                 // the next blocks' band entries travel while this one is factored
    const bool more = k + 1 < nblk;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      cpre[m] = more ? sub_at(k, 16 * kq + m) : 0.0;
      apre[m] = more ? diag_at(k + 1, 16 * kq + m) : 0.0;
    }
    const int b = chol64_lds(As, Fs, Xs, T1, rs, sh, tid);
    if (b) { bad = 64 * k + b; break; }                        // uniform
    // columns 64k .. 64k + 63 of L inside the diagonal block: L[r][c] = u_rc rs_c (r > c), p_c rs_c on the diagonal; 1 / l_cc
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int c = 16 * kq + m, col = 64 * k + c;
      // (rows beyond N are the identity padding: their entries in the columns < N are exact zeros, written like the rest)
      if (i >= c && col < N) Lc[(size_t)col * BAND_LD + (i - c)] = As[i * LD + c] * rs[c];
      if (!more && c >= i && col < N) Lc[(size_t)col * BAND_LD + (64 + i - c)] = 0.0;   // last block: nothing below it
    }
    if (tid < 64 && 64 * k + tid < N) {
      Lc[(size_t)(64 * k + tid) * BAND_LD + BAND_B + 1] = rs[tid];
      lsum += log(As[tid * LD + tid]);
    }
    if (!more) break;
    // X = C L_kk^-T: X[a][j] = rs_j sum_m C[a][m] Lu^-1[j][m]  (C upper, Lu^-1 lower: X upper triangular) -> Fs
    __syncthreads();                                           // (everybody has read As / rs for the write-out above)
#pragma unroll
    for (int m = 0; m < 16; ++m) Xs[i * LD + 16 * kq + m] = cpre[m];
    __syncthreads();
    for (int t = w; t < 16; t += 4) {
      const int ib = t >> 2, jb = t & 3;
      v4d acc = zero;
      for (int mb = ib; mb <= jb; ++mb) acc = blk_mma_nt(acc, Xs, 16 * ib, 16 * mb, T1, 16 * jb, 16 * mb, lr, lk, 1.0);
      const double sc = rs[16 * jb + lr];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[rr] *= sc;
      blk_store(Fs, 16 * ib, 16 * jb, lr, lk, acc);            // (zeros for jb < ib: the loop above is empty)
    }
    __syncthreads();
    // rows 64(k+1) + a of columns 64k + j (a <= j): d = 64 + a - j; and S_{k+1} = A_{k+1} + delta I - X X' (lower blocks) -> As
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int c = 16 * kq + m;
      if (c >= i) Lc[(size_t)(64 * k + c) * BAND_LD + (64 + i - c)] = Fs[i * LD + c];   // (column < N: there is a next block)
      As[i * LD + c] = apre[m];
    }
    __syncthreads();
    for (int t = w; t < 10; t += 4) {                          // lower blocks (r >= c) of the 4 x 4 block matrix
      const int r = t >= 6 ? 3 : (t >= 3 ? 2 : (t >= 1 ? 1 : 0)), c = t - r * (r + 1) / 2;
      v4d acc = blk_load(As, 16 * r, 16 * c, lr, lk);
      for (int jb = r; jb < 4; ++jb) acc = blk_mma_nt(acc, Fs, 16 * r, 16 * jb, Fs, 16 * c, 16 * jb, lr, lk, -1.0);
      blk_store(As, 16 * r, 16 * c, lr, lk, acc);
    }
    __syncthreads();
  }
  // log-determinant: the 64 partial sums through LDS (rs is free now)
  __syncthreads();
  if (tid < 64) rs[tid] = lsum;
  __syncthreads();
  if (tid == 0) {
    double sum = 0.0;
    for (int c = 0; c < 64; ++c) sum += rs[c];
    logdet[blockIdx.x] = sum;
    fail[blockIdx.x] = bad;
  }
}

static bool band_factor_blocked() {
  static const bool on = [] { const char* e = std::getenv("MMG_BAND_FACTOR"); return !(e && std::string(e) == "chain"); }();   // chain: round 5's kernel (A/B)
  return on;
}
static int launch_band_factor(hipStream_t st, int ng, const double* dBand, int N, const double* dd, double* L, double* dlog, int* dfail) {
  if (band_factor_blocked()) {
    // (per call, not once per process: the attribute belongs to the device the calling context is bound to)
    if (hipFuncSetAttribute((const void*)band_factor_blk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BFAC_LDS) != hipSuccess) return 1;
    hipLaunchKernelGGL(band_factor_blk_kernel, dim3(ng), dim3(256), BFAC_LDS, st, dBand, N, dd, L, dlog, dfail);
  } else {
    hipLaunchKernelGGL(band_factor_kernel, dim3(ng), dim3(256), 0, st, dBand, N, dd, L, dlog, dfail);
  }
  return 0;
}

// sum over the wave, the same value in every lane: butterflies inside the 16-lane rows by DPP, the four row sums through
// v_readlane (a ds_bpermute butterfly is six dependent LDS round trips)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_fast(double v) {
  v += dpp_move<0xB1>(v);                                     // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);                                     // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);                                    // row_half_mirror
  v += dpp_move<0x140>(v);                                    // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// F[r] = L^-1 z_r, G[r] = L^-T F[r] for the q1 columns z_r of Zr ([q1][N]).  Wave 0 does the arithmetic; both
// substitutions are column sweeps (x_j leaves lane j mod 64 through v_readlane, every other lane subtracts its L entry
// times it), so a step is one LDS read and one multiply-add per lane and there is no reduction.  The backward sweep needs
// ROW j of L -- entries of the 64 columns before j -- hence two 64-column chunks of L at a time.  Waves 1-3 fetch the
// chunk needed next into the third buffer of a ring meanwhile (one wave loading 36 KB per 64 steps by itself spent
// more time fetching than solving).
// sel (or nullptr): workgroup g works on factor sel[g] of Lall (factors kept by mmg_reml_band_factor) and writes result g
__global__ __launch_bounds__(256) void band_solve_kernel(const double* __restrict__ Lall, int N, const double* __restrict__ Zr,
                                                         int q1, double* __restrict__ Fall, double* __restrict__ Gall,
                                                         const int* __restrict__ sel) {
  // Round 6: two right-hand sides at a time, one per wave (waves 0 and 1 walk the same chunks in lock step; q1 = 2 for the
  // intercept-only model: one round instead of two, 1.8 -> ~1.0 ms at N = 5000); waves 2 and 3 fetch.
  const int t = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
  const bool loader = wv_ >= 2;
  const int lt = threadIdx.x - 128;
  const double* Lc = Lall + (size_t)(sel ? sel[blockIdx.x] : (int)blockIdx.x) * N * BAND_LD;
  __shared__ __attribute__((aligned(16))) double ch[3][64 * BAND_LD];
  const int nchunk = (N + 63) / 64;
  auto ring = [](int cb) { return (cb + 3) % 3; };            // cb >= -2
  auto load_chunk = [&](int cb) {                             // loaders: chunk cb (columns 64 cb ..) -> its ring slot; zeros beyond the matrix
    double2* dst = (double2*)ch[ring(cb)];
    const int nj = (cb < 0 || cb >= nchunk) ? 0 : min(64, N - cb * 64);
    const double2* src = (const double2*)(Lc + (size_t)(cb < 0 ? 0 : cb) * 64 * BAND_LD);
    for (int i = lt; i < 32 * BAND_LD; i += 128) dst[i] = 2 * i < nj * BAND_LD ? src[i] : double2{0.0, 0.0};
  };
  for (int r0 = 0; r0 < q1; r0 += 2) {
    const int r = r0 + wv_;                                    // this wave's right-hand side (compute waves)
    const bool solver = !loader && r < q1;                     // (a compute wave without one only keeps the barriers)
    const int rr_ = r < q1 ? r : 0;
    const double* z0 = Zr + (size_t)rr_ * N;
    double* f = Fall + ((size_t)blockIdx.x * q1 + rr_) * N;
    double* g = Gall + ((size_t)blockIdx.x * q1 + rr_) * N;
    // ---- forward: lane t holds the running right-hand side of unknown i = t (mod 64) of the current window
    __syncthreads();
    if (loader) load_chunk(0);
    __syncthreads();
    double z = (solver && t < N) ? z0[t] : 0.0;
    for (int cb = 0; cb < nchunk; ++cb) {
      if (loader) {
        load_chunk(cb + 1);
      } else if (solver) {
        const int j0 = cb * 64, nj = min(64, N - j0);
        const double* cur = ch[ring(cb)];
        const double znext = (j0 + 64 + t < N) ? z0[j0 + 64 + t] : 0.0;
        double wkeep = 0.0;
        for (int jj = 0; jj < nj; ++jj) {
          const int c = (t - jj) & 63;
          const double* lj = cur + jj * BAND_LD;
          const double lval = lj[c == 0 ? BAND_B : c];
          const double wv = lane_value(z, jj) * lj[BAND_B + 1];
          if (c == 0) { wkeep = wv; z = fma(-lval, wv, znext); }
          else z = fma(-lval, wv, z);
        }
        if (t < nj) f[j0 + t] = wkeep;
      }
      __syncthreads();
    }
    // ---- backward: lane t holds the running right-hand side of unknown i = t (mod 64) of the window BELOW column j
    // (f is read back by the thread that wrote it: index = t mod 64)
    const int top = nchunk - 1, top0 = top * 64, ntop = N - top0;
    if (loader) { load_chunk(top); load_chunk(top - 1); }
    __syncthreads();
    double x = 0.0;
    if (solver) x = t < ntop ? f[top0 + t] : (top0 + t - 64 >= 0 ? f[top0 + t - 64] : 0.0);
    for (int cb = top; cb >= 0; --cb) {
      if (loader) {
        load_chunk(cb - 2);
      } else if (solver) {
        const int j0 = cb * 64, nj = min(64, N - j0);
        const double* cur = ch[ring(cb)];
        const double* low = ch[ring(cb - 1)];
        const double xnext = (cb >= 1 && t < nj) ? f[j0 - 64 + t] : 0.0;
        double xkeep = 0.0;
        for (int jj = nj - 1; jj >= 0; --jj) {
          const int c = (jj - t) & 63;                        // j - i for the unknown i this lane holds
          // L[j][i]: column i lives at local column t of its chunk (this one for t < jj, the one below otherwise)
          const double lval = c == 0 ? low[jj * BAND_LD + BAND_B] : (t < jj ? cur : low)[t * BAND_LD + c];
          const double xj = lane_value(x, jj) * cur[jj * BAND_LD + BAND_B + 1];
          if (c == 0) { xkeep = xj; x = fma(-lval, xj, xnext); }
          else x = fma(-lval, xj, x);
        }
        if (t < nj) g[j0 + t] = xkeep;
      }
      __syncthreads();
    }
  }
}

// trace of (L L')^-1 from the band of the inverse:  Z_ij = -(1/l_jj) sum_{k>j} Z_ik L_kj (i > j),
// Z_jj = 1/l_jj^2 - (1/l_jj) sum_{k>j} L_kj Z_kj;  row / column i of the window lives at slot i mod 64.  Four waves: wave w
// sums over the slots 16 w .. 16 w + 15 (L entries through v_readlane), the partial sums meet in LDS.
__global__ __launch_bounds__(256) void band_trace_kernel(const double* __restrict__ Lall, int N, double* __restrict__ trace,
                                                         const int* __restrict__ sel) {
  const int t = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* Lc = Lall + (size_t)(sel ? sel[blockIdx.x] : (int)blockIdx.x) * N * BAND_LD;
  constexpr int ZS = 65;                                      // row stride of the window: column writes hit 64 banks
  __shared__ double Zw[64 * ZS];
  __shared__ double ch[64 * BAND_LD];
  __shared__ double psum[4][64];
  for (int i = threadIdx.x; i < 64 * ZS; i += 256) Zw[i] = 0.0;
  double tr = 0.0;
  const int nchunk = (N + 63) / 64;
  for (int cb = nchunk - 1; cb >= 0; --cb) {
    const int j0 = cb * 64, nj = min(64, N - j0);
    __syncthreads();
    for (int i = threadIdx.x; i < nj * BAND_LD; i += 256) ch[i] = Lc[(size_t)j0 * BAND_LD + i];
    __syncthreads();
    for (int jj = nj - 1; jj >= 0; --jj) {
      const int c = (t - jj) & 63;
      const double* lj = ch + jj * BAND_LD;
      const double rinv = lj[BAND_B + 1];
      const double lv = lj[c == 0 ? BAND_B : c];              // L[i_t][j], i_t = j + (c ? c : 64)
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        s0 = fma(lane_value(lv, 16 * w + u), Zw[(16 * w + u) * ZS + t], s0);
        s1 = fma(lane_value(lv, 16 * w + u + 1), Zw[(16 * w + u + 1) * ZS + t], s1);
      }
      psum[w][t] = s0 + s1;
      __syncthreads();                                        // every wave has read the window
      const double zt = -((psum[0][t] + psum[1][t]) + (psum[2][t] + psum[3][t])) * rinv;   // Z[i_t][j]
      const double dsum = wave_sum_fast(lv * zt);
      const double zjj = rinv * rinv - rinv * dsum;
      tr += zjj;
      const double put = (c == 0) ? zjj : zt;                 // slot jj changes owner: index j + 64 leaves, j enters
      if (w == 0) Zw[jj * ZS + t] = put;
      if (w == 1) Zw[t * ZS + jj] = put;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) trace[blockIdx.x] = tr;
}

// out[blk][a][b] = <F_a, F_b>, out2 likewise for G: one block per (delta, a, b), fixed summation order
// The trace in 64-column blocks (round 6).  L is block bidiagonal (diagonal blocks L_k lower, sub-diagonal blocks M_k upper
// triangular), and Z = (L L')^-1 restricted to the block pattern obeys, from the last block up,
//     G_k = M_k L_k^-1,     Z_kk = L_k^-T L_k^-1 + G_k' Z_{k+1,k+1} G_k,
// so a block costs one inverse of a 64 x 64 triangle (unit_lower_inverse64) and ~190 block products of 16^3 on the matrix pipe
// instead of 64 steps of (16 readlanes + LDS round trip + two barriers).  tr (B + delta I)^-1 = sum_k tr Z_kk.
constexpr int BTR_LDS = (4 * 64 * LD + 64 + 8) * (int)sizeof(double);
__global__ __launch_bounds__(256) void band_trace_blk_kernel(const double* __restrict__ Lall, int N, double* __restrict__ trace,
                                                             const int* __restrict__ sel) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *Fs = sm, *Xs = sm + 64 * LD, *T1 = sm + 2 * 64 * LD, *Zs = sm + 3 * 64 * LD, *rs = sm + 4 * 64 * LD;
  const int tid = threadIdx.x, i = tid & 63, kq = tid >> 6;
  const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  const double* Lc = Lall + (size_t)(sel ? sel[blockIdx.x] : (int)blockIdx.x) * N * BAND_LD;
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  const int nblk = (N + 63) / 64;
  for (int e = tid; e < 64 * LD; e += 256) Zs[e] = 0.0;
  double tr = 0.0;
  for (int k = nblk - 1; k >= 0; --k) {
    // F = strict lower triangle of Lu = L_k diag(1 / l_jj); rows / columns beyond N: the identity
    double mpre[16];
    if (tid < 64) rs[tid] = 64 * k + tid < N ? Lc[(size_t)(64 * k + tid) * BAND_LD + BAND_B + 1] : 1.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int c = 16 * kq + m, col = 64 * k + c;
      const bool in = col < N && 64 * k + i < N;
      Fs[i * LD + c] = (i > c && in) ? Lc[(size_t)col * BAND_LD + (i - c)] * Lc[(size_t)col * BAND_LD + BAND_B + 1] : 0.0;
      // M_k[a = i][c] = L[64(k+1) + i][64k + c], c >= i (zero for the last block and beyond N)
      mpre[m] = (c >= i && k + 1 < nblk && col < N && 64 * (k + 1) + i < N) ? Lc[(size_t)col * BAND_LD + (64 + i - c)] : 0.0;
    }
    __syncthreads();
    unit_lower_inverse64(Fs, Xs, T1, tid);                      // T1 = Lu^-1 (ends with a barrier)
    {
      const double ri = rs[i];                                 // L_k^-1 = diag(1 / l_ii) Lu^-1
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        T1[i * LD + 16 * kq + m] *= ri;
        Fs[i * LD + 16 * kq + m] = mpre[m];                    // M (F is dead)
      }
    }
    __syncthreads();
    for (int t = w; t < 16; t += 4) {                          // G = M L^-1 -> Xs: blocks (a, j), contraction blocks >= max(a, j)
      const int ab = t >> 2, jb = t & 3;
      v4d acc = zero;
      for (int mb = ab > jb ? ab : jb; mb < 4; ++mb) acc = blk_mma(acc, Fs, 16 * ab, 16 * mb, T1, 16 * mb, 16 * jb, lr, lk, 1.0);
      blk_store(Xs, 16 * ab, 16 * jb, lr, lk, acc);
    }
    __syncthreads();
    for (int t = w; t < 16; t += 4) {                          // W = Z G -> Fs (M is dead)
      const int ab = t >> 2, jb = t & 3;
      v4d acc = zero;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc = blk_mma(acc, Zs, 16 * ab, 16 * mb, Xs, 16 * mb, 16 * jb, lr, lk, 1.0);
      blk_store(Fs, 16 * ab, 16 * jb, lr, lk, acc);
    }
    __syncthreads();
    v4d zk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                              // Z_kk = L^-T L^-1 + G' W: block (ib, jb) = (u, w) per wave
      const int ib = u, jb = w;
      v4d acc = zero;
      for (int mb = ib > jb ? ib : jb; mb < 4; ++mb) acc = blk_mma_tn(acc, T1, 16 * mb, 16 * ib, T1, 16 * mb, 16 * jb, lr, lk, 1.0);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc = blk_mma_tn(acc, Xs, 16 * mb, 16 * ib, Fs, 16 * mb, 16 * jb, lr, lk, 1.0);
      zk[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) blk_store(Zs, 16 * u, 16 * w, lr, lk, zk[u]);   // (Z's last readers passed the barrier above)
    __syncthreads();
    if (tid < 64 && 64 * k + tid < N) tr += Zs[tid * LD + tid];
  }
  __syncthreads();
  if (tid < 64) rs[tid] = tr;
  __syncthreads();
  if (tid == 0) {
    double sum = 0.0;
    for (int c = 0; c < 64; ++c) sum += rs[c];
    trace[blockIdx.x] = sum;
  }
}
static bool band_trace_blocked() {
  static const bool on = [] { const char* e = std::getenv("MMG_BAND_TRACE"); return !(e && std::string(e) == "chain"); }();   // chain: round 5's kernel (A/B)
  return on;
}
static int launch_band_trace(hipStream_t st, int ng, const double* L, int N, double* dtr, const int* dsel) {
  if (band_trace_blocked()) {
    if (hipFuncSetAttribute((const void*)band_trace_blk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BTR_LDS) != hipSuccess) return 1;
    hipLaunchKernelGGL(band_trace_blk_kernel, dim3(ng), dim3(256), BTR_LDS, st, L, N, dtr, dsel);
  } else {
    hipLaunchKernelGGL(band_trace_kernel, dim3(ng), dim3(256), 0, st, L, N, dtr, dsel);
  }
  return 0;
}

__global__ __launch_bounds__(256) void band_gram_kernel(const double* __restrict__ Fall, const double* __restrict__ Gall, int N,
                                                        int q1, double* __restrict__ ff, double* __restrict__ gg) {
  const int blk = blockIdx.x, a = blockIdx.y / q1, b = blockIdx.y % q1;
  if (b > a) return;
  const double* fa = Fall + ((size_t)blk * q1 + a) * N;
  const double* fb = Fall + ((size_t)blk * q1 + b) * N;
  const double* ga = Gall + ((size_t)blk * q1 + a) * N;
  const double* gb = Gall + ((size_t)blk * q1 + b) * N;
  double sf = 0.0, sg = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) { sf = fma(fa[i], fb[i], sf); sg = fma(ga[i], gb[i], sg); }
  sf = wave_sum(sf); sg = wave_sum(sg);
  __shared__ double w[8];
  if ((threadIdx.x & 63) == 0) { w[threadIdx.x >> 6] = sf; w[4 + (threadIdx.x >> 6)] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double vf = (w[0] + w[1]) + (w[2] + w[3]), vg = (w[4] + w[5]) + (w[6] + w[7]);
    ff[((size_t)blk * q1 + a) * q1 + b] = ff[((size_t)blk * q1 + b) * q1 + a] = vf;
    gg[((size_t)blk * q1 + a) * q1 + b] = gg[((size_t)blk * q1 + b) * q1 + a] = vg;
  }
}

// ---- kernels of the reduction -------------------------------------------------------------------------------------

// Tall-skinny products  out [64 x kb] = A' B  (A [n x 64] ld lda, B [n x kb] ld ldb, kb <= 64): rocBLAS runs an output
// this small on a handful of workgroups with the whole contraction inside each (0.4 ms at n = 20,000).  Stage 1: one
// workgroup per slice of rows -> part[g][64 x 64]; stage 2 adds the slices in a fixed order.
constexpr int TS_ROWS = 32;
__global__ __launch_bounds__(256) void tsmm_tn_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                                      int64_t ldb, int kb, int n, int rows_per, double* __restrict__ part) {
  __shared__ double As[TS_ROWS][65], Bs[TS_ROWS][65];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int r_begin = blockIdx.x * rows_per, r_end = min(n, r_begin + rows_per);
  double acc[4][4] = {};
  for (int r0 = r_begin; r0 < r_end; r0 += TS_ROWS) {
    const int rr = tid & 31, c0 = tid >> 5;
    const bool in = r0 + rr < r_end;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + 8 * k;
      As[rr][c] = in ? A[(r0 + rr) + (int64_t)c * lda] : 0.0;
      Bs[rr][c] = (in && c < kb) ? B[(r0 + rr) + (int64_t)c * ldb] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < TS_ROWS; ++q) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[q][ty * 4 + i]; bv[i] = Bs[q][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = fma(av[i], bv[k], acc[i][k]);
    }
    __syncthreads();
  }
  double* o = part + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) o[(ty * 4 + i) + 64 * (tx * 4 + k)] = acc[i][k];
}

__global__ __launch_bounds__(256) void tsmm_reduce_kernel(const double* __restrict__ part, int G, double* __restrict__ out, int ldo) {
  const int e = blockIdx.x * 256 + threadIdx.x;                // element of the 64 x 64 result, column-major
  double s = 0.0;
  for (int g = 0; g < G; ++g) s += part[(size_t)g * 4096 + e];
  out[(e & 63) + ldo * (e >> 6)] = s;
}

// Householder QR of a tall panel P [n x 64], one launch per column (j = -1 prepares column 0).  Launch j
//   * finishes reflector j from what launch j - 1 left behind: dots[c] = sum_{r > j} P[r][j] P[r][c] (summed per
//     workgroup into part_in) and the pivot row (pivrow_in) give the norm, beta, tau and w_c = v'P[:, c] without
//     another pass over the panel;
//   * applies it to the workgroup's 256 rows, columns j+1..63, and in the same sweep accumulates the dots of column
//     j + 1 of the UPDATED panel (part_out) and copies its pivot row (pivrow_out);
//   * writes column j of V (explicit: zeros above, 1 on the diagonal) and zeroes P below the diagonal.
// rocsolver_dgeqrf spends 13.8 ms per 50,000 x 64 panel (some 300 small launches); this is 65 launches that each read
// and write the remaining columns once.  (The same 65 steps as ONE cooperative launch with grid-wide barriers and
// agent-scope loads of the exchanged sums were slower: panel QR + T 1.78 against 1.27 s at N = 50,000, 0.50 against 0.42
// at 20,000 -- a grid barrier over ~200 workgroups costs more than the kernel boundary it replaces.)
constexpr int QR_ROWS = 256;
constexpr double HH_TINY2 = 1e-200;                          // (norm below which a Householder column counts as zero)^2
__global__ __launch_bounds__(256) void panel_qr_step_kernel(double* __restrict__ P, int64_t lda, int n, int j,
                                                            const double* __restrict__ part_in, double* __restrict__ part_out,
                                                            const double* __restrict__ pivrow_in, double* __restrict__ pivrow_out,
                                                            int G, double* __restrict__ V, double* __restrict__ tau,
                                                            double* __restrict__ rdiag) {
  __shared__ double sred[4][64], sdots[64], sw[64];
  const int tid = threadIdx.x, rl = tid & 63, cg = tid >> 6;
  const int r0 = blockIdx.x * QR_ROWS;
  double scale = 0.0;
  sw[rl] = 0.0;
  if (j >= 0) {
    double s = 0.0;
    for (int g = cg; g < G; g += 4) s += part_in[(size_t)g * 64 + rl];
    sred[cg][rl] = s;
    __syncthreads();
    if (tid < 64) sdots[tid] = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
    __syncthreads();
    const double x0 = pivrow_in[j], xn2 = sdots[j];
    double tauj = 0.0, beta = x0;
    // A column whose norm is below HH_TINY is left alone (H = I; what lies below its diagonal is dropped: an error of
    // 1e-100 on a matrix of O(1) entries).  In a rank-deficient panel every column past the rank is the rounding noise
    // of the one before it -- 1e-13, 1e-26, ... -- and past 1e-154 the squares of its entries underflow: the norm that
    // defines tau no longer matches the column that defines v, and the "reflector" stops being orthogonal (a kinship of
    // rank 3 at N = 1024: tau |v|^2 = 2.59 at column 61, REML sums off by 2e-5; LAPACK's dlarfg rescales instead).
    if (fma(x0, x0, xn2) > HH_TINY2) {
      beta = -copysign(sqrt(fma(x0, x0, xn2)), x0);
      tauj = (beta - x0) / beta;
      scale = 1.0 / (x0 - beta);
    }
    if (tid < 64) sw[tid] = tid > j ? tauj * fma(sdots[tid], scale, pivrow_in[tid]) : 0.0;     // tau w_c
    if (blockIdx.x == 0 && tid == 0) { tau[j] = tauj; rdiag[j] = beta; }
  }
  __syncthreads();
  const int jn = j + 1;
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  // v_r and the updated element of column j + 1 of this thread's rows, read before any wave writes that column
  double vrs[QR_ROWS / 64], pns[QR_ROWS / 64];
#pragma unroll
  for (int k = 0; k < QR_ROWS / 64; ++k) {
    const int r = r0 + rl + 64 * k;
    vrs[k] = pns[k] = 0.0;
    if (r >= n || r < j) continue;
    vrs[k] = j < 0 ? 0.0 : (r == j ? 1.0 : P[r + (int64_t)j * lda] * scale);
    pns[k] = jn < 64 ? fma(-vrs[k], sw[jn], P[r + (int64_t)jn * lda]) : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < QR_ROWS / 64; ++k) {
    const int r = r0 + rl + 64 * k;
    if (r >= n || r < j) continue;
    const double vr = vrs[k], pn = pns[k];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = jn + cg + 4 * i;
      if (c < 64) {
        const double val = fma(-vr, sw[c], P[r + (int64_t)c * lda]);
        if (j >= 0) P[r + (int64_t)c * lda] = val;
        if (r == jn) pivrow_out[c] = val;
        if (r > jn) acc[i] = fma(pn, val, acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const double s = wave_sum(acc[i]);
    const int c = jn + cg + 4 * i;
    if (rl == 0 && c < 64) part_out[(size_t)blockIdx.x * 64 + c] = s;
  }
  __syncthreads();                                             // every wave has read column j of its rows
  if (j >= 0 && cg == 0) {
    for (int k = 0; k < QR_ROWS / 64; ++k) {
      const int r = r0 + rl + 64 * k;
      if (r >= n) continue;
      double vr = 0.0;
      if (r == j) vr = 1.0;
      else if (r > j) { vr = P[r + (int64_t)j * lda] * scale; P[r + (int64_t)j * lda] = 0.0; }
      V[r + (int64_t)j * n] = vr;
    }
  }
}

__global__ void panel_qr_finish_kernel(double* __restrict__ P, int64_t lda, const double* __restrict__ rdiag) {
  const int j = threadIdx.x;
  P[j + (int64_t)j * lda] = rdiag[j];
}

// T (upper triangular, ld 64) of the compact WY form Q = I - V T V' from S = V'V and tau (LAPACK dlarft, forward /
// columnwise):  T[0:j, j] = -tau_j T[0:j, 0:j] S[0:j, j],  T[j][j] = tau_j
__global__ __launch_bounds__(64) void form_t_kernel(const double* __restrict__ S, const double* __restrict__ tau, double* __restrict__ T) {
  __shared__ double Ts[64][65];
  const int i = threadIdx.x;
  for (int c = 0; c < 64; ++c) Ts[i][c] = 0.0;
  __syncthreads();
  for (int j = 0; j < 64; ++j) {
    const double tj = tau[j];
    double v = 0.0;
    if (i < j) {
      for (int k = i; k < j; ++k) v = fma(Ts[i][k], S[k + 64 * j], v);
      v *= -tj;
    } else if (i == j) v = tj;
    __syncthreads();
    Ts[i][j] = v;
    __syncthreads();
  }
  for (int c = 0; c < 64; ++c) T[i + 64 * c] = Ts[i][c];
}

// Cm [128 x 64] (ld 128) = [ -1/2 S2 ; T ]:  Y = [V | A22 V] Cm = A22 V T - 1/2 V (T'V'A22 V T)
__global__ __launch_bounds__(256) void form_coef_kernel(const double* __restrict__ S2, const double* __restrict__ T, double* __restrict__ Cm) {
  const int e = blockIdx.x * 256 + threadIdx.x;                // 0..4095
  const int i = e & 63, c = e >> 6;
  Cm[i + 128 * c] = -0.5 * S2[i + 64 * c];
  Cm[64 + i + 128 * c] = T[i + 64 * c];
}

// W [n x 64] = A V for symmetric A [n x n] of which only the LOWER triangle is read (column-major, ld lda), V [n x 64]
// (ld n), on v_mfma_f64_16x16x4_f64.  rocBLAS runs this shape (64 output columns) at 10-15 TFLOP/s (dsymm_64: 8.4 s of
// the 32.7 s reduction at N = 50,000; the block-row dgemm form is slower still).  One workgroup per 64 output rows walks
// the 64 x 64 tiles of its block row: left of the diagonal as stored, right of it the mirrored tile read transposed
// (both contiguous in memory), the diagonal tile completed from its lower half.  Wave w owns output columns
// 16 w .. 16 w + 15 (4 accumulator tiles of 16 rows).  Operand fragments (lane l: row / column l % 16, k = l / 16):
//   stored tile      As[k][row], row stride 80 doubles  -> the four 16-lane groups of a ds_read_b64 start 640 B apart
//   transposed tile  As[i][k],   row stride 66 doubles  -> bank (in 8-byte units) 2 i + k: distinct per half-wave
//   V tile           Vt[col][k], row stride 66
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int SY_SA = 80, SY_ST = 66;
// blockIdx.y = slice of the contraction range (tiles [ntile y / S, ntile (y + 1) / S)) writing its own copy of W: at N = 5000
// one workgroup per 64 output rows is 40-78 workgroups on 256 CUs, each a chain of ~40 tile steps
// TRI: A is LOWER TRIANGULAR instead of symmetric (the triangular inverse of reml_chol.hip: X21 = -X22 (L21 X11)): the tiles
// right of the diagonal are skipped and the diagonal tile is its lower half alone.
// Round 6: interior tiles (all 64 rows and all 64 k inside the matrix -- every tile but those of the last block row / column)
// are fetched by 32 unconditional loads per thread and multiplied from LDS with the fragments of step ks + 1 read while the
// MFMAs of step ks run.  Before, every load carried its own bounds predicate: the compiler turned the 32 loads of a tile into
// as many exec-mask branches with a vmcnt(0) wait each, and the kernel ran its memory phase and its MFMA phase one after the
// other (timing ablations at n = 4936: 95 us as it was, 59 without MFMAs, 82 without loads; the MFMAs alone take 41).
// ABL (make EXPERIMENTS=1, MMG_SYM_ABL; WRONG results): 1 = no MFMAs, 2 = no global loads after the first tile.
template <bool TRI, int ABL = 0>
__global__ __launch_bounds__(256, 2) void sym_skinny_kernel(const double* __restrict__ A, int64_t lda, int n,
                                                            const double* __restrict__ V, double* __restrict__ Wout) {
  __shared__ double As[64 * SY_SA];
  __shared__ double Vt[64 * SY_ST];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int I0 = blockIdx.x * 64;
  const int lr = l & 15, lk = l >> 4;
  v4d acc[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = v4d{0.0, 0.0, 0.0, 0.0};
  const int ntile = TRI ? blockIdx.x + 1 : (n + 63) / 64;      // TRI: only the tiles up to the diagonal
  const int kt0 = (int)((int64_t)ntile * blockIdx.y / gridDim.y), kt1 = (int)((int64_t)ntile * (blockIdx.y + 1) / gridDim.y);
  double* __restrict__ W = Wout + (size_t)blockIdx.y * n * 64;
  // the next tile pair travels in registers while the current one is multiplied (element e = tid + 256 i of a tile)
  double ra[16], rv[16];
  // 32-bit element offsets from wave-uniform tile bases (scalar base + one VGPR per address)
  const int lo = tid & 63, h0 = tid >> 6;
  const int ldi = (int)lda;
  const bool rows_full = I0 + 64 <= n;
  auto fetch = [&](int kt) {
    const int K0 = kt * 64;
    const double* vb = V + K0;
    if (rows_full && K0 + 64 <= n) {
      // interior tile: stored tile (k <= own rows; the diagonal tile whole -- its upper half is masked on the way into LDS)
      // or the mirrored one, both as they lie in memory
      const double* ab = K0 <= I0 ? A + (int64_t)K0 * lda + I0 : A + (int64_t)I0 * lda + K0;
#pragma unroll
      for (int i = 0; i < 16; ++i) rv[i] = vb[lo + (h0 + 4 * i) * n];
#pragma unroll
      for (int i = 0; i < 16; ++i) ra[i] = ab[lo + (h0 + 4 * i) * ldi];
      return;
    }
    const bool vin = K0 + lo < n;
#pragma unroll
    for (int i = 0; i < 16; ++i) rv[i] = vin ? vb[lo + (h0 + 4 * i) * n] : 0.0;                            // k = lo, col = hi
    if (K0 < I0) {
      const double* ab = A + (int64_t)K0 * lda + I0;
      const bool in = I0 + lo < n;
#pragma unroll
      for (int i = 0; i < 16; ++i) ra[i] = in ? ab[lo + (h0 + 4 * i) * ldi] : 0.0;                         // row = lo, k = hi
    } else if (K0 > I0) {
      const double* ab = A + (int64_t)I0 * lda + K0;
      const bool in = K0 + lo < n;
#pragma unroll
      for (int i = 0; i < 16; ++i) ra[i] = (in && I0 + h0 + 4 * i < n) ? ab[lo + (h0 + 4 * i) * ldi] : 0.0;   // k = lo, i = hi
    } else {
      const double* ab = A + (int64_t)I0 * lda + I0;
      const bool in = I0 + lo < n;
#pragma unroll
      for (int i = 0; i < 16; ++i) ra[i] = (in && lo >= h0 + 4 * i) ? ab[lo + (h0 + 4 * i) * ldi] : 0.0;   // lower half
    }
  };
  if (kt0 < kt1) fetch(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int K0 = kt * 64;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) Vt[(h0 + 4 * i) * SY_ST + lo] = rv[i];
    if (K0 < I0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) As[(h0 + 4 * i) * SY_SA + lo] = ra[i];
    } else if (K0 > I0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) As[(h0 + 4 * i) * SY_ST + lo] = ra[i];
    } else if (TRI) {
#pragma unroll
      for (int i = 0; i < 16; ++i) As[(h0 + 4 * i) * SY_SA + lo] = lo >= h0 + 4 * i ? ra[i] : 0.0;          // zero above the diagonal
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int hi = h0 + 4 * i;
        if (lo >= hi) { As[hi * SY_SA + lo] = ra[i]; As[lo * SY_SA + hi] = ra[i]; }
      }
    }
    __syncthreads();
    if (kt + 1 < kt1 && ABL != 2) fetch(kt + 1);
    if (ABL == 1) {
      acc[0][0] += As[tid] + Vt[tid];
    } else if (K0 <= I0) {
      const double* ap = As + lk * SY_SA + lr;                 // + 4 ks SY_SA + 16 rt
      const double* bp = Vt + (16 * w + lr) * SY_ST + lk;      // + 4 ks
      double a_cur[4], b_cur = bp[0];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) a_cur[rt] = ap[16 * rt];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        double a_nxt[4] = {0.0, 0.0, 0.0, 0.0}, b_nxt = 0.0;
        if (ks < 15) {
          b_nxt = bp[4 * (ks + 1)];
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) a_nxt[rt] = ap[4 * (ks + 1) * SY_SA + 16 * rt];
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[rt], b_cur, acc[rt], 0, 0, 0);
        b_cur = b_nxt;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) a_cur[rt] = a_nxt[rt];
      }
    } else {
      const double* ap = As + lr * SY_ST + lk;                 // + 16 rt SY_ST + 4 ks
      const double* bp = Vt + (16 * w + lr) * SY_ST + lk;
      double a_cur[4], b_cur = bp[0];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) a_cur[rt] = ap[16 * rt * SY_ST];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        double a_nxt[4] = {0.0, 0.0, 0.0, 0.0}, b_nxt = 0.0;
        if (ks < 15) {
          b_nxt = bp[4 * (ks + 1)];
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) a_nxt[rt] = ap[16 * rt * SY_ST + 4 * (ks + 1)];
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[rt], b_cur, acc[rt], 0, 0, 0);
        b_cur = b_nxt;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) a_cur[rt] = a_nxt[rt];
      }
    }
  }
  // D: lane l, register r = row 4 r + l / 16, column l % 16 of the 16 x 16 tile
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = I0 + 16 * rt + 4 * r + lk;
      if (row < n) W[row + (int64_t)(16 * w + lr) * n] = acc[rt][r];
    }
}

// W[e] = sum_s Wp[s][e] over the S contraction slices of sym_skinny_kernel, fixed order
__global__ void wsum_kernel(const double* __restrict__ Wp, int S, int64_t count, double* __restrict__ W) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  double s = Wp[e];
  for (int k = 1; k < S; ++k) s += Wp[(size_t)k * count + e];
  W[e] = s;
}

// dst[r + c ldd] = sign * sum_s Wp[s][r + c n]  (r < n, c < 64): the slice sum written into a panel of a larger matrix
__global__ void wsum_into_kernel(const double* __restrict__ Wp, int S, int64_t n, double* __restrict__ dst, int64_t ldd, double sign) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * 64) return;
  double s = Wp[e];
  for (int k = 1; k < S; ++k) s += Wp[(size_t)k * n * 64 + e];
  dst[(e % n) + (e / n) * ldd] = sign * s;
}

void launch_tall_product(hipStream_t st, const double* A, int64_t lda, int64_t n, const double* V, double* W_or_Wp, int S, bool tri) {
  const dim3 grid((unsigned)((n + 63) / 64), (unsigned)S);
  if (tri) hipLaunchKernelGGL(sym_skinny_kernel<true>, grid, dim3(256), 0, st, A, lda, (int)n, V, W_or_Wp);
  else hipLaunchKernelGGL(sym_skinny_kernel<false>, grid, dim3(256), 0, st, A, lda, (int)n, V, W_or_Wp);
}

void launch_slice_sum_into(hipStream_t st, const double* Wp, int S, int64_t n, double* dst, int64_t ldd, double sign) {
  hipLaunchKernelGGL(wsum_into_kernel, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, Wp, S, n, dst, ldd, sign);
}

// The last block column of the reduction, 2 <= n < 64 rows below the band: everything (the n x 64 panel, the n x n trailing
// matrix, the rotated columns of [X y]) fits one workgroup's LDS.  Unblocked Householder: reflector j from column j of the
// panel, applied to the rest of the panel, to both sides of the trailing matrix and to [X y].
__global__ __launch_bounds__(256) void band_tail_kernel(double* __restrict__ A, int64_t N, int64_t k0, double* __restrict__ Zr, int q1) {
  constexpr int LDT = 65;
  __shared__ double Ps[64 * LDT], As[64 * LDT], Zs[17 * 64], v[64], u[64], dc[64 + 17];
  __shared__ double sc[4];
  const int tid = threadIdx.x;
  const int64_t a0 = k0 + BAND_B;
  const int n = (int)(N - a0);
  for (int e = tid; e < 4096; e += 256) {
    const int r = e & 63, c = e >> 6;
    Ps[r * LDT + c] = r < n ? A[(a0 + r) + (k0 + c) * N] : 0.0;
    double a = 0.0;
    if (r < n && c < n) a = r >= c ? A[(a0 + r) + (a0 + c) * N] : A[(a0 + c) + (a0 + r) * N];
    As[r * LDT + c] = a;
  }
  for (int e = tid; e < 64 * q1; e += 256) {
    const int r = e & 63, c = e >> 6;
    Zs[c * 64 + r] = r < n ? Zr[(size_t)c * N + a0 + r] : 0.0;
  }
  __syncthreads();
  const int nref = n - 1 < 64 ? n - 1 : 64;
  for (int j = 0; j < nref; ++j) {
    if (tid < 64) {
      // |x[j+1:]|^2 by one wave
      double x = (tid > j && tid < n) ? Ps[tid * LDT + j] : 0.0;
      double s2 = wave_sum(x * x);
      if (tid == 0) {
        const double x0 = Ps[j * LDT + j];
        double tau = 0.0, beta = x0, scale = 0.0;
        if (fma(x0, x0, s2) > HH_TINY2) {                      // see panel_qr_step_kernel
          beta = -copysign(sqrt(fma(x0, x0, s2)), x0);
          tau = (beta - x0) / beta;
          scale = 1.0 / (x0 - beta);
        }
        sc[0] = tau; sc[1] = beta; sc[2] = scale;
      }
    }
    __syncthreads();
    const double tau = sc[0], beta = sc[1], scale = sc[2];
    if (tid < 64) v[tid] = tid < j || tid >= n ? 0.0 : (tid == j ? 1.0 : Ps[tid * LDT + j] * scale);
    __syncthreads();
    // dots of v with the remaining panel columns, with the columns of [X y], and u = A22 v
    if (tid < 64) {
      double d = 0.0;
      if (tid > j) for (int r = j; r < n; ++r) d = fma(v[r], Ps[r * LDT + tid], d);
      dc[tid] = d;
    } else if (tid < 64 + q1) {
      const int c = tid - 64;
      double d = 0.0;
      for (int r = j; r < n; ++r) d = fma(v[r], Zs[c * 64 + r], d);
      dc[tid] = d;
    } else if (tid >= 128 && tid < 192) {
      const int r = tid - 128;
      double d = 0.0;
      if (r < n) for (int c = j; c < n; ++c) d = fma(As[r * LDT + c], v[c], d);
      u[r] = d;
    }
    __syncthreads();
    if (tid < 64) {
      const double al = wave_sum(v[tid] * u[tid]);             // v'A22 v
      u[tid] = tau * fma(-0.5 * tau * al, v[tid], u[tid]);     // w
    }
    __syncthreads();
    for (int e = tid; e < 4096; e += 256) {
      const int r = e & 63, c = e >> 6;
      if (r < n) {
        if (c > j) Ps[r * LDT + c] = fma(-tau * v[r], dc[c], Ps[r * LDT + c]);
        if (c < n) As[r * LDT + c] -= v[r] * u[c] + u[r] * v[c];
      }
    }
    for (int e = tid; e < 64 * q1; e += 256) {
      const int r = e & 63, c = e >> 6;
      if (r < n) Zs[c * 64 + r] = fma(-tau * v[r], dc[64 + c], Zs[c * 64 + r]);
    }
    if (tid < 64 && tid < n) Ps[tid * LDT + j] = tid == j ? beta : (tid > j ? 0.0 : Ps[tid * LDT + j]);
    __syncthreads();
  }
  for (int e = tid; e < 4096; e += 256) {
    const int r = e & 63, c = e >> 6;
    if (r < n) {
      A[(a0 + r) + (k0 + c) * N] = Ps[r * LDT + c];
      if (c <= r) A[(a0 + r) + (a0 + c) * N] = As[r * LDT + c];
    }
  }
  for (int e = tid; e < 64 * q1; e += 256) {
    const int r = e & 63, c = e >> 6;
    if (r < n) Zr[(size_t)c * N + a0 + r] = Zs[c * 64 + r];
  }
}

// ---- the reduction ---------------------------------------------------------------------------------------------------
// Round 4: panels by Cholesky-QR (two passes) with the orthogonal factor in basis-kernel form, own products throughout --
// 16 launches per panel, none of them a library call, against ~85 (67 of them panel_qr_step_kernel) in band_reduce_hh
// below: at N = 5000 the reduction was 128 ms of launch latency around 10 ms of arithmetic.  Per panel P [n x 64] (the rows
// below the band), dense64.h:
//   G = P'P -> R1 = chol(G)', R1^-1 -> Q1 = P R1^-1 -> G2 = Q1'Q1 -> R2, R2^-1, Qtop = Q1[0:64] R2^-1, signs S;
//   H = I - V M V' with V = [I; 0] - Q S, M = (I - Qtop S)^-T (H orthogonal, H'P = [S R2 R1; 0]) -> W = A22 V ->
//   Y = W M - 1/2 V M'(V'W) M -> A22 -= V Y' + Y V'   and   [X y] -= V M'(V'[X y]).
// Cholesky-QR needs cond(P)^2 < 1/eps; a panel that is numerically rank deficient (duplicated individuals, an exactly
// low-rank K) raises a device flag -- no host round trip per panel -- and the caller redoes the whole reduction with
// Householder panels (band_reduce_hh), which has no such limit.
static int band_reduce_hh(mmg_ctx* ctx, mmg_reml* r);

static int band_reduce_cqr(mmg_ctx* ctx, mmg_reml* r, bool* suspect) {
  rocblas_handle h;
  int rc = reml_handle(ctx, &h);
  if (rc) return rc;
  if (dense64_init()) return set_err(ctx, MMG_E_HIP, "hipFuncSetAttribute (dense64 kernels)");
  const int64_t N = r->N;
  const int q1 = r->q + 1, b = BAND_B;
  hipStream_t st = ctx->stream;
  const auto t_start = std::chrono::steady_clock::now();
  if (!r->dBand) RC_HIP(ctx, hipMalloc(&r->dBand, (size_t)N * BAND_LD * sizeof(double)));
  if (!r->dZr) RC_HIP(ctx, hipMalloc(&r->dZr, (size_t)N * q1 * sizeof(double)));
  double* A = r->dL;
  r->linv_delta = NAN;                                         // the work copy of K takes dL over
  RC_HIP(ctx, hipMemcpyAsync(A, r->dK, (size_t)N * N * sizeof(double), hipMemcpyDeviceToDevice, st));
  RC_HIP(ctx, hipMemcpyAsync(r->dZr, r->dB, (size_t)N * q1 * sizeof(double), hipMemcpyDeviceToDevice, st));
  const int64_t nmax = std::max<int64_t>(N - b, 1);
  // contraction slices of W = A22 V so that a launch has >= ~256 workgroups
  const int smax = (int)std::max<int64_t>(1, std::min<int64_t>(16, (nmax + 63) / 64 / 3));
  Scratch sc;
  double *V = nullptr, *W = nullptr, *Wp = nullptr, *Y = nullptr, *small = nullptr, *part = nullptr, *part2 = nullptr;
  PanelFlags* flags = nullptr;
  RC_HIP(ctx, sc.alloc(&V, (size_t)2 * nmax * b * sizeof(double)));    // two panels' V: look-ahead (below)
  RC_HIP(ctx, sc.alloc(&W, (size_t)nmax * b * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&Y, (size_t)nmax * b * sizeof(double)));
  if (smax > 1) RC_HIP(ctx, sc.alloc(&Wp, (size_t)smax * nmax * b * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&small, (size_t)(6 * b * b + 17 * b) * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&part, (size_t)2 * D64_MAX_SLICES * 4096 * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&part2, (size_t)2 * 128 * 4096 * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&flags, sizeof(PanelFlags)));
  RC_HIP(ctx, hipMemsetAsync(flags, 0, sizeof(PanelFlags), st));
  double *R1 = small, *R1inv = small + b * b, *Mk = small + 2 * b * b, *Cb = small + 3 * b * b, *Cm = small + 4 * b * b, *Cz = small + 5 * b * b;
  double* partz = part + (size_t)D64_MAX_SLICES * 4096;
  const bool verbose = std::getenv("MMG_REML_VERBOSE") != nullptr;
  double tsec[4] = {0, 0, 0, 0};
  auto lap = [&](int which, std::chrono::steady_clock::time_point& tp) {
    if (!verbose) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    tsec[which] += std::chrono::duration<double>(now - tp).count();
    tp = now;
  };
  // (64 x 64 products A'B: Gram slices on the matrix pipe, summed by a second small launch -- a head kernel that added 40
  // slices itself spent 46 us reading 1.3 MB through one CU)
  // MMG_BAND_GRAPH=1 (A/B): the whole panel loop -- ~16 dependent launches per panel, 1200 at N = 5000, none of them a
  // library call -- captured into ONE hipGraph and replayed, instead of 1200 stream launches.  Measured at N = 5000
  // (tools/reml_time.py): the host enqueues a launch in ~4 us and the kernels average 23 us, so the stream never
  // waits for the host; what a panel costs is the dependent-kernel latency on the device, which a graph does not remove.
  static const bool use_graph = [] { const char* e = std::getenv("MMG_BAND_GRAPH"); return e && e[0] == '1'; }();
  const bool capture = use_graph && !verbose;
  const auto t_cap0 = std::chrono::steady_clock::now();
  if (capture) RC_HIP(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  // Look-ahead (round 5; MMG_BAND_LOOKAHEAD=0: the single-stream order of round 4).  A panel's factorisation is a chain of
  // two single-workgroup head kernels and four small launches (~200 us of which the chip works ~30) that needs only the FIRST
  // block column of the trailing matrix of the panel before.  So the two-sided update of panel k is split: its first block
  // column at once (launch_nt_update_col0, 77 tiles), then the rest on the compute stream while the factorisation of panel
  // k + 1 runs beside it on the context's second stream; W = A22 V of panel k + 1 waits for both.  V is double-buffered
  // (the rest of update k still reads V_k while the factorisation of k + 1 writes V_{k+1}); every other scratch buffer is
  // used by one side at a time (the factorisation starts after band_y of the panel before and ends before sym_skinny of its own).
  // Measured (tools/band_lookahead_ab.py, profiles/r5_band_lookahead_ab.txt): N = 12,000 118.7 -> 110.8 ms, but N = 5000 27.4 ->
  // 30.6 and N = 1000 4.4 -> 5.7: there the rest of an update is ~70 us against ~300 us of dependent panel work, and the two
  // event hops per panel cost more than the overlap returns.  On from N = 8192; MMG_BAND_LOOKAHEAD=1 / 0 force it.
  static const int la_env = [] { const char* e = std::getenv("MMG_BAND_LOOKAHEAD"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
  const bool lookahead = (la_env == 1 || (la_env < 0 && N >= 8192)) && !verbose && !capture;
  hipStream_t sf = lookahead ? ctx->stream2 : st;             // the stream of the panel factorisations
  hipEvent_t ev_fact[2] = {nullptr, nullptr}, ev_col0[2] = {nullptr, nullptr};
  struct EvGuard { hipEvent_t* e; int n; ~EvGuard() { for (int i = 0; i < n; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } g1{ev_fact, 2}, g2{ev_col0, 2};
  if (lookahead) {
    for (int i = 0; i < 2; ++i) {
      RC_HIP(ctx, hipEventCreateWithFlags(&ev_fact[i], hipEventDisableTiming));
      RC_HIP(ctx, hipEventCreateWithFlags(&ev_col0[i], hipEventDisableTiming));
    }
    RC_HIP(ctx, hipEventRecord(ev_col0[1], st));               // "panel -1": the copies of K and [X y] above
    RC_HIP(ctx, hipStreamWaitEvent(sf, ev_col0[1], 0));
  }
  // round-6 fusions, each switchable for A/B runs (tools/band_prof.py): MMG_BAND_FUSE=<bits>: 1 = Q1'Q1 slices inside rows_gemm,
  // 2 = the two reductions of (V'W, V'Z) in one launch, 4 = the next panel's P'P slices inside the rank-2b update's first block column
  static const int fuse_bits = [] { const char* e = std::getenv("MMG_BAND_FUSE"); return e ? std::atoi(e) : 7; }();
  const bool fuse_gram = (fuse_bits & 1) != 0, fuse_red2 = (fuse_bits & 2) != 0, fuse_g1 = (fuse_bits & 4) != 0;
  int g1_ready = 0;                                           // slices of the NEXT panel's P'P already in part2
  double* const Vbuf[2] = {V, V + (size_t)nmax * b};
  int64_t k0 = 0;
  int pk = 0;                                                 // panel counter (parity of the V buffer and the events)
  for (; N - k0 - b >= 2; k0 += b, ++pk) {
    auto tp = std::chrono::steady_clock::now();
    if (verbose) { (void)hipStreamSynchronize(st); tp = std::chrono::steady_clock::now(); }
    const int64_t a0 = k0 + b, n = N - a0;
    if (n < b) {                                              // the last, short block column: one workgroup
      if (lookahead) RC_HIP(ctx, hipStreamWaitEvent(st, ev_col0[(pk + 1) & 1], 0));   // (already ordered: same stream; kept for symmetry)
      hipLaunchKernelGGL(band_tail_kernel, dim3(1), dim3(256), 0, st, A, N, k0, r->dZr, q1);
      lap(0, tp);
      continue;
    }
    double* P = A + a0 + k0 * N;
    double* A22 = A + a0 + a0 * N;
    double* Zs = r->dZr + a0;
    double* V = Vbuf[pk & 1];
    // ---- panel: V, M, R  (stream sf; behind the first block column of the update before)
    if (lookahead && pk > 0) RC_HIP(ctx, hipStreamWaitEvent(sf, ev_col0[(pk - 1) & 1], 0));
    {
      // P'P: the slices the update of the panel before left behind (g1_ready), or a pass over P
      const int G1 = g1_ready > 0 ? g1_ready : launch_gram_slices(sf, P, N, P, N, b, n, part2, 128);
      g1_ready = 0;
      launch_gram_reduce(sf, part2, G1, part, 64);
      launch_cholqr_head1(sf, part, 1, R1, R1inv, flags);
      // Q1 = P R1^-1 and the slices of Q1'Q1 -- in one launch while a slice per 64 rows fits the buffer (n <= 8192)
      int G2;
      if (fuse_gram && (n + 63) / 64 <= 128) G2 = launch_rows_gemm_gram(sf, P, N, V, n, n, R1inv, part2);
      else {
        launch_rows_gemm(sf, P, N, V, n, n, R1inv);
        G2 = launch_gram_slices(sf, V, n, V, n, b, n, part2, 128);
      }
      launch_gram_reduce(sf, part2, G2, part, 64);
      launch_cholqr_head2(sf, part, 1, R1, V, n, Mk, Cb, P, N, flags);
      launch_rows_gemm(sf, V + b, n, V + b, n, n - b, Cb);    // V[64:] = Q1[64:] (-R2^-1 S)
    }
    if (lookahead) {
      RC_HIP(ctx, hipEventRecord(ev_fact[pk & 1], sf));
      RC_HIP(ctx, hipStreamWaitEvent(st, ev_fact[pk & 1], 0));
    }
    lap(0, tp);
    // ---- W = A22 V
    // ~3 tile steps per workgroup while the launch stays within ~1024 workgroups
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)smax, 1024 / ((n + 63) / 64), (n + 63) / 64 / 3}));
#ifdef MMG_EXPERIMENTS
    static const int sym_abl = [] { const char* e = std::getenv("MMG_SYM_ABL"); return e ? std::atoi(e) : 0; }();
    if (sym_abl == 1) hipLaunchKernelGGL((sym_skinny_kernel<false, 1>), dim3((unsigned)((n + 63) / 64), S), dim3(256), 0, st, A22, N, (int)n, V, S > 1 ? Wp : W);
    else if (sym_abl == 2) hipLaunchKernelGGL((sym_skinny_kernel<false, 2>), dim3((unsigned)((n + 63) / 64), S), dim3(256), 0, st, A22, N, (int)n, V, S > 1 ? Wp : W);
    else
#endif
    hipLaunchKernelGGL(sym_skinny_kernel<false>, dim3((unsigned)((n + 63) / 64), S), dim3(256), 0, st, A22, N, (int)n, V, S > 1 ? Wp : W);
    if (S > 1) hipLaunchKernelGGL(wsum_kernel, dim3((unsigned)((n * b + 255) / 256)), dim3(256), 0, st, Wp, S, n * b, W);
    lap(1, tp);
    // ---- Y = W M - 1/2 V M'(V'W) M and the rotated columns of [X y]
    {
      const int G = launch_gram_slices2(st, V, n, n, W, n, b, part2, Zs, N, q1, part2 + (size_t)128 * 4096, 128);
      if (fuse_red2) launch_gram_reduce2(st, part2, part2 + (size_t)128 * 4096, G, part, partz);
      else {
        launch_gram_reduce(st, part2, G, part, 64);
        launch_gram_reduce(st, part2 + (size_t)128 * 4096, G, partz, 64);
      }
    }
    launch_band_coef(st, part, 1, partz, 1, q1, Mk, Cm, Cz);
    launch_band_y(st, V, W, n, Mk, Cm, Y, Zs, N, Cz, q1);
    lap(2, tp);
    // ---- A22 -= V Y' + Y V' (lower tiles): the first block column, then the rest
    if (lookahead) {
      launch_nt_update_col0(st, A22, N, n, V, Y, Y, V, n, n);
      RC_HIP(ctx, hipEventRecord(ev_col0[pk & 1], st));
      if (n > b) launch_nt_update_lower(st, A22 + b + b * N, N, n - b, V + b, Y + b, Y + b, V + b, n, n);
    } else {
      // the tiles of the first block column also write the next panel's Gram slices (part2 is free: its last readers, the
      // reductions in front of band_coef, are behind us on this stream) -- when there is a next full panel
      const bool next_full = n - b >= b && N - (k0 + b) - b >= 2;
      g1_ready = launch_nt_update_lower(st, A22, N, n, V, Y, Y, V, n, n, (fuse_g1 && next_full) ? part2 : nullptr);
    }
    lap(3, tp);
  }
  if (lookahead) {                                            // nothing of this call may be left on the second stream
    RC_HIP(ctx, hipEventRecord(ev_fact[0], sf));
    RC_HIP(ctx, hipStreamWaitEvent(st, ev_fact[0], 0));
  }
  if (capture) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    RC_HIP(ctx, hipStreamEndCapture(st, &graph));
    const auto t_cap1 = std::chrono::steady_clock::now();
    RC_HIP(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const auto t_cap2 = std::chrono::steady_clock::now();
    RC_HIP(ctx, hipGraphLaunch(exec, st));
    RC_HIP(ctx, hipStreamSynchronize(st));
    const auto t_cap3 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) { return std::chrono::duration<double>(b2 - a).count() * 1e3; };
    if (std::getenv("MMG_BAND_GRAPH_VERBOSE"))
      fprintf(stderr, "[reml] N=%lld: band reduction as a hipGraph: capture %.2f ms, instantiate %.2f ms, launch + run %.2f ms\n", (long long)N,
              ms(t_cap0, t_cap1), ms(t_cap1, t_cap2), ms(t_cap2, t_cap3));
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
  }
  RC_HIP(ctx, hipGetLastError());
  PanelFlags hf;
  RC_HIP(ctx, hipMemcpyAsync(&hf, flags, sizeof(hf), hipMemcpyDeviceToHost, st));
  RC_HIP(ctx, hipStreamSynchronize(st));
  if (verbose)
    fprintf(stderr, "[reml] N=%lld: band reduction (Cholesky-QR panels, %d of them, second factor by series %d, first order %d%s) %.3f s: panel %.3f, A22 V %.3f, coefficients + Y %.3f, rank-2b update %.3f\n",
            (long long)N, hf.panels, hf.series, hf.tiny, hf.bad ? "; a panel was rank deficient -> Householder panels" : "",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), tsec[0], tsec[1], tsec[2], tsec[3]);
  *suspect = hf.bad != 0;
  r->band_k0 = k0;                                            // band_reduce_hh finishes the short panels from here
  return MMG_OK;
}

static int band_reduce(mmg_ctx* ctx, mmg_reml* r) {
  const char* e = std::getenv("MMG_BAND_IMPL");
  const std::string impl = e ? e : "";
  const auto t_start = std::chrono::steady_clock::now();
  r->band_k0 = 0;
  if (impl.empty() || impl == "cqr") {
    bool suspect = false;
    int rc = band_reduce_cqr(ctx, r, &suspect);
    if (rc) return rc;
    if (suspect) r->band_k0 = 0;                              // start over: dL is re-copied from K
    r->band_fallback = suspect;
  }
  const int rc = band_reduce_hh(ctx, r);
  r->band_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  return rc;
}

static int band_reduce_hh(mmg_ctx* ctx, mmg_reml* r) {
  rocblas_handle h;
  int rc = reml_handle(ctx, &h);
  if (rc) return rc;
  const int64_t N = r->N;
  const int q1 = r->q + 1, b = BAND_B;
  hipStream_t st = ctx->stream;
  const auto t_start = std::chrono::steady_clock::now();
  if (!r->dBand) RC_HIP(ctx, hipMalloc(&r->dBand, (size_t)N * BAND_LD * sizeof(double)));
  if (!r->dZr) RC_HIP(ctx, hipMalloc(&r->dZr, (size_t)N * q1 * sizeof(double)));
  double* A = r->dL;                                          // work copy of K, reduced in place (lower triangle)
  r->linv_delta = NAN;                                        // ... so whatever L^-1 it cached is gone (advisor r5)
  const int64_t k_first = r->band_k0;                         // > 0: band_reduce_cqr did the panels before this one
  if (k_first == 0) {
    RC_HIP(ctx, hipMemcpyAsync(A, r->dK, (size_t)N * N * sizeof(double), hipMemcpyDeviceToDevice, st));
    RC_HIP(ctx, hipMemcpyAsync(r->dZr, r->dB, (size_t)N * q1 * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  Scratch sc;
  double *VW = nullptr, *YV = nullptr, *T = nullptr, *tau = nullptr, *M1 = nullptr, *M2 = nullptr, *small = nullptr, *part = nullptr;
  const int64_t nmax = std::max<int64_t>(N - b - k_first, 1);
  const int GQ = (int)((nmax + QR_ROWS - 1) / QR_ROWS);       // workgroups of a panel-QR step
  const int GT = 128;                                         // row slices of a tall-skinny product
  RC_HIP(ctx, sc.alloc(&VW, (size_t)nmax * 2 * b * sizeof(double)));      // [V | A22 V] -> [V | Y]
  RC_HIP(ctx, sc.alloc(&YV, (size_t)nmax * 2 * b * sizeof(double)));      // [Y | V]
  RC_HIP(ctx, sc.alloc(&T, (size_t)b * b * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&tau, (size_t)2 * b * sizeof(double)));            // tau, diagonal of R
  RC_HIP(ctx, sc.alloc(&M1, (size_t)b * b * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&M2, (size_t)b * b * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&small, (size_t)(2 * b * b + 4 * b) * sizeof(double)));   // Cm [128 x 64], two pivot rows
  RC_HIP(ctx, sc.alloc(&part, (size_t)std::max<int64_t>((int64_t)GT * 4096, 2 * (int64_t)GQ * 64) * sizeof(double)));
  double* V = VW;
  double* W = VW + (size_t)nmax * b;                          // second half at its largest; per panel: VW + n * b
  double* Cm = small;
  double* pivrow = small + 2 * b * b;
  const double one = 1.0, zero = 0.0, mone = -1.0, mhalf = -0.5;
  // MMG_BAND_IMPL=lib: every panel through rocSOLVER / rocBLAS calls alone (geqrf, larft, symm, trmm, syr2k) -- the
  // first version, kept for A/B runs and as the path of panels shorter than 256 rows
  const bool lib_only = [] { const char* e = std::getenv("MMG_BAND_IMPL"); return e && std::string(e) == "lib"; }();
  const int64_t bs = [] { const char* e = std::getenv("MMG_BAND_BS"); return e ? std::max<int64_t>(256, std::atoll(e) / 64 * 64) : int64_t(4096); }();
  const bool verbose = std::getenv("MMG_REML_VERBOSE") != nullptr;
  double tsec[4] = {0, 0, 0, 0};                              // panel QR + T, A22 V, small products, rank-2b update
  auto lap = [&](int which, std::chrono::steady_clock::time_point& tp) {
    if (!verbose) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    tsec[which] += std::chrono::duration<double>(now - tp).count();
    tp = now;
  };
  auto tsmm = [&](const double* Am, int64_t lda, const double* Bm, int64_t ldb, int kb, int64_t n, double* out, int ldo) {
    const int rows_per = (int)((((n + GT - 1) / GT) + TS_ROWS - 1) / TS_ROWS * TS_ROWS);
    const int G = (int)((n + rows_per - 1) / rows_per);
    hipLaunchKernelGGL(tsmm_tn_kernel, dim3(G), dim3(256), 0, st, Am, lda, Bm, ldb, kb, (int)n, rows_per, part);
    hipLaunchKernelGGL(tsmm_reduce_kernel, dim3(16), dim3(256), 0, st, part, G, out, ldo);
  };
  const bool dbg = std::getenv("MMG_BAND_DEBUG") != nullptr;  // column norms of the rotated [X y] panel by panel (orthogonality)
  auto dbg_norms = [&](long long at) {
    if (!dbg) return;
    std::vector<double> hz((size_t)N * q1);
    (void)hipMemcpyAsync(hz.data(), r->dZr, hz.size() * sizeof(double), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "[band dbg] before panel at column %lld: |Z_c|^2 =", at);
    for (int c = 0; c < q1; ++c) { double t = 0; for (int64_t i = 0; i < N; ++i) t += hz[(size_t)c * N + i] * hz[(size_t)c * N + i]; fprintf(stderr, " %.15g", t); }
    fprintf(stderr, "\n");
  };
  for (int64_t k0 = k_first; N - k0 - b >= 2; k0 += b) {
    dbg_norms((long long)k0);
    auto tp = std::chrono::steady_clock::now();
    if (verbose) { (void)hipStreamSynchronize(st); tp = std::chrono::steady_clock::now(); }
    const int64_t a0 = k0 + b;                                // first row / column of the trailing matrix
    const int64_t n = N - a0;                                 // rows below the band in this block column
    const int nr = (int)std::min<int64_t>(n, b);              // reflectors
    double* P = A + a0 + k0 * N;                              // [n x b] panel
    double* A22 = A + a0 + a0 * N;                            // [n x n] trailing matrix
    double* Zs = r->dZr + a0;
    W = VW + (size_t)n * b;
    if (!lib_only && n < b) {                                 // the last block column: one workgroup (as in band_reduce_cqr)
      hipLaunchKernelGGL(band_tail_kernel, dim3(1), dim3(256), 0, st, A, N, k0, r->dZr, q1);
      lap(0, tp);
      continue;
    }
    // (round 4: panels shorter than 256 rows took the library path too; rocSOLVER's geqrf loses orthogonality on the noise
    // columns of a rank-deficient panel the way the own kernel did before HH_TINY2 -- 1e-6 on the REML sums -- so they are
    // the own kernels' now)
    if (lib_only) {
      RC_RB(ctx, rocsolver_dgeqrf(h, (rocblas_int)n, b, P, (rocblas_int)N, tau));
      hipLaunchKernelGGL(band_build_v_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nr), dim3(256), 0, st, P, N, (int)n, nr, V);
      RC_RB(ctx, rocsolver_dlarft(h, rocblas_forward_direction, rocblas_column_wise, (rocblas_int)n, nr, V, (rocblas_int)n, tau, T, b));
      lap(0, tp);
      // X = A22 V T;  Y = X - 1/2 V (T'V'X);  A22 <- A22 - Y V' - V Y'   ( = Q' A22 Q,  Q = I - V T V' )
      RC_RB(ctx, rocblas_dsymm_64(h, rocblas_side_left, rocblas_fill_lower, n, nr, &one, A22, N, V, n, &zero, W, n));
      lap(1, tp);
      RC_RB(ctx, rocblas_dtrmm_64(h, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, n, nr,
                                  &one, T, b, W, n, W, n));
      RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_transpose, rocblas_operation_none, nr, nr, n, &one, V, n, W, n, &zero, M1, b));
      RC_RB(ctx, rocblas_dtrmm_64(h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, nr,
                                  nr, &one, T, b, M1, b, M1, b));
      RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, n, nr, nr, &mhalf, V, n, M1, b, &one, W, n));
      lap(2, tp);
      RC_RB(ctx, rocblas_dsyr2k_64(h, rocblas_fill_lower, rocblas_operation_none, n, nr, &mone, W, n, V, n, &one, A22, N));
      lap(3, tp);
      // the rotated columns of [X y]:  Z[a0:] <- Q' Z[a0:] = Z - V T' (V'Z)
      RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_transpose, rocblas_operation_none, nr, q1, n, &one, V, n, Zs, N, &zero, M2, b));
      RC_RB(ctx, rocblas_dtrmm_64(h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, nr,
                                  q1, &one, T, b, M2, b, M2, b));
      RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, n, q1, nr, &mone, V, n, M2, b, &one, Zs, N));
      lap(2, tp);
      continue;
    }
    // ---- panel QR: V, tau, R
    {
      const int G = (int)((n + QR_ROWS - 1) / QR_ROWS);
      for (int j = -1; j < b; ++j) {
        const int in = (j + 2) & 1, out = (j + 1) & 1;        // parity of the slice sums / pivot row a launch reads / writes
        hipLaunchKernelGGL(panel_qr_step_kernel, dim3(G), dim3(256), 0, st, P, N, (int)n, j, part + (size_t)in * GQ * 64,
                           part + (size_t)out * GQ * 64, pivrow + in * b, pivrow + out * b, G, V, tau, tau + b);
      }
      hipLaunchKernelGGL(panel_qr_finish_kernel, dim3(1), dim3(b), 0, st, P, N, tau + b);
      tsmm(V, n, V, n, b, n, M1, b);                          // S = V'V
      hipLaunchKernelGGL(form_t_kernel, dim3(1), dim3(64), 0, st, M1, tau, T);
    }
    lap(0, tp);
    if (dbg && k0 == k_first) {                               // is (V, T) an orthogonal Q = I - V T V'?  D = T + T' - T' (V'V) T must vanish
      std::vector<double> hv((size_t)n * b), ht((size_t)b * b), htau(b);
      (void)hipMemcpyAsync(hv.data(), V, hv.size() * sizeof(double), hipMemcpyDeviceToHost, st);
      (void)hipMemcpyAsync(ht.data(), T, ht.size() * sizeof(double), hipMemcpyDeviceToHost, st);
      (void)hipMemcpyAsync(htau.data(), tau, b * sizeof(double), hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
      std::vector<double> g((size_t)b * b, 0.0);
      for (int a = 0; a < b; ++a) for (int c = 0; c < b; ++c) { double t = 0; for (int64_t i = 0; i < n; ++i) t += hv[i + (size_t)a * n] * hv[i + (size_t)c * n]; g[a + (size_t)c * b] = t; }
      double worst_tau = 0, vmax = 0;
      for (int a = 0; a < b; ++a) worst_tau = std::max(worst_tau, std::fabs(htau[a] * g[a + (size_t)a * b] - 2.0) * (htau[a] != 0.0));
      for (double x : hv) vmax = std::max(vmax, std::fabs(x));
      // D = T + T' - T' G T
      std::vector<double> gt((size_t)b * b, 0.0);
      for (int a = 0; a < b; ++a) for (int c = 0; c < b; ++c) { double t = 0; for (int k = 0; k < b; ++k) t += g[a + (size_t)k * b] * ht[k + (size_t)c * b]; gt[a + (size_t)c * b] = t; }
      double dmax = 0; int da = 0, dc = 0;
      for (int a = 0; a < b; ++a) for (int c = 0; c < b; ++c) {
        double t = ht[a + (size_t)c * b] + ht[c + (size_t)a * b];
        for (int k = 0; k < b; ++k) t -= ht[k + (size_t)a * b] * gt[k + (size_t)c * b];
        if (std::fabs(t) > dmax) { dmax = std::fabs(t); da = a; dc = c; }
      }
      fprintf(stderr, "[band dbg] first panel: max |V| %.3g, max |tau_j |v_j|^2 - 2| %.3g, max |T + T' - T'(V'V)T| %.3g at (%d, %d); tau[0..7] =", vmax, worst_tau, dmax, da, dc);
      for (int a = 0; a < 8; ++a) fprintf(stderr, " %.6g", htau[a]);
      fprintf(stderr, "\n[band dbg] j: tau_j |v_j|^2 tau|v|^2:");
      for (int a = 0; a < b; ++a) if (std::fabs(htau[a] * g[a + (size_t)a * b] - 2.0) > 1e-9) fprintf(stderr, "  %d: %.6g %.6g %.6g;", a, htau[a], g[a + (size_t)a * b], htau[a] * g[a + (size_t)a * b]);
      fprintf(stderr, "\n");
    }
    // ---- W = A22 V from the lower triangle alone
    hipLaunchKernelGGL(sym_skinny_kernel<false>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, A22, N, (int)n, V, W);
    lap(1, tp);
    // ---- Y = [V | W] [ -1/2 T'(V'W)T ; T ]
    tsmm(V, n, W, n, b, n, M1, b);                            // V'W
    RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, b, b, b, &one, M1, b, T, b, &zero, M2, b));
    RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_transpose, rocblas_operation_none, b, b, b, &one, T, b, M2, b, &zero, M1, b));
    hipLaunchKernelGGL(form_coef_kernel, dim3(16), dim3(256), 0, st, M1, T, Cm);
    double* Y = YV;
    RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, n, b, 2 * b, &one, VW, n, Cm, 2 * b, &zero, Y, n));
    RC_HIP(ctx, hipMemcpyAsync(YV + (size_t)n * b, V, (size_t)n * b * sizeof(double), hipMemcpyDeviceToDevice, st));   // [Y | V]
    RC_HIP(ctx, hipMemcpyAsync(W, Y, (size_t)n * b * sizeof(double), hipMemcpyDeviceToDevice, st));                     // [V | Y]
    // the rotated columns of [X y]
    tsmm(V, n, Zs, N, q1, n, M2, b);
    RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_transpose, rocblas_operation_none, b, q1, b, &one, T, b, M2, b, &zero, M1, b));
    RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, n, q1, b, &mone, V, n, M1, b, &one, Zs, N));
    lap(2, tp);
    // ---- A22 <- A22 - [V | Y] [Y | V]' on the blocks on and below the diagonal of the absolute grid
    for (int64_t rb = a0 / bs * bs; rb < N; rb += bs) {
      const int64_t r_lo = std::max(rb, a0), r_hi = std::min(N, rb + bs), m = r_hi - r_lo;
      RC_RB(ctx, rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_transpose, m, r_hi - a0, 2 * b, &mone, VW + (r_lo - a0), n,
                                  YV, n, &one, A + r_lo + a0 * N, N));
    }
    lap(3, tp);
  }
  dbg_norms(-1);
  hipLaunchKernelGGL(band_extract_kernel, dim3((unsigned)N), dim3(64), 0, st, A, N, r->dBand);
  RC_HIP(ctx, hipGetLastError());
  RC_HIP(ctx, hipStreamSynchronize(st));
  r->band_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  if (verbose)
    fprintf(stderr, "[reml] N=%lld: band reduction (b = %d, %s, panels from column %lld) %.3f s: panel QR + T %.3f, A22 V %.3f, small products %.3f, rank-2b update %.3f\n",
            (long long)N, b, lib_only ? "library calls" : "Householder panels, block-lower products", (long long)k_first, r->band_s, tsec[0],
            tsec[1], tsec[2], tsec[3]);
  r->band_ready = true;
  return MMG_OK;
}

void reml_band_free(mmg_reml* r) {
  hipFree(r->dBand); hipFree(r->dZr); hipFree(r->dBandWs);
  r->dBand = r->dZr = nullptr;
  r->dBandWs = nullptr; r->band_ws_bytes = 0;
  r->band_ready = false;
}

int reml_band_sums(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s2, double* s3, double* s4,
                   double* ldh, double* trh) {
  if (!r->band_ready) {
    int rc = band_reduce(ctx, r);
    if (rc) return rc;
  }
  const int N = r->N, q = r->q, q1 = q + 1;
  hipStream_t st = ctx->stream;
  const bool verbose = std::getenv("MMG_REML_VERBOSE") != nullptr;
  // deltas in groups: L of a group is group x N x 72 doubles
  const int64_t per = (int64_t)N * BAND_LD * sizeof(double);
  // (256 since round 5, as many as mmg_reml_band_factor keeps: with 128 a call of 129-256 values was two chains back to back,
  // tools/band_chain_width.py: 8.4 -> 16.3 ms at N = 5000, now 9.7)
  const int group = (int)std::max<int64_t>(1, std::min<int64_t>(256, (int64_t(8) << 30) / per));
  for (int g0 = 0; g0 < nd; g0 += group) {
    const int ng = std::min(group, nd - g0);
    double *dd = nullptr, *L = nullptr, *F = nullptr, *G = nullptr, *sca = nullptr;
    int* dfail = nullptr;
    const size_t nsc = (size_t)ng * (2 + 2 * q1 * q1);
    // factors kept by mmg_reml_band_factor: when EVERY variance ratio of the group is among them (same bits) the factor sweep is
    // skipped and the chains below read the kept factors through an index list
    std::vector<int> sel;
    if (ctx->band_keep_owner == r && !r->keep_deltas.empty()) {
      sel.resize(ng);
      for (int k = 0; k < ng && !sel.empty(); ++k) {
        const auto it = std::find(r->keep_deltas.begin(), r->keep_deltas.end(), deltas[g0 + k]);
        if (it == r->keep_deltas.end()) sel.clear();
        else sel[k] = (int)(it - r->keep_deltas.begin());
      }
    }
    const bool kept = !sel.empty();
    {
      // one buffer kept with the workspace, carved at 256-byte boundaries
      auto up = [](size_t b) { return (b + 255) / 256 * 256; };
      const size_t b_dd = up(ng * sizeof(double)), b_L = kept ? 0 : up((size_t)ng * per), b_F = up((size_t)ng * q1 * N * sizeof(double)),
                   b_sca = up(nsc * sizeof(double)), b_fail = up(ng * sizeof(int));
      const size_t need = b_dd + b_L + 2 * b_F + b_sca + b_fail;
      // up to 2 GB the buffer belongs to the CONTEXT and outlives this workspace (a workspace lives for one emmax() call; entry
      // points of a context do not overlap, and this one returns with its streams idle); larger ones stay with the workspace
      const bool shared = need <= ((size_t)2 << 30);
      void*& ws = shared ? ctx->band_ws : r->dBandWs;
      size_t& cap = shared ? ctx->band_ws_cap : r->band_ws_bytes;
      if (cap < need) {
        RC_HIP(ctx, hipStreamSynchronize(st));
        (void)hipFree(ws);
        ws = nullptr; cap = 0;
        RC_HIP(ctx, hipMalloc(&ws, need));
        cap = need;
      }
      char* w = (char*)ws;
      dd = (double*)w; w += b_dd;
      L = (double*)w; w += b_L;
      F = (double*)w; w += b_F;
      G = (double*)w; w += b_F;
      sca = (double*)w; w += b_sca;
      dfail = (int*)w;
    }
    double* dlog = sca;
    double* dtr = sca + ng;
    double* dff = sca + 2 * ng;
    double* dgg = dff + (size_t)ng * q1 * q1;
    RC_HIP(ctx, hipMemsetAsync(sca, 0, nsc * sizeof(double), st));
    const auto t0 = std::chrono::steady_clock::now();
    const int* dsel = nullptr;
    if (kept) {
      L = (double*)ctx->band_keep;
      RC_HIP(ctx, hipMemcpyAsync(dfail, sel.data(), ng * sizeof(int), hipMemcpyHostToDevice, st));    // (the failure flags' place)
      dsel = dfail;
    } else {
      RC_HIP(ctx, hipMemcpyAsync(dd, deltas + g0, ng * sizeof(double), hipMemcpyHostToDevice, st));
      if (launch_band_factor(st, ng, r->dBand, N, dd, L, dlog, dfail)) return set_err(ctx, MMG_E_HIP, "hipFuncSetAttribute (band_factor_blk_kernel)");
      std::vector<int> bad(ng);
      RC_HIP(ctx, hipMemcpyAsync(bad.data(), dfail, ng * sizeof(int), hipMemcpyDeviceToHost, st));
      RC_HIP(ctx, hipStreamSynchronize(st));
      for (int k = 0; k < ng; ++k)
        if (bad[k])
          return set_err(ctx, MMG_E_LIB, "K + delta I is not positive definite (banded Cholesky, pivot " + std::to_string(bad[k]) + ")");
    }
    const auto t1 = std::chrono::steady_clock::now();
    // the trace recurrence and the substitutions both read L and nothing of each other: two single-workgroup chains per
    // delta, 3.5 and 1.8 ms at N = 5000 -- side by side on the context's two streams (round 5: 9.2 -> 7.4 ms per call)
    static const bool side = [] { const char* e = std::getenv("MMG_BAND_TRACE_SIDE"); return !(e && e[0] == '0'); }();
    if (side) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      RC_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
      RC_HIP(ctx, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
      (void)hipEventRecord(e0, st);
      (void)hipStreamWaitEvent(ctx->stream2, e0, 0);
      if (launch_band_trace(ctx->stream2, ng, L, N, dtr, dsel)) return set_err(ctx, MMG_E_HIP, "hipFuncSetAttribute (band_trace_blk_kernel)");
      (void)hipEventRecord(e1, ctx->stream2);
      hipLaunchKernelGGL(band_solve_kernel, dim3(ng), dim3(256), 0, st, L, N, r->dZr, q1, F, G, dsel);
      (void)hipStreamWaitEvent(st, e1, 0);
      (void)hipEventDestroy(e0);                               // (released when the work that references them has completed)
      (void)hipEventDestroy(e1);
    } else {
      hipLaunchKernelGGL(band_solve_kernel, dim3(ng), dim3(256), 0, st, L, N, r->dZr, q1, F, G, dsel);
      if (launch_band_trace(st, ng, L, N, dtr, dsel)) return set_err(ctx, MMG_E_HIP, "hipFuncSetAttribute (band_trace_blk_kernel)");
    }
    hipLaunchKernelGGL(band_gram_kernel, dim3(ng, q1 * q1), dim3(256), 0, st, F, G, N, q1, dff, dgg);
    RC_HIP(ctx, hipGetLastError());
    std::vector<double> hs(nsc);
    RC_HIP(ctx, hipMemcpyAsync(hs.data(), sca, nsc * sizeof(double), hipMemcpyDeviceToHost, st));
    RC_HIP(ctx, hipStreamSynchronize(st));
    if (kept)
      for (int k = 0; k < ng; ++k) hs[k] = r->keep_logdet[sel[k]];            // log|B + delta I| came with the kept factor
    if (verbose) {
      const auto t2 = std::chrono::steady_clock::now();
      fprintf(stderr, "[reml] N=%d: %d deltas through the band: factor %.1f ms%s, solves + trace %.1f ms\n", N, ng,
              std::chrono::duration<double>(t1 - t0).count() * 1e3, kept ? " (kept factors)" : "", std::chrono::duration<double>(t2 - t1).count() * 1e3);
    }
    for (int k = 0; k < ng; ++k) {
      const double* ff = hs.data() + 2 * ng + (size_t)k * q1 * q1;
      const double* gg = hs.data() + 2 * ng + (size_t)ng * q1 * q1 + (size_t)k * q1 * q1;
      std::vector<double> a((size_t)q * q), bvec((size_t)q), B2((size_t)q * q);
      for (int i = 0; i < q; ++i) {
        for (int j = 0; j < q; ++j) { a[i * q + j] = ff[i * q1 + j]; B2[i * q + j] = gg[i * q1 + j]; }
        bvec[i] = ff[i * q1 + q];
      }
      const double c = ff[q * q1 + q];
      std::vector<double> beta = bvec;
      double logdet_a = 0.0;
      if (!chol_solve_small(q, a, beta, 1, &logdet_a)) return set_err(ctx, MMG_E_LIB, "X'H^-1 X is not positive definite");
      std::vector<double> aB2 = B2;
      chol_solve_small(q, a, aB2, q, nullptr);
      double tr_aB2 = 0.0, bb = 0.0, v3 = gg[q * q1 + q];
      for (int i = 0; i < q; ++i) {
        tr_aB2 += aB2[i * q + i];
        bb += bvec[i] * beta[i];
        v3 -= 2.0 * beta[i] * gg[i * q1 + q];
        for (int j = 0; j < q; ++j) v3 += beta[i] * beta[j] * gg[i * q1 + j];
      }
      if (ldh) ldh[g0 + k] = hs[k];
      if (trh) trh[g0 + k] = hs[ng + k];
      s1[g0 + k] = c - bb;
      s2[g0 + k] = hs[k] + logdet_a - r->logdet_xtx;
      s3[g0 + k] = v3;
      s4[g0 + k] = hs[ng + k] - tr_aB2;
    }
  }
  return MMG_OK;
}

// mmg_reml_band_factor: the banded Cholesky factors of B + delta I for nd variance ratios, kept in the context (<= 256 of them and
// <= 2 GB; more: nothing is kept and nothing fails) -- a later reml_band_sums whose variance ratios are ALL among them skips
// its factor sweep.  One workgroup per variance ratio: the sweep costs for 227 what it costs for 51 (4.0 ms at N = 5000).
int reml_band_factor_keep(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas) {
  r->keep_deltas.clear(); r->keep_logdet.clear();
  if (ctx->band_keep_owner == r) ctx->band_keep_owner = nullptr;
  if (!r->band_ready) {
    int rc = band_reduce(ctx, r);
    if (rc) return rc;
  }
  const int N = r->N;
  const size_t per = (size_t)N * BAND_LD * sizeof(double);
  if (nd <= 0 || nd > 256 || (size_t)nd * per > ((size_t)2 << 30)) return MMG_OK;
  hipStream_t st = ctx->stream;
  if (ctx->band_keep_cap < (size_t)nd * per) {
    RC_HIP(ctx, hipStreamSynchronize(st));
    (void)hipFree(ctx->band_keep);
    ctx->band_keep = nullptr; ctx->band_keep_cap = 0;
    RC_HIP(ctx, hipMalloc(&ctx->band_keep, (size_t)nd * per));
    ctx->band_keep_cap = (size_t)nd * per;
  }
  ctx->band_keep_owner = nullptr;                            // (whatever another workspace kept there is gone)
  Scratch sc;
  double *dd = nullptr, *dlog = nullptr;
  int* dfail = nullptr;
  RC_HIP(ctx, sc.alloc(&dd, nd * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&dlog, nd * sizeof(double)));
  RC_HIP(ctx, sc.alloc(&dfail, nd * sizeof(int)));
  RC_HIP(ctx, hipMemcpyAsync(dd, deltas, nd * sizeof(double), hipMemcpyHostToDevice, st));
  if (launch_band_factor(st, nd, r->dBand, N, dd, (double*)ctx->band_keep, dlog, dfail)) return set_err(ctx, MMG_E_HIP, "hipFuncSetAttribute (band_factor_blk_kernel)");
  RC_HIP(ctx, hipGetLastError());
  std::vector<int> bad(nd);
  std::vector<double> ld(nd);
  RC_HIP(ctx, hipMemcpyAsync(bad.data(), dfail, nd * sizeof(int), hipMemcpyDeviceToHost, st));
  RC_HIP(ctx, hipMemcpyAsync(ld.data(), dlog, nd * sizeof(double), hipMemcpyDeviceToHost, st));
  RC_HIP(ctx, hipStreamSynchronize(st));
  for (int k = 0; k < nd; ++k)
    if (bad[k])
      return set_err(ctx, MMG_E_LIB, "K + delta I is not positive definite (banded Cholesky, pivot " + std::to_string(bad[k]) + ")");
  r->keep_deltas.assign(deltas, deltas + nd);
  r->keep_logdet = ld;
  ctx->band_keep_owner = r;
  return MMG_OK;
}

}  // namespace mmg
