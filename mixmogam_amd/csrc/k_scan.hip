// k_scan.hip -- the EMMAX per-SNP scan (replaces the chunked GEMM + per-SNP lstsq loop of
// linear_models.py:1316-1349 with its closed form):
//     num_m = (s_m . w)^2,  den_m = s_m' A s_m,  rss_m = h0_rss - num_m/den_m,
//     F_m = (h0_rss/rss_m - 1) * df2,  p_m = f.sf(F_m, 1, df2).
//
// den is the hot part (2 N^2 flop per SNP).  Genotypes are small integers, so the product
// S . A is computed EXACTLY on the int8 matrix cores after writing the (fp64) matrix as D
// unsigned 7-bit digits ("Ozaki" splitting) of the entries shifted into the non-negative range:
//     2*A_jk = step * (sum_d 128^d u_d[j][k] - offset),  offset = 2^(7D-1),  k < j
// (round 1 used balanced base-256 digits; non-negative digit bytes cost the power-capped matrix pipe 7 % less per
// plane, gemm_i8_core.h SCAN_DIGIT_BITS -- the offset's contribution offset * sum_{j>k} s_j s_k is taken out again by
// the finalize kernels, which know (sum s)^2 - sum s^2).
// Only the strictly lower triangle is stored (A symmetric: s'As = sum_i A_ii s_i^2 +
// sum_{k<j} 2 A_jk s_j s_k), which halves the MFMA work; the diagonal term and s.w are
// evaluated in fp64 by the HBM-bound finalize kernel (one wave per 8 SNP rows, coalesced
// 16-byte genotype reads), which also evaluates F and the p-value.
//
// All integer partial sums are exact and accumulated with 64-bit integer atomics, so results
// are bitwise reproducible from run to run and independent of the tile schedule.
//
// This file: model quantisation, the finalize kernel and the p-value function.  The quadratic-form GEMM itself
// lives in k_scan_w4s.hip; superseded generations (8-wave, bit-packed, 16x16x64) are kept under experiments/ and are
// only compiled by `make EXPERIMENTS=1`.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstdio>
#include "f_sf.h"
#include "gemm_i8_core.h"
#include "mmg_internal.h"

namespace mmg {

// ------------------------------------------------------------------ model quantisation
// (Round 4: a grid-stride sweep with one atomic per wave of a 2048-block grid; the first version issued one atomic per wave
// of an N^2 / 256-block grid -- 390,000 atomics on one address, 2.3 ms at N = 5000 for a 200 MB read.)
__global__ __launch_bounds__(256) void absmax_offdiag_kernel(const double* __restrict__ A, int32_t N, unsigned long long* out) {
  const int64_t total = (int64_t)N * N, stride = (int64_t)gridDim.x * 256;
  double v = 0.0;
  for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += stride) {
    const int i = (int)(gid / N), j = (int)(gid - (int64_t)i * N);
    if (j < i) v = fmax(v, fabs(A[gid]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0 && v > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(v));
}

void launch_absmax_offdiag(mmg_ctx* ctx, const double* A, int32_t N, unsigned long long* out_bits) {
  const int64_t total = (int64_t)N * N;
  const unsigned blocks = (unsigned)std::min<int64_t>(2048, (total + 255) / 256);
  hipLaunchKernelGGL(absmax_offdiag_kernel, dim3(blocks), dim3(256), 0, ctx->stream, A, N, out_bits);
}

// one thread = 16 consecutive k of row j; Bq[d][j][k] = digit d (SCAN_DIGIT_BITS wide, unsigned) of
// rint(2 A[j][k] / step) + offset for the stored entries k < j < N, 0 elsewhere.  offset = 2^(7 D - 1) makes every stored
// value non-negative; the integer the GEMM accumulates is then q' = q + offset * sum_{j>k} s_j s_k, and the finalize
// kernels take the second term out again (it is a multiple of (sum s)^2 - sum s^2, which they have anyway).
__global__ void quantize_kernel(const double* __restrict__ A, int32_t N, int32_t Npad, int D, double inv_step,
                                long long offset, int8_t* __restrict__ Bq, double* __restrict__ diag,
                                long long* __restrict__ z0_sum,
                                long long* __restrict__ z0_tile /*[nJ][nJ] sums of the lowest digit per 256 x 256 tile*/) {
  const int chunks = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  long long z0acc = 0;                                   // sum of the lowest digits (adaptive scan: their mean is the
                                                         // bias of a pass that leaves that plane out)
  if (gid < (int64_t)Npad * chunks) {
  const int j = (int)(gid / chunks), c = (int)(gid % chunks);
  uint32_t out[6][4];
#pragma unroll
  for (int d = 0; d < 6; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) out[d][e] = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = c * 16 + e;
    long long Z = 0;
    if (j < N && k < j) Z = __double2ll_rn(2.0 * A[(int64_t)j * N + k] * inv_step) + offset;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      if (d < D) {
        const long long z = Z & ((1 << SCAN_DIGIT_BITS) - 1);
        Z >>= SCAN_DIGIT_BITS;
        if (d == 0) z0acc += z;
        out[d][e >> 2] |= ((uint32_t)z) << (8 * (e & 3));
      }
    }
  }
  for (int d = 0; d < D; ++d)
    *(uint4*)(Bq + ((int64_t)d * Npad + j) * Npad + c * 16) = make_uint4(out[d][0], out[d][1], out[d][2], out[d][3]);
  if (c == 0) diag[j] = (j < N) ? A[(int64_t)j * N + j] : 0.0;
  }
  if (z0_tile) {
    // one atomic per 16 lanes (= the 16 chunks of one row inside one 256-column tile; Npad is a multiple of 256, so a group of 16
    // consecutive lanes never straddles a row or a tile) instead of one per lane: 1.6 M atomics on 400 addresses were most of
    // this kernel's 1.0 ms at N = 5000
    long long v = z0acc;
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    if ((threadIdx.x & 15) == 0 && v != 0 && gid < (int64_t)Npad * chunks) {
      const int j = (int)(gid / chunks), c = (int)(gid % chunks);
      atomicAdd((unsigned long long*)(z0_tile + (int64_t)(j >> 8) * (Npad >> 8) + (c >> 4)), (unsigned long long)v);
    }
  }
  if (z0_sum) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z0acc += __shfl_xor(z0acc, o);
    if ((threadIdx.x & 63) == 0 && z0acc != 0) atomicAdd((unsigned long long*)z0_sum, (unsigned long long)z0acc);
  }
}

void launch_quantize(mmg_ctx* ctx, const double* A, int32_t N, int32_t Npad, int D, double inv_step, long long offset,
                     int8_t* Bq, double* diag, long long* z0_sum, long long* z0_tile) {
  const int64_t total = (int64_t)Npad * (Npad >> 4);
  hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, A, N, Npad,
                     D, inv_step, offset, Bq, diag, z0_sum, z0_tile);
}

// ------------------------------------------------------------------ p-value (f_sf.h)
__global__ void f_sf_kernel(const double* __restrict__ F, int64_t n, double nu, double lnbeta, double* __restrict__ p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n) p[gid] = f_sf_1(F[gid], nu, lnbeta);
}

void launch_f_sf(mmg_ctx* ctx, const double* F, int64_t n, int32_t df2, double lnbeta, double* p) {
  hipLaunchKernelGGL(f_sf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, F, n, (double)df2,
                     lnbeta, p);
}

// The integer the GEMM accumulated carries offset * sum_{j>k} s_j s_k on top of the quadratic form (quantize_kernel):
// taken out in 64-bit modular arithmetic -- exact whatever the number of planes, the true value is far below 2^63.
__device__ __forceinline__ double quad_without_offset(unsigned long long q, unsigned long long offset, long long sm,
                                                      long long sq) {
  const unsigned long long pairs = (unsigned long long)((sm * sm - sq) / 2);     // sum_{j>k} s_j s_k: (sum s)^2 - sum s^2 is even
  return (double)(long long)(q - offset * pairs);
}

// ------------------------------------------------------------------ finalize
// HBM-bound.  A block of 4 waves handles 32 SNP rows (8 per wave).  Per 1024-column chunk the block
// stages w and diag(A) once into LDS (lane-interleaved 16-byte units: conflict-free ds_read_b128)
// and every lane streams its 16 genotype bytes of each of its wave's 8 rows with one 16-byte load.
constexpr int FR = 8;                      // SNP rows per wave
constexpr int FIN_ROWS = 4 * FR;           // per block
__global__ __launch_bounds__(256, 2) void scan_finalize_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int64_t M, int32_t Npad, const double* __restrict__ w,
    const double* __restrict__ diag, const unsigned long long* __restrict__ q, double step, unsigned long long offset,
    double bias, double h0_rss, double nu,
    double lnbeta, double* __restrict__ rss, double* __restrict__ Fst, double* __restrict__ pv,
    double* __restrict__ dotv, double* __restrict__ denv, double* __restrict__ sumv, double* __restrict__ ddv,
    double* __restrict__ ssqv) {
  __shared__ double2 lw[8 * 64], ldg[8 * 64];          // [unit e>>1][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t m0 = (int64_t)blockIdx.x * FIN_ROWS + (tid >> 6) * FR;
  double dw[FR], dd[FR];
  int sm[FR], sq[FR];
#pragma unroll
  for (int rr = 0; rr < FR; ++rr) { dw[rr] = 0.0; dd[rr] = 0.0; sm[rr] = 0; sq[rr] = 0; }
  const int nchunks = Npad >> 4;
  for (int c0 = 0; c0 < nchunks; c0 += 64) {
    __syncthreads();
    {
      const int k = c0 * 16 + tid * 4;                  // 4 consecutive columns per thread
      double2 w0 = make_double2(0, 0), w1 = w0, d0 = w0, d1 = w0;
      if (k < Npad) {
        w0 = *(const double2*)(w + k); w1 = *(const double2*)(w + k + 2);
        d0 = *(const double2*)(diag + k); d1 = *(const double2*)(diag + k + 2);
      }
      const int l = tid >> 2, u = (tid & 3) * 2;        // owning lane, first 16-byte unit
      lw[u * 64 + l] = w0; lw[(u + 1) * 64 + l] = w1;
      ldg[u * 64 + l] = d0; ldg[(u + 1) * 64 + l] = d1;
    }
    __syncthreads();
    const int c = c0 + lane;
    if (c < nchunks) {
      double wv[16], dv[16];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double2 t = lw[u * 64 + lane], x = ldg[u * 64 + lane];
        wv[2 * u] = t.x; wv[2 * u + 1] = t.y; dv[2 * u] = x.x; dv[2 * u + 1] = x.y;
      }
#pragma unroll
      for (int rr = 0; rr < FR; ++rr) {
        // rows beyond M are inside the padded store (Mpad is a multiple of 256) and hold zeros
        const uint4 v = *(const uint4*)(S + (m0 + rr) * ldS + c * 16);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        // sum s and sum s^2 four bytes at a time on the integer dot-product unit (v_dot4_i32_i8)
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
          sm[rr] = __builtin_amdgcn_sdot4((int)wds[d4], 0x01010101, sm[rr], false);
          sq[rr] = __builtin_amdgcn_sdot4((int)wds[d4], (int)wds[d4], sq[rr], false);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int xi = (int)(int8_t)((wds[e >> 2] >> (8 * (e & 3))) & 0xff);
          const double xd = (double)xi;
          dw[rr] = fma(xd, wv[e], dw[rr]);
          dd[rr] = fma(xd * xd, dv[e], dd[rr]);
        }
      }
    }
  }
  double my_dw = 0.0, my_dd = 0.0;
  int my_sm = 0, my_sq = 0;
#pragma unroll
  for (int rr = 0; rr < FR; ++rr) {
    double a = dw[rr], b = dd[rr];
    int s = sm[rr], s2 = sq[rr];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
      s += __shfl_xor(s, o);
      s2 += __shfl_xor(s2, o);
    }
    if (lane == rr) { my_dw = a; my_dd = b; my_sm = s; my_sq = s2; }
  }
  const int64_t m = m0 + lane;
  if (lane < FR && m < M) {
    const double qd = quad_without_offset(q[m], offset, my_sm, my_sq);
    // bias: adaptive first pass only -- step * (mean of the digit plane that was left out), times the number of
    // (j > k) products sum_{j>k} s_j s_k = ((sum s)^2 - sum s^2) / 2: removes the coherent part of the rounding error
    const double den = fma(step, qd, my_dd) + bias * (0.5 * ((double)my_sm * (double)my_sm - (double)my_sq));
    const double num = my_dw * my_dw;
    double r = h0_rss;
    // den ~ 0: monomorphic after projection; the reference's lstsq returns no residual and rss
    // stays h0_rss (linear_models.py:1308,1329)
    if (den > 1e-7 * my_dd && den > 0.0) r = h0_rss - num / den;
    const double ratio = h0_rss / r;
    const double F = (ratio - 1.0) * nu;
    if (rss) rss[m] = r;
    if (Fst) Fst[m] = F;
    if (dotv) dotv[m] = my_dw;
    if (denv) denv[m] = den;
    if (sumv) sumv[m] = (double)my_sm;
    if (ddv) ddv[m] = my_dd;
    if (ssqv) ssqv[m] = (double)my_sq;
  }
}

// ---- adaptive precision (api.hip:mmg_emmax_scan_device): the first pass runs the three upper digit planes, i.e. the
// matrix truncated to 21 bits.  Its error in den = s'As, after the mean of the dropped digit has been put back, is a sum
// of ~(sum s^2)^2/2 independent errors uniform over a width step21 = 128 step:
//     sigma_m = step21 / sqrt(12) * (sum_i s_i^2) / sqrt(2)          (absolute, 1 sigma)
// and p moves by (F/2 + 1) * |d den| / den at most.  A SNP is refined (lowest plane added) when six sigma of that
// could move p by more than `target`: large F, or a den that is small against its own rounding noise (a SNP nearly
// collinear with the covariates).  idx[atomicAdd(cnt)] = m; order arbitrary, everything downstream is indexed.
__global__ void scan_select_kernel(const double* __restrict__ F, const double* __restrict__ den,
                                   const double* __restrict__ ssq, const unsigned long long* __restrict__ q, int64_t M,
                                   double sig_unit, double target, int use_F, int64_t* __restrict__ idx,
                                   unsigned long long* __restrict__ cnt) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const double d = den[m], six_sigma = 6.0 * sig_unit * ssq[m];
  // use_F = 0: the form is wanted for its own sake (permutation test): refine where six sigma exceed target * den
  const double Fm = use_F ? F[m] : 0.0;
  bool refine = !(d > 0.0) ? six_sigma > 0.0 : (0.5 * Fm + 1.0) * six_sigma > target * d;
  // plus about one SNP in 256 regardless -- picked by bits of its own first-pass integer, so that the choice does not
  // depend on where the SNP sits in the launch: a sample of thousands on which the refinement pass measures the
  // rounding error against its six-sigma prediction (the check that guards the error model), at the cost of one
  // more wave of workgroups in the second launch
  refine = refine || ((q[m] >> 13) & 255) == 17;   // (the low 8 bits of a first-pass integer are zero)
  if (refine) idx[atomicAdd(cnt, 1ull)] = m;
}

// compact store row i <- store row idx[i]; 16 bytes per thread
__global__ void gather_rows_kernel(const int8_t* __restrict__ S, int32_t Npad, const int64_t* __restrict__ idx, int64_t cnt,
                                   int8_t* __restrict__ Sc) {
  const int chunks = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= cnt * chunks) return;
  const int64_t i = gid / chunks;
  const int c = (int)(gid % chunks);
  *(uint4*)(Sc + i * (int64_t)Npad + c * 16) = *(const uint4*)(S + idx[i] * (int64_t)Npad + c * 16);
}

// q[idx[i]] += q2[i]; den / rss / F recomputed exactly as scan_finalize_kernel does from the full integer;
// eps = max |den_before / den_after - 1| (what the 21-bit pass was off by on this sample)
__global__ void scan_refine_kernel(const int64_t* __restrict__ idx, int64_t cnt, unsigned long long* __restrict__ q,
                                   const unsigned long long* __restrict__ q2, const double* __restrict__ dd,
                                   const double* __restrict__ dot, const double* __restrict__ ssq,
                                   const double* __restrict__ sumv, double sig_unit, double step,
                                   unsigned long long offset,
                                   double h0_rss, double nu,
                                   double* __restrict__ den, double* __restrict__ rss, double* __restrict__ Fst,
                                   unsigned long long* __restrict__ eps_bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double eps = 0.0, ratio = 0.0;
  if (i < cnt) {
    const int64_t m = idx[i];
    const unsigned long long qn = q[m] + q2[i];
    q[m] = qn;
    const double my_dd = dd[m], my_dw = dot[m];
    const double d_old = den[m];
    // the same expression as the finalize kernels evaluate for a scan over all planes (no first-pass bias)
    const double sm = sumv[m];
    const double d_new = fma(step, quad_without_offset(qn, offset, (long long)sm, (long long)ssq[m]), my_dd);
    const double num = my_dw * my_dw;
    double r = h0_rss;
    if (d_new > 1e-7 * my_dd && d_new > 0.0) r = h0_rss - num / d_new;
    den[m] = d_new;
    rss[m] = r;
    Fst[m] = (h0_rss / r - 1.0) * nu;
    if (d_new > 0.0) eps = fabs(d_old / d_new - 1.0);
    const double six_sigma = 6.0 * sig_unit * ssq[m];
    if (six_sigma > 0.0) ratio = fabs(d_old - d_new) / six_sigma;      // observed rounding error / its 6-sigma prediction
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    eps = fmax(eps, __shfl_xor(eps, o));
    ratio = fmax(ratio, __shfl_xor(ratio, o));
  }
  if ((threadIdx.x & 63) == 0) {
    if (eps > 0.0) atomicMax(eps_bits, (unsigned long long)__double_as_longlong(eps));
    if (ratio > 0.0) atomicMax(eps_bits + 1, (unsigned long long)__double_as_longlong(ratio));
  }
}

// ---- the exact tier (round 4).  D digit planes carry 7 D - 1 bits of the LARGEST entry of the matrix, so den = s'As is good to
// step / sqrt(24) * sum s^2 in absolute terms -- and a SNP that lies in the span of the kinship's large eigenvalues (few
// distinct genotype vectors, population-structure markers) has a den that is orders of magnitude below sum s^2 * max |A|:
// p moves by (F / 2 + 1) |d den| / den, 6e-6 at F = 7 on a kinship of 12 genotype classes with all four planes
// (tools/random_parity2.py).  After the planes are in, a SNP whose six-sigma bound still exceeds the target is recomputed from
// the fp64 matrix itself.  scan_select_exact_kernel: the same criterion as scan_select_kernel at the full-plane sigma, no sample.
// coherent: the matrix failed the noise test of the adaptive schedule (blocks that are constant to the last digit: a kinship of a
// few genotype classes) -- equal entries round the same way, the errors of a SNP's (sum s)^2 / 2 products add up instead of
// averaging, and the bound is the worst case step / 2 per product, not six sigma of independent ones (measured: 100 x six sigma).
__global__ void scan_select_exact_kernel(const double* __restrict__ F, const double* __restrict__ den,
                                         const double* __restrict__ ssq, const double* __restrict__ sumv, int64_t M,
                                         double sig_full, double half_step, int coherent, double target, int use_F,
                                         int64_t* __restrict__ idx, unsigned long long* __restrict__ cnt) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const double d = den[m];
  const double six_sigma = coherent ? half_step * 0.5 * fabs(sumv[m] * sumv[m] - ssq[m]) : 6.0 * sig_full * ssq[m];
  const double Fm = use_F ? F[m] : 0.0;
  const bool exact = !(d > 0.0) ? six_sigma > 0.0 : (0.5 * Fm + 1.0) * six_sigma > target * d;
  if (exact) idx[atomicAdd(cnt, 1ull)] = m;
}

// Sd[i][c] = (double) Sc[i][c] for the gathered rows (N columns, ld N): the uniform operand of scan_exact_den_kernel
__global__ void rows_to_f64_kernel(const int8_t* __restrict__ Sc, int32_t Npad, int32_t N, int64_t rows, double* __restrict__ Sd) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * N) return;
  const int64_t i = e / N;
  Sd[e] = (double)Sc[i * Npad + (e - i * N)];
}

// part[m][jb] = sum over the 256 columns j of block jb of s_mj (sum_i A_ij s_mi) for EX_SB gathered SNPs per workgroup: thread
// t owns column j = 256 jb + t, walks the rows of A (coalesced over the threads), and the SNPs' values s_mi are the same for
// every thread -- uniform loads (scalar cache) feeding v_fma_f64 as its SGPR operand, no LDS.  A is read once per EX_SB SNPs.
constexpr int EX_SB = 8;
__global__ __launch_bounds__(256) void scan_exact_den_kernel(const double* __restrict__ Sd, int32_t N, int64_t cnt,
                                                             const double* __restrict__ A64, double* __restrict__ part, int nJB) {
  const int tid = threadIdx.x, jb = blockIdx.x;
  const int64_t m0 = (int64_t)blockIdx.y * EX_SB;
  const int j = jb * 256 + tid;
  const bool valid = j < N;
  const int jc = valid ? j : N - 1;
  double acc[EX_SB];
#pragma unroll
  for (int m = 0; m < EX_SB; ++m) acc[m] = 0.0;
  const double* srow[EX_SB];
#pragma unroll
  for (int m = 0; m < EX_SB; ++m) srow[m] = Sd + (size_t)(m0 + m < cnt ? m0 + m : cnt - 1) * N;   // (rows past cnt: a valid row, result unused)
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    const double a = A64[(size_t)i * N + jc];
#pragma unroll
    for (int m = 0; m < EX_SB; ++m) acc[m] = fma(a, srow[m][i], acc[m]);
  }
  __shared__ double red[4][EX_SB];
#pragma unroll
  for (int m = 0; m < EX_SB; ++m) {
    double v = valid ? acc[m] * srow[m][jc] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) red[tid >> 6][m] = v;
  }
  __syncthreads();
  if (tid < EX_SB && m0 + tid < cnt) part[(size_t)(m0 + tid) * nJB + jb] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// den[idx[i]] = sum_jb part[i][jb] (fixed order: deterministic); rss and F as the finalize kernels compute them
__global__ void scan_exact_apply_kernel(const int64_t* __restrict__ idx, int64_t cnt, const double* __restrict__ part, int nJB,
                                        const double* __restrict__ dd, const double* __restrict__ dot, double h0_rss, double nu,
                                        double* __restrict__ den, double* __restrict__ rss, double* __restrict__ Fst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  double d = 0.0;
  for (int b = 0; b < nJB; ++b) d += part[(size_t)i * nJB + b];
  const int64_t m = idx[i];
  const double my_dd = dd[m], my_dw = dot[m];
  double r = h0_rss;
  if (d > 1e-7 * my_dd && d > 0.0) r = h0_rss - my_dw * my_dw / d;
  den[m] = d;
  rss[m] = r;
  Fst[m] = (h0_rss / r - 1.0) * nu;
}

// The check behind the tier's error model: for a sample of gathered SNPs, |den from the planes - den from the fp64 matrix| over
// the six-sigma bound of independent roundings.  max over the sample (bits of a non-negative double order like integers).
__global__ void scan_exact_check_kernel(const int64_t* __restrict__ idx, int64_t cnt, const double* __restrict__ part, int nJB,
                                        const double* __restrict__ den, const double* __restrict__ ssq, double sig_used,
                                        unsigned long long* __restrict__ ratio_bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  double d = 0.0;
  for (int b = 0; b < nJB; ++b) d += part[(size_t)i * nJB + b];
  const int64_t m = idx[i];
  const double six_sigma = 6.0 * sig_used * ssq[m];
  if (six_sigma > 0.0) atomicMax(ratio_bits, (unsigned long long)__double_as_longlong(fabs(den[m] - d) / six_sigma));
}
__global__ void scan_sample_idx_kernel(int64_t M, int64_t cnt, int64_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt) idx[i] = i * M / cnt;
}
void launch_scan_sample_idx(mmg_ctx* ctx, int64_t M, int64_t cnt, int64_t* idx) {
  hipLaunchKernelGGL(scan_sample_idx_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, M, cnt, idx);
}
void launch_scan_exact_check(mmg_ctx* ctx, const int64_t* idx, int64_t cnt, const double* part, int32_t N, const mmg_scan_result& res,
                             double sig_used, unsigned long long* ratio_bits) {
  hipLaunchKernelGGL(scan_exact_check_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, idx, cnt, part,
                     (N + 255) / 256, res.den, res.ssq, sig_used, ratio_bits);
}

void launch_scan_select_exact(mmg_ctx* ctx, const mmg_scan_result& res, int64_t M, double sig_full, double half_step,
                              bool coherent, double target, unsigned long long* cnt, bool use_F) {
  hipLaunchKernelGGL(scan_select_exact_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, res.F, res.den,
                     res.ssq, res.sum, M, sig_full, half_step, coherent ? 1 : 0, target, use_F ? 1 : 0, res.idx, cnt);
}
void launch_rows_to_f64(mmg_ctx* ctx, const int8_t* Sc, int32_t Npad, int32_t N, int64_t rows, double* Sd) {
  const int64_t total = rows * N;
  if (total <= 0) return;
  hipLaunchKernelGGL(rows_to_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, Sc, Npad, N, rows, Sd);
}
void launch_scan_exact_den(mmg_ctx* ctx, const double* Sd, int32_t N, int64_t cnt, const double* A64, double* part) {
  const int nJB = (N + 255) / 256;
  hipLaunchKernelGGL(scan_exact_den_kernel, dim3((unsigned)nJB, (unsigned)((cnt + EX_SB - 1) / EX_SB)), dim3(256), 0, ctx->stream,
                     Sd, N, cnt, A64, part, nJB);
}
void launch_scan_exact_apply(mmg_ctx* ctx, const int64_t* idx, int64_t cnt, const double* part, int32_t N, mmg_scan_result& res,
                             double h0_rss, int32_t df2) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(scan_exact_apply_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, idx, cnt, part,
                     (N + 255) / 256, res.dd, res.dot, h0_rss, (double)df2, res.den, res.rss, res.F);
}

void launch_scan_select(mmg_ctx* ctx, const mmg_scan_result& res, int64_t M, double sig_unit, double target,
                        unsigned long long* cnt, bool use_F) {
  hipLaunchKernelGGL(scan_select_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, res.F, res.den,
                     res.ssq, res.q, M, sig_unit, target, use_F ? 1 : 0, res.idx, cnt);
}
void launch_gather_rows(mmg_ctx* ctx, const mmg_geno* g, const int64_t* idx, int64_t cnt, int8_t* Sc) {
  const int64_t total = cnt * (g->Npad >> 4);
  if (total <= 0) return;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g->d, g->Npad,
                     idx, cnt, Sc);
}
void launch_scan_refine(mmg_ctx* ctx, const int64_t* idx, int64_t cnt, const mmg_scan_model& md, mmg_scan_result& res,
                        const unsigned long long* q2, double sig_unit, double h0_rss, int32_t df2,
                        unsigned long long* eps_bits) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(scan_refine_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, idx, cnt, res.q,
                     q2, res.dd, res.dot, res.ssq, res.sum, sig_unit, md.step, (unsigned long long)md.offset, h0_rss, (double)df2,
                     res.den, res.rss, res.F,
                     eps_bits);
}

// out[m] = s_m . v for an arbitrary fp64 vector v (zero padded to Npad); same streaming shape.
__global__ __launch_bounds__(256, 2) void snp_dot_kernel(const int8_t* __restrict__ S, int64_t ldS, int64_t M,
                                                         int32_t Npad, const double* __restrict__ v,
                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  constexpr int DR = 4;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * DR;
  if (m0 >= M) return;
  double dw[DR];
#pragma unroll
  for (int rr = 0; rr < DR; ++rr) dw[rr] = 0.0;
  for (int c = lane; c < (Npad >> 4); c += 64) {
    double wv[16];
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const double2 t = *(const double2*)(v + c * 16 + e);
      wv[e] = t.x; wv[e + 1] = t.y;
    }
#pragma unroll
    for (int rr = 0; rr < DR; ++rr) {
      const uint4 u = *(const uint4*)(S + (m0 + rr) * ldS + c * 16);
      const uint32_t wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 16; ++e)
        dw[rr] = fma((double)(int)(int8_t)((wds[e >> 2] >> (8 * (e & 3))) & 0xff), wv[e], dw[rr]);
    }
  }
  double mine = 0.0;
#pragma unroll
  for (int rr = 0; rr < DR; ++rr) {
    double a = dw[rr];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == rr) mine = a;
  }
  if (lane < DR && m0 + lane < M) out[m0 + lane] = mine;
}

void launch_snp_dot(mmg_ctx* ctx, const mmg_geno* g, const double* v, double* out) {
  const int64_t nwaves = (g->M + 3) / 4;
  hipLaunchKernelGGL(snp_dot_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, ctx->stream, g->d,
                     (int64_t)g->Npad, g->M, g->Npad, v, out);
}

// the same kernel over any int8 matrix [rows x ld] with row length len16 (a multiple of 16): out[r] = row_r . v
void launch_snp_dot_raw(mmg_ctx* ctx, const int8_t* S, int64_t ldS, int64_t rows, int32_t len16, const double* v,
                        double* out) {
  const int64_t nwaves = (rows + 3) / 4;
  hipLaunchKernelGGL(snp_dot_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, ctx->stream, S, ldS, rows, len16, v,
                     out);
}

// The same arithmetic from the by-products of the quadratic-form GEMM (k_scan_w4s.hip, LIN): the raw accumulators of the
// model's linear rows, raw[m][h][i] -- h = 0: digits 0..6 of s.w (i = 0..6) and sum s (i = 7); h = 1: digits 0..6 of
// sum_i A_ii s_i.  Binary store (sum s^2 = sum s); raw2 (store of 0/1/2 codes): [m][8] = digits 0..6 of sum_i A_ii [s_i = 2]
// and the number of 2s from lin_hi_bits_kernel below -- s^2 = s + 2 [s = 2].  72 (104) bytes per SNP instead of Npad.  Digits 0-3 and 4-6 are combined
// as two exact integers (below 2^52 and 2^44) and added in double precision.
__device__ __forceinline__ double digits7_to_f64(int4 lo4, int d4, int d5, int d6) {
  const long long lo = (long long)lo4.x + ((long long)lo4.y << 8) + ((long long)lo4.z << 16) + ((long long)lo4.w << 24);
  const long long hi = (long long)d4 + ((long long)d5 << 8) + ((long long)d6 << 16);
  return fma((double)hi, 4294967296.0, (double)lo);
}

__global__ void scan_finalize_lin_kernel(int64_t M, const unsigned long long* __restrict__ q, const int* __restrict__ raw,
                                         double step, unsigned long long offset, double step_w, double step_d, double bias,
                                         double h0_rss, double nu, const int* __restrict__ raw2,
                                         double* __restrict__ rss, double* __restrict__ Fst, double* __restrict__ dotv,
                                         double* __restrict__ denv, double* __restrict__ sumv, double* __restrict__ ddv,
                                         double* __restrict__ ssqv) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int4* rw = (const int4*)(raw + m * 16);
  const int4 a0 = rw[0], a1 = rw[1], b0 = rw[2], b1 = rw[3];
  const double my_dw = digits7_to_f64(a0, a1.x, a1.y, a1.z) * step_w;
  double my_dd = digits7_to_f64(b0, b1.x, b1.y, b1.z) * step_d;
  long long ssq_i = a1.w;
  if (raw2) {
    const int4 c0 = ((const int4*)(raw2 + m * 8))[0], c1 = ((const int4*)(raw2 + m * 8))[1];
    my_dd = fma(2.0 * step_d, digits7_to_f64(c0, c1.x, c1.y, c1.z), my_dd);
    ssq_i += 2 * (long long)c1.w;
  }
  const double sm = (double)a1.w;
  const double qd = quad_without_offset(q[m], offset, (long long)a1.w, ssq_i);
  const double den = fma(step, qd, my_dd) + bias * (0.5 * (sm * sm - (double)ssq_i));
  const double num = my_dw * my_dw;
  double r = h0_rss;
  if (den > 1e-7 * my_dd && den > 0.0) r = h0_rss - num / den;
  const double F = (h0_rss / r - 1.0) * nu;
  if (rss) rss[m] = r;
  if (Fst) Fst[m] = F;
  if (dotv) dotv[m] = my_dw;
  if (denv) denv[m] = den;
  if (sumv) sumv[m] = sm;
  if (ddv) ddv[m] = my_dd;
  if (ssqv) ssqv[m] = (double)ssq_i;
}

void launch_scan_finalize_lin(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, mmg_scan_result& res,
                              double h0_rss, int32_t df2, double lnbeta, bool with_p, double bias, const int* raw2) {
  hipLaunchKernelGGL(scan_finalize_lin_kernel, dim3((unsigned)((g->M + 255) / 256)), dim3(256), 0, ctx->stream, g->M, res.q,
                     res.linraw, md.step, (unsigned long long)md.offset, md.lin_step_w, md.lin_step_d, bias, h0_rss, (double)df2,
                     raw2,
                     res.rss, res.F, res.dot,
                     res.den, res.sum, res.dd, res.ssq);
  if (with_p && res.p && g->M > 0) launch_f_sf(ctx, res.F, g->M, df2, lnbeta, res.p);
}

// ------------------------------------------------------------------ stores of 0/1/2 codes: the [s = 2] bit image
// HBM-bound, once per store content: 16 genotype bytes -> 16 bits (bit 1 of every byte, LSB first).
__global__ __launch_bounds__(256) void pack_hi_bits_kernel(const int8_t* __restrict__ S, int64_t units, uint16_t* __restrict__ out) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const uint4 v = ((const uint4*)S)[u];
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned bits = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) bits |= ((((w[d] >> 1) & 0x01010101u) * 0x01020408u) >> 24) << (4 * d);   // byte i -> bit i
  out[u] = (uint16_t)bits;
}

void launch_pack_hi_bits(mmg_ctx* ctx, const mmg_geno* g) {
  const int64_t units = g->Mpad * (int64_t)(g->Npad >> 4);
  hipLaunchKernelGGL(pack_hi_bits_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, ctx->stream, g->d, units,
                     (uint16_t*)g->hi2);
}

// raw2[m][0..6] = sum_k digit_d(diag_k) [s_mk = 2], raw2[m][7] = sum_k [s_mk = 2]: a [8 x N] x [N x M] integer product on the
// matrix cores with the SNP operand expanded from the bit image in registers -- 0.6 GB of HBM at N = 5000 x M = 1e6 where
// the finalize pass over the store reads 5 GB.  A wave owns 64 SNPs (two 32-column tiles that share every table fragment,
// read from a chunk of the table in LDS); per block of 1024 individuals a lane loads up to 64 contiguous image bytes per
// tile (its SNP's bits of half h of the block) and runs v_mfma_i32_32x32x32_i8 on 16 bits at a time: slot (h, i) of the K
// dimension of MFMA t is individual k0 + h * half + 16 t + i for both operands.  A = the table (rows 0..7, lanes of rows
// 8..31 hold zeros), B = the SNPs; in the result lane (snp, h) holds digits 4 h .. 4 h + 3 of its SNP in registers 0..3.
__device__ __forceinline__ v4i expand16_bits(unsigned x16) {
  v4i o;
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (int)((((x16 >> (4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u);   // bit i -> byte i
  return o;
}

constexpr int HB_KC = 4096;                 // individuals per table chunk in LDS
constexpr int HB_LD = HB_KC + 16;           // row stride of the chunk: the eight rows start four banks apart
__global__ __launch_bounds__(256) void lin_hi_bits_kernel(const uint8_t* __restrict__ img, int64_t ld_img, int64_t Mpad,
                                                          int32_t Npad, const int8_t* __restrict__ tab,
                                                          int* __restrict__ raw2) {
  extern __shared__ __attribute__((aligned(16))) char hb_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 64;        // grid = Mpad / 256 blocks: every wave has its 64 SNPs
  const uint8_t* p0 = img + (m0 + r) * ld_img;
  const uint8_t* p1 = p0 + 32 * ld_img;
  const bool row = r < 8;
  v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
  for (int c0 = 0; c0 < Npad; c0 += HB_KC) {
    const int kc = min(HB_KC, Npad - c0);
    __syncthreads();
    for (int u = threadIdx.x; u < 8 * (kc >> 4); u += 256) {       // the chunk of the table: 16-byte units, row-major
      const int rr = u / (kc >> 4), cc = u % (kc >> 4);
      *(v4i*)(hb_lds + rr * HB_LD + cc * 16) = *(const v4i*)(tab + (int64_t)rr * Npad + c0 + cc * 16);
    }
    __syncthreads();
    // K blocks of 1024 individuals (768 / 512 / 256 at the end): lane (., h) takes the block's half h, i.e. up to 64
    // contiguous image bytes of its SNP -- a whole 128-byte line per SNP and block between the two halves
    for (int k0 = 0; k0 < kc; k0 += 1024) {
      const int half = min(1024, kc - k0) >> 1;                    // 512, 384, 256 or 128 individuals
      const int nq = half >> 7;                                    // 16-byte pieces per lane: 4, 3, 2, 1
      const int kb = c0 + k0 + h * half;
      uint4 x0[4], x1[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        if (qd < nq) { x0[qd] = *(const uint4*)(p0 + (kb >> 3) + qd * 16); x1[qd] = *(const uint4*)(p1 + (kb >> 3) + qd * 16); }
      const char* tl = hb_lds + (r & 7) * HB_LD + (k0 + h * half);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (qd >= nq) break;
        const unsigned w0[4] = {x0[qd].x, x0[qd].y, x0[qd].z, x0[qd].w}, w1[4] = {x1[qd].x, x1[qd].y, x1[qd].z, x1[qd].w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          v4i a = v4i{0, 0, 0, 0};
          if (row) a = *(const v4i*)(tl + qd * 128 + 16 * t);
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, expand16_bits((w0[t >> 1] >> (16 * (t & 1))) & 0xffffu), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, expand16_bits((w1[t >> 1] >> (16 * (t & 1))) & 0xffffu), acc1, 0, 0, 0);
        }
      }
    }
  }
  *(v4i*)(raw2 + (m0 + r) * 8 + h * 4) = v4i{acc0[0], acc0[1], acc0[2], acc0[3]};
  *(v4i*)(raw2 + (m0 + 32 + r) * 8 + h * 4) = v4i{acc1[0], acc1[1], acc1[2], acc1[3]};
}

void launch_lin_hi_bits(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, int* raw2) {
  const int lds = 8 * HB_LD;
  hipFuncSetAttribute((const void*)lin_hi_bits_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(lin_hi_bits_kernel, dim3((unsigned)(g->Mpad / 256)), dim3(256), lds, ctx->stream, g->hi2,
                     (int64_t)(g->Npad >> 3), g->Mpad, g->Npad, md.lin_tab, raw2);
}

void launch_scan_finalize(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, mmg_scan_result& res,
                          double h0_rss, int32_t df2, double lnbeta, bool with_p, double bias) {
  hipLaunchKernelGGL(scan_finalize_kernel, dim3((unsigned)(g->Mpad / FIN_ROWS)), dim3(256), 0, ctx->stream, g->d,
                     (int64_t)g->Npad, g->M, g->Npad, md.w, md.diag, res.q, md.step, (unsigned long long)md.offset, bias, h0_rss,
                     (double)df2, lnbeta,
                     res.rss, res.F, res.p, res.dot, res.den, res.sum, res.dd, res.ssq);
  // p-values in their own launch: one lane per SNP (in the finalize kernel only 8 of 64 lanes hold a
  // finished SNP, and the continued fraction is ~100 dependent fp64 divisions long)
  if (with_p && res.p && g->M > 0) launch_f_sf(ctx, res.F, g->M, df2, lnbeta, res.p);
}

}  // namespace mmg
