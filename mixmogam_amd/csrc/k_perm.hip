// k_perm.hip -- EMMAX permutation test (replaces the per-SNP multi-RHS lstsq loop of
// linear_models.py:1157-1164).  For permutation p and SNP m
//     rss_{m,p} = Ys_p.Ys_p - (t_m.Ys_p)^2 / (t_m.t_m),   t_m = H (s_m - mean(s_m))      (:1159-1163)
// so  min_m rss_{m,p} = Ys_p.Ys_p - max_m G_{m,p}^2 / tt_m  with
//     G_{m,p} = s_m.W_p - mu_m * sum(W_p),  W = H'Ys  [N x P],   tt_m = s~' (H'H) s~.
// tt comes from the scan's quadratic-form GEMM (model H'H); G is the GEMM S . W computed exactly
// on the int8 matrix cores with W written as 4 unsigned 7-bit digits per permutation column
// (per-column scale, entries shifted into the non-negative range: gemm_i8_w4s.h ROWS_OFFSET).  The four digit rows of one permutation sit in the four 32-row MFMA tiles
// of ONE wave, so the digits are recombined in registers (exact int64), squared, scaled by
// 1/tt_m and max-reduced over SNPs without ever leaving the chip; one 64-bit atomicMax per
// (permutation, workgroup) at the end (order independent -> reproducible).
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>
#include "gemm_i8_core.h"
#include "gemm_i8_w4s.h"
#include "mmg_internal.h"

namespace mmg {

constexpr int PERM_TILE = 64;                       // permutations per workgroup tile
constexpr int PERM_LDS_EXTRA = PERM_TILE * 16;      // step + csum per permutation (fp64)

__global__ void perm_center_kernel(const double* __restrict__ den, const double* __restrict__ dot,
                                   const double* __restrict__ sum, int64_t M, int64_t Mpad, double invN, double c0,
                                   double* __restrict__ mu, double* __restrict__ inv) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mpad) return;
  double u = 0.0, iv = 0.0;
  if (m < M) {
    u = sum[m] * invN;
    const double tt = den[m] - 2.0 * u * dot[m] + u * u * c0;
    if (tt > 1e-7 * fabs(den[m]) && tt > 0.0) iv = 1.0 / tt;
  }
  mu[m] = u;
  inv[m] = iv;
}

void launch_perm_center(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_result& r, double c0, double* d_mu,
                        double* d_inv) {
  hipLaunchKernelGGL(perm_center_kernel, dim3((unsigned)((g->Mpad + 255) / 256)), dim3(256), 0, ctx->stream, r.den,
                     r.dot, r.sum, g->M, g->Mpad, 1.0 / (double)g->N, c0, d_mu, d_inv);
}

// ---- the stand-alone test works on CENTRED operands: with C = I - 11'/N (:1159 centres every SNP),
//     t.t = s~'A's~ = s'(C A' C)s,   t.Ys_p = s~.W_p = s.(C W_p),
// so the quadratic form of the model C A' C IS t.t -- no den - 2 mu s.v + mu^2 c0 with its cancellation, hence the
// adaptive digit schedule of the scan applies to it (three planes + the fourth where six sigma exceed 2.5e-7 of t.t
// itself) -- and the GEMM needs no mu * sum(W_p) term.
__global__ void center_sym_kernel(double* __restrict__ A, int32_t N, const double* __restrict__ v, double c0) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)N * N) return;
  const int j = (int)(gid / N), k = (int)(gid % N);
  const double invN = 1.0 / (double)N;
  A[gid] = A[gid] - (v[j] + v[k]) * invN + c0 * invN * invN;
}
void launch_center_sym(mmg_ctx* ctx, double* A, int32_t N, const double* v, double c0) {
  const int64_t total = (int64_t)N * N;
  hipLaunchKernelGGL(center_sym_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, A, N, v, c0);
}

// one wave per row
__global__ __launch_bounds__(256) void center_rows_kernel(double* __restrict__ Wt, int32_t N, int32_t P) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  double s = 0.0;
  for (int k = lane; k < N; k += 64) s += Wt[(int64_t)p * N + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const double mean = s / (double)N;
  for (int k = lane; k < N; k += 64) Wt[(int64_t)p * N + k] -= mean;
}
// A [N x N] row-major: every row minus colsum / N (the column-centring C A of a matrix held row by row)
__global__ void sub_row_mean_kernel(double* __restrict__ A, int32_t N, const double* __restrict__ colsum) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)N * N) A[e] -= colsum[e % N] / (double)N;
}
void launch_sub_row_mean(mmg_ctx* ctx, double* A, int32_t N, const double* colsum) {
  const int64_t total = (int64_t)N * N;
  hipLaunchKernelGGL(sub_row_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, A, N, colsum);
}

void launch_center_rows(mmg_ctx* ctx, double* Wt, int32_t N, int32_t P) {
  hipLaunchKernelGGL(center_rows_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, ctx->stream, Wt, N, P);
}

// mu = 0 (the operands are centred), inv = 1 / t.t with the scan's rule for a form that vanishes (a SNP that is
// constant over the individuals: den ~ 0 against its own diagonal part)
__global__ void perm_inv_kernel(const double* __restrict__ den, const double* __restrict__ dd, int64_t M, int64_t Mpad,
                                double* __restrict__ mu, double* __restrict__ inv) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mpad) return;
  double iv = 0.0;
  if (m < M) {
    const double tt = den[m];
    if (tt > 1e-7 * fabs(dd[m]) && tt > 0.0) iv = 1.0 / tt;
  }
  mu[m] = 0.0;
  inv[m] = iv;
}
void launch_perm_inv(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_result& r, double* d_mu, double* d_inv) {
  hipLaunchKernelGGL(perm_inv_kernel, dim3((unsigned)((g->Mpad + 255) / 256)), dim3(256), 0, ctx->stream, r.den, r.dd,
                     g->M, g->Mpad, d_mu, d_inv);
}

// tt from the quadratic form a preceding EMMAX scan of the same SNPs left behind: den = s'(H'H - sum_c u_c u_c')s with
// u_c = H'Q_c (linear_models.py:1300-1303), so s'H'Hs = den + sum_c (s.u_c)^2 and, centred (:1159),
// tt = s'H'Hs - 2 mu s.v + mu^2 c0 with v = H'H 1.  dots: [1 + q][Mpad] = s.v, s.u_0 .. s.u_{q-1}.
__global__ void perm_center_reuse_kernel(const double* __restrict__ den, const double* __restrict__ dots, int q,
                                         const double* __restrict__ sum, int64_t M, int64_t Mpad, double invN, double c0,
                                         double* __restrict__ mu, double* __restrict__ inv) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mpad) return;
  double u = 0.0, iv = 0.0;
  if (m < M) {
    u = sum[m] * invN;
    double full = den[m];
    for (int c = 0; c < q; ++c) { const double d = dots[(int64_t)(1 + c) * Mpad + m]; full = fma(d, d, full); }
    const double tt = full - 2.0 * u * dots[m] + u * u * c0;
    if (tt > 1e-7 * fabs(full) && tt > 0.0) iv = 1.0 / tt;
  }
  mu[m] = u;
  inv[m] = iv;
}

void launch_perm_center_reuse(mmg_ctx* ctx, const mmg_geno* g, const double* den, const double* dots, int q,
                              const double* sum, double c0, double* d_mu, double* d_inv) {
  hipLaunchKernelGGL(perm_center_reuse_kernel, dim3((unsigned)((g->Mpad + 255) / 256)), dim3(256), 0, ctx->stream, den,
                     dots, q, sum, g->M, g->Mpad, 1.0 / (double)g->N, c0, d_mu, d_inv);
}

// one wave per permutation row: max |W_p|, sum W_p
__global__ __launch_bounds__(256) void perm_rowstat_kernel(const double* __restrict__ Wt, int32_t N, int32_t P,
                                                           double* __restrict__ step, double* __restrict__ csum) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  double mx = 0.0, s = 0.0;
  for (int k = lane; k < N; k += 64) {
    const double v = Wt[(int64_t)p * N + k];
    mx = fmax(mx, fabs(v));
    s += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmax(mx, __shfl_xor(mx, o));
    s += __shfl_xor(s, o);
  }
  if (lane == 0) {
    if (!(mx > 0.0)) mx = 1.0;
    step[p] = mx / ROWS_ZMAX;          // |rint(W/step)| <= 2^27 - 1: shifted by 2^27 it is four unsigned 7-bit digits
    csum[p] = s;
  }
}

// Wq [nPT][256][Npad]: row wmi*128 + d*32 + pl of tile pt holds digit d of permutation pt*64 + wmi*32 + pl
__global__ void perm_quantize_kernel(const double* __restrict__ Wt, int32_t N, int32_t Npad, int32_t P,
                                     const double* __restrict__ step, int8_t* __restrict__ Wq) {
  const int chunks = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Ppad = (P + PERM_TILE - 1) / PERM_TILE * PERM_TILE;
  if (gid >= (int64_t)Ppad * chunks) return;
  const int p = (int)(gid / chunks), c = (int)(gid % chunks);
  uint32_t out[4][4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) out[d][e] = 0;
  if (p < P) {
    const double inv = 1.0 / step[p];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = c * 16 + e;
      // padding columns (k >= N) hold digit 0 in every plane: they meet zero genotypes, and an all-zero operand
      // column keeps the GEMM's accumulators independent of what the offset would have put there
      if (k < N) {
        const long long Z = __double2ll_rn(Wt[(int64_t)p * N + k] * inv) + ROWS_OFFSET;   // in [1, 2^28 - 1]
#pragma unroll
        for (int d = 0; d < 4; ++d)
          out[d][e >> 2] |= ((uint32_t)((Z >> (ROWS_DIGIT_BITS * d)) & 127)) << (8 * (e & 3));
      }
    }
  }
  const int pt = p / PERM_TILE, wmi = (p % PERM_TILE) / 32, pl = p % 32;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int row = wmi * 128 + d * 32 + pl;
    *(uint4*)(Wq + ((int64_t)pt * TM + row) * Npad + c * 16) = make_uint4(out[d][0], out[d][1], out[d][2], out[d][3]);
  }
}

__global__ __launch_bounds__(NTHREADS, 2) void perm_gemm_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Wq, int64_t ldW, int nPT,
    int nch, int sb_per_chunk, int nks, const double* __restrict__ step, const double* __restrict__ csum,
    const double* __restrict__ mu, const double* __restrict__ inv, const double* __restrict__ ssum,
    unsigned long long* __restrict__ maxstat) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = blockIdx.x;
  const int x = b & 7, i = b >> 3;
  const int pt = x + 8 * (i / nch), chunk = i % nch;
  if (pt >= nPT) return;
  const int sb0 = chunk * sb_per_chunk;
  const int sb1 = min(sb0 + sb_per_chunk, nSb);
  if (sb0 >= sb1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
  // per-permutation scale and column sum for this tile -> LDS (behind the staging buffers)
  double* ex = (double*)(lds + LDS_BYTES);
  if (threadIdx.x < PERM_TILE) {
    ex[threadIdx.x] = step[pt * PERM_TILE + threadIdx.x];
    ex[PERM_TILE + threadIdx.x] = csum[pt * PERM_TILE + threadIdx.x];
  }
  __syncthreads();
  const int8_t* P = Wq + (int64_t)pt * TM * ldW;
  double maxv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) maxv[e] = 0.0;
  for (int sb = sb0; sb < sb1; ++sb) {
    const int8_t* Q = S + (int64_t)sb * TN * ldS;
    v16i acc[4][2];
    gemm_tile_i8(P, ldW, Q, ldS, 0, nks, lds, acc);
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int64_t snp = (int64_t)sb * TN + wn * 64 + nn * 32 + r;
      const double u = mu[snp], iv = inv[snp];
      const int ss = (int)ssum[snp];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const double gd = digits4_to_f64<false>(acc[0][nn][e], acc[1][nn][e], acc[2][nn][e], acc[3][nn][e], ss);
        const double G = fma(gd, ex[pl], -u * ex[PERM_TILE + pl]);
        maxv[e] = fmax(maxv[e], G * G * iv);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    double v = maxv[e];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if (r == 0) {
      const int p = pt * PERM_TILE + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      atomicMax(maxstat + p, (unsigned long long)__double_as_longlong(v));   // v >= 0: bit order == value order
    }
  }
}

// The same GEMM + max-reduce on the 4-wave job stream of gemm_i8_w4s.h (wave tile 128 x 128: the four digit rows
// of 32 permutations x 128 SNPs; the SNP blocks of the workgroup's chunk are the jobs of ONE pipeline, so the
// prefetch never drains between blocks).  Bit-identical to perm_gemm_kernel: the integers are exact and the
// maximum is order independent.
template <bool FAST>
__global__ __launch_bounds__(W4_THREADS) void perm_gemm_w4_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Wq, int64_t ldW, int nPT,
    const int2* __restrict__ groups, int nch, int sb_per_chunk, int nks, const double* __restrict__ step,
    const double* __restrict__ csum, const double* __restrict__ mu, const double* __restrict__ inv,
    const double* __restrict__ ssum, unsigned long long* __restrict__ maxstat) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int pt, chunk;
  if (!w4_group_place(groups, blockIdx.x, nPT, nch, pt, chunk)) return;
  const int sb0 = chunk * sb_per_chunk;
  const int sb1 = min(sb0 + sb_per_chunk, nSb);
  if (sb0 >= sb1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  double* ex = (double*)(lds + LDS_BYTES);
  if (threadIdx.x < PERM_TILE) {
    ex[threadIdx.x] = step[pt * PERM_TILE + threadIdx.x];
    ex[PERM_TILE + threadIdx.x] = csum[pt * PERM_TILE + threadIdx.x];
  }
  __syncthreads();
  const int8_t* P = Wq + (int64_t)pt * TM * ldW;
  double maxv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) maxv[e] = 0.0;
  double u[4], iv[4];
  int ss[4];
  w4s_stream(
      sb0, sb1, ldW, ldS, lds,
      [&](int sb) { return W4Job{P, S + (int64_t)sb * TN * ldS, nks}; },
      [&](int sb) {
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int64_t snp = (int64_t)sb * TN + wn * 128 + nn * 32 + r;
          u[nn] = mu[snp];
          iv[nn] = inv[snp];
          ss[nn] = (int)ssum[snp];
        }
      },
      [&](int, v16i (&acc)[4][4]) {
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int pl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const double gd = digits4_to_f64<FAST>(acc[0][nn][e], acc[1][nn][e], acc[2][nn][e], acc[3][nn][e], ss[nn]);
            const double G = fma(gd, ex[pl], -u[nn] * ex[PERM_TILE + pl]);
            maxv[e] = fmax(maxv[e], G * G * iv[nn]);
          }
        }
      });
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    double v = maxv[e];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if (r == 0) {
      const int p = pt * PERM_TILE + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      atomicMax(maxstat + p, (unsigned long long)__double_as_longlong(v));
    }
  }
}

int upload_group_table(mmg_ctx* ctx, const std::vector<int2>& tab) {
  if (tab.size() > ctx->grp_cap) {
    if (ctx->grp_tab) MMG_HIP(ctx, hipFree(ctx->grp_tab));
    ctx->grp_tab = nullptr;
    ctx->grp_cap = 0;
    MMG_HIP(ctx, hipMalloc(&ctx->grp_tab, 2 * tab.size() * sizeof(int2)));
    ctx->grp_cap = 2 * tab.size();
  }
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));       // an earlier upload may still read grp_host
  ctx->grp_host = tab;
  MMG_HIP(ctx, hipMemcpyAsync(ctx->grp_tab, ctx->grp_host.data(), tab.size() * sizeof(int2), hipMemcpyHostToDevice,
                              ctx->stream));
  return MMG_OK;
}

// Rows of Wt [P x N] (device fp64) -> four unsigned 7-bit digits per row (entries shifted by ROWS_OFFSET) with a per-row step, in the operand
// layout of the 256-row P tile (64 rows x 4 digits per tile; see perm_quantize_kernel).  Wq: [ceil(P/64)][256][Npad],
// dstep / dcsum: [ceil(P/64)*64] (zero beyond P).  Shared with the eigen-rotation GEMM (k_rot.hip).
int quantize_rows_4digits(mmg_ctx* ctx, const double* dWt, int32_t N, int32_t Npad, int32_t P, int8_t* Wq, double* dstep,
                          double* dcsum) {
  const int nPT = (P + PERM_TILE - 1) / PERM_TILE;
  const int Ppad = nPT * PERM_TILE;
  MMG_HIP(ctx, hipMemsetAsync(dstep, 0, Ppad * sizeof(double), ctx->stream));
  MMG_HIP(ctx, hipMemsetAsync(dcsum, 0, Ppad * sizeof(double), ctx->stream));
  hipLaunchKernelGGL(perm_rowstat_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, ctx->stream, dWt, N, P, dstep,
                     dcsum);
  const int64_t total = (int64_t)Ppad * (Npad >> 4);
  hipLaunchKernelGGL(perm_quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dWt, N,
                     Npad, P, dstep, Wq);
  MMG_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

int run_perm(mmg_ctx* ctx, const mmg_geno* g, int32_t N, const double* dWt, int32_t P, const double* d_inv,
             const double* d_mu, const double* d_ssum, int ndigits, double* d_maxstat) {
  (void)ndigits;
  const int Npad = g->Npad;
  const int nPT = (P + PERM_TILE - 1) / PERM_TILE;
  const int Ppad = nPT * PERM_TILE;
  Scratch sc;
  double *dstep = nullptr, *dcsum = nullptr;
  int8_t* Wq = nullptr;
  MMG_HIP(ctx, sc.alloc(&dstep, Ppad * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dcsum, Ppad * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&Wq, (size_t)nPT * TM * Npad));
  int rcq = quantize_rows_4digits(ctx, dWt, N, Npad, P, Wq, dstep, dcsum);
  if (rcq) return rcq;
  return run_perm_q(ctx, g, Wq, dstep, dcsum, P, d_inv, d_mu, d_ssum, d_maxstat);
}

// the GEMM + max-reduce over a digit image of W prepared once (mmg_perm_plan): Wq [ceil(P/64)][256][Npad];
// d_ssum [Mpad]: the exact genotype sum of every SNP (takes the digit offset out of the accumulators)
int run_perm_q(mmg_ctx* ctx, const mmg_geno* g, const int8_t* Wq, const double* dstep, const double* dcsum, int32_t P,
               const double* d_inv, const double* d_mu, const double* d_ssum, double* d_maxstat) {
  const int Npad = g->Npad;
  const int nPT = (P + PERM_TILE - 1) / PERM_TILE;
  const int Ppad = nPT * PERM_TILE;
  const int nSb = (int)(g->Mpad / TN);
  MMG_HIP(ctx, hipMemsetAsync(d_maxstat, 0, Ppad * sizeof(double), ctx->stream));
  // grid: permutation tiles x SNP-block chunks, ~4 workgroups per CU in flight over the launch
  const int rounds = (nPT + 7) / 8;
  int nch = std::max(1, (4 * 256) / (8 * rounds));
  nch = std::min(nch, nSb);
  const int per = (nSb + nch - 1) / nch;
  nch = (nSb + per - 1) / per;
  // MMG_PERM_KERNEL=w8: the first-generation 8-wave kernel (A/B runs; same bits)
  static const bool w8 = [] { const char* e = std::getenv("MMG_PERM_KERNEL"); return e && std::string(e) == "w8"; }();
  if (w8) {
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)perm_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     LDS_BYTES + PERM_LDS_EXTRA));
    EvScope ev(ctx, EV_PERM);
    hipLaunchKernelGGL(perm_gemm_kernel, dim3((unsigned)(8 * rounds * nch)), dim3(NTHREADS), LDS_BYTES + PERM_LDS_EXTRA,
                       ctx->stream, g->d, (int64_t)Npad, nSb, Wq, (int64_t)Npad, nPT, nch, per, Npad / BK, dstep, dcsum,
                       d_mu, d_inv, d_ssum, (unsigned long long*)d_maxstat);
  } else {
    // groups of 2 permutation tiles x 16 SNP chunks per XCD (gemm_i8_w4s.h: w4_group_table)
    int GV = 2;
    if (const char* e = std::getenv("MMG_PERM_GV")) GV = std::max(1, std::min(32, std::atoi(e)));
    int nch4 = std::min(nSb, std::max(32, (4 * 256) / (8 * rounds) / 32 * 32));
    const int per4 = (nSb + nch4 - 1) / nch4;
    nch4 = (nSb + per4 - 1) / per4;
    const std::vector<int2> tab = w4_group_table(rounds, nch4, GV);
    int rct = upload_group_table(ctx, tab);
    if (rct) return rct;
    const bool fast = w4_digits_fast(g->smax, Npad) && !std::getenv("MMG_W4_SLOW_EPI");
    const void* fn = fast ? (const void*)perm_gemm_w4_kernel<true> : (const void*)perm_gemm_w4_kernel<false>;
    MMG_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + PERM_LDS_EXTRA));
    EvScope ev(ctx, EV_PERM);
#define MMG_LAUNCH_PERM_W4(F)                                                                                          \
  hipLaunchKernelGGL(perm_gemm_w4_kernel<F>, dim3((unsigned)(256 * tab.size())), dim3(W4_THREADS),                     \
                     LDS_BYTES + PERM_LDS_EXTRA, ctx->stream, g->d, (int64_t)Npad, nSb, Wq, (int64_t)Npad, nPT,        \
                     ctx->grp_tab, nch4, per4, Npad / BK, dstep, dcsum, d_mu, d_inv, d_ssum,                            \
                     (unsigned long long*)d_maxstat)
    if (fast) MMG_LAUNCH_PERM_W4(true); else MMG_LAUNCH_PERM_W4(false);
#undef MMG_LAUNCH_PERM_W4
  }
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

}  // namespace mmg
