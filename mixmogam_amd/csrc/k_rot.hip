// k_rot.hip -- eigen-rotated genotype store and the multi-phenotype EMMAX scan over it.
//
// The reference scans several phenotypes over the same genotypes and kinship as a LOOP of independent
// LinearMixedModel / emmax() runs (phenotypeData.py:70-78 over pids, hdf5_data.py:262-330 per file): every
// phenotype has its own variance ratio delta_p, hence its own H_p = diag((lambda+delta_p)^-1/2) U' and its own
// N x N scan matrix.  With K = U' diag(lambda) U shared, everything SNP-dependent of every such scan is a function
// of the ROTATED SNP tau_m = U s_m alone (linear_models.py:898,1290-1303,1328 written in the eigenbasis):
//     t_m   = (I - Q_p Q_p')(w_p * tau_m),            w_p = (lambda + delta_p)^-1/2
//     den   = t_m.t_m = sum_i w_pi^2 tau_mi^2 - sum_c (sum_i Q_p[i][c] w_pi tau_mi)^2
//     dot   = t_m.r_p = sum_i w_pi r_pi tau_mi        (r_p: residual of the transformed phenotype, Q_p'r_p = 0)
//     rss   = h0_rss_p - dot^2 / den,   F = (h0_rss_p / rss - 1) df2,   p = f.sf(F, 1, df2)
// so the O(N^2) work per SNP is done ONCE (T = S U', an exact int8-MFMA digit GEMM, 4 balanced base-256 digits per
// eigenvector with a per-eigenvector step -- the operand layout and mainloop of the permutation GEMM, k_perm.hip),
// kept in HBM as fp64, eigen-major inside blocks of 256 SNPs (T[m / 256][i][m % 256]: a scan workgroup streams one
// contiguous 2 KB x N region, page after page; 8 N bytes per SNP: 41 GB at N = 5000, M = 1e6 -- what 288 GB are for),
// and every phenotype afterwards costs one HBM-bound pass of 2 + q fused multiply-adds per element:
//   * rot_gemm_kernel:  one register of a 32x32 accumulator = 32 consecutive SNPs of one eigen-coordinate, so the
//     eigen-major store is written as the accumulators stand (256-B segments, no shuffle);
//   * scan_multi_kernel: one lane per SNP, coalesced 512-B reads of T[i][m..m+63]; the 2 + q coefficients of each of
//     up to 8 phenotypes are wave-uniform (scalar loads, SGPR operands of v_fma_f64): at 8 phenotypes x (2 + 1)
//     columns the fp64 VALU rate (16 lanes/clk/SIMD) and the HBM stream (8 B per element) are in balance.
// Roofline of the pass: HBM, 8 N bytes per SNP per <= 8 phenotypes.  Roofline of the rotation: int8 MFMA,
// 2 * 4 * N^2 ops per SNP (no symmetry to exploit -- twice the work of one single-phenotype scan, amortised over
// every phenotype that follows).
#include <algorithm>
#include <cstdlib>
#include "f_sf.h"
#include <string>
#include <vector>
#include "gemm_i8_core.h"
#include "gemm_i8_w4s.h"
#include "mmg_internal.h"

namespace mmg {

constexpr int ROT_TILE = 64;                       // eigenvectors per workgroup tile (x 4 digits = 256 operand rows)

__global__ __launch_bounds__(NTHREADS, 2) void rot_gemm_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Vq, int64_t ldV, int nVT, int nch,
    int sb_per_chunk, int nks, const double* __restrict__ step, double* __restrict__ T, int64_t nrows) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = blockIdx.x;
  const int x = b & 7, i = b >> 3;
  const int vt = x + 8 * (i / nch), chunk = i % nch;       // the 32 CUs of an XCD share one eigen tile (L2-resident)
  if (vt >= nVT) return;
  const int sb0 = chunk * sb_per_chunk;
  const int sb1 = min(sb0 + sb_per_chunk, nSb);
  if (sb0 >= sb1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
  double* ex = (double*)(lds + LDS_BYTES);                 // per-eigenvector step of this tile
  if (threadIdx.x < ROT_TILE) ex[threadIdx.x] = step[vt * ROT_TILE + threadIdx.x];
  __syncthreads();
  const int8_t* P = Vq + (int64_t)vt * TM * ldV;
  for (int sb = sb0; sb < sb1; ++sb) {
    const int8_t* Q = S + (int64_t)sb * TN * ldS;
    double* Tb = T + (int64_t)sb * nrows * TN;             // block sb: [nrows][256 SNPs]
    v16i acc[4][2];                                        // acc[d]: digit d of the wave's 32 eigenvectors
    gemm_tile_i8(P, ldV, Q, ldS, 0, nks, lds, acc);
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int snp = wn * 64 + nn * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const long long gi = (long long)acc[0][nn][e] + ((long long)acc[1][nn][e] << 8) +
                             ((long long)acc[2][nn][e] << 16) + ((long long)acc[3][nn][e] << 24);
        Tb[(int64_t)(vt * ROT_TILE + pl) * TN + snp] = (double)gi * ex[pl];
      }
    }
  }
}

// The rotation on the 4-wave job stream of gemm_i8_w4s.h (wave tile: 4 digits x 32 eigenvectors x 128 SNPs; the SNP
// blocks of the workgroup's chunk are the jobs of one pipeline).  Same integers, same stores as rot_gemm_kernel.
template <bool FAST>
__global__ __launch_bounds__(W4_THREADS) void rot_gemm_w4_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Vq, int64_t ldV, int nVT,
    const int2* __restrict__ groups, int nch, int sb_per_chunk, int nks, const double* __restrict__ step,
    double* __restrict__ T, int64_t nrows) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int vt, chunk;
  if (!w4_group_place(groups, blockIdx.x, nVT, nch, vt, chunk)) return;
  const int sb0 = chunk * sb_per_chunk;
  const int sb1 = min(sb0 + sb_per_chunk, nSb);
  if (sb0 >= sb1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  double* ex = (double*)(lds + LDS_BYTES);
  if (threadIdx.x < ROT_TILE) ex[threadIdx.x] = step[vt * ROT_TILE + threadIdx.x];
  __syncthreads();
  const int8_t* P = Vq + (int64_t)vt * TM * ldV;
  w4s_stream(
      sb0, sb1, ldV, ldS, lds, [&](int sb) { return W4Job{P, S + (int64_t)sb * TN * ldS, nks}; }, [](int) {},
      [&](int sb, v16i (&acc)[4][4]) {
        double* Tb = T + (int64_t)sb * nrows * TN;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int snp = wn * 128 + nn * 32 + r;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int pl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            Tb[(int64_t)(vt * ROT_TILE + pl) * TN + snp] =
                digits4_to_f64<FAST>(acc[0][nn][e], acc[1][nn][e], acc[2][nn][e], acc[3][nn][e]) * ex[pl];
          }
        }
      });
}

int run_rotate(mmg_ctx* ctx, const mmg_geno* g, const int8_t* Vq, const double* dstep, int nVT, double* T) {
  const int nSb = (int)(g->Mpad / TN);
  if (nSb == 0 || nVT == 0) return MMG_OK;
  const int rounds = (nVT + 7) / 8;
  int nch = std::max(1, (16 * 256) / (8 * rounds));        // ~16 workgroups per CU over the launch
  nch = std::min(nch, nSb);
  const int per = (nSb + nch - 1) / nch;
  nch = (nSb + per - 1) / per;
  // MMG_ROT_KERNEL=w8: the first-generation 8-wave kernel (A/B runs; same bits)
  static const bool w8 = [] { const char* e = std::getenv("MMG_ROT_KERNEL"); return e && std::string(e) == "w8"; }();
  if (w8) {
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)rot_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     LDS_BYTES + ROT_TILE * 8));
    EvScope ev(ctx, EV_ROT);
    hipLaunchKernelGGL(rot_gemm_kernel, dim3((unsigned)(8 * rounds * nch)), dim3(NTHREADS), LDS_BYTES + ROT_TILE * 8,
                       ctx->stream, g->d, (int64_t)g->Npad, nSb, Vq, (int64_t)g->Npad, nVT, nch, per, g->Npad / BK,
                       dstep, T, (int64_t)nVT * ROT_TILE);
  } else {
    // groups of 4 eigen tiles x 8 SNP chunks per XCD (gemm_i8_w4s.h: w4_group_table)
    int GV = 4;
    if (const char* e = std::getenv("MMG_ROT_GV")) GV = std::max(1, std::min(32, std::atoi(e)));
    int nch4 = std::min(nSb, std::max(32, (16 * 256) / (8 * rounds) / 32 * 32));
    const int per4 = (nSb + nch4 - 1) / nch4;
    nch4 = (nSb + per4 - 1) / per4;
    const std::vector<int2> tab = w4_group_table(rounds, nch4, GV);
    int rct = upload_group_table(ctx, tab);
    if (rct) return rct;
    const bool fast = w4_digits_fast(g->smax, g->Npad) && !std::getenv("MMG_W4_SLOW_EPI");
    const void* fn = fast ? (const void*)rot_gemm_w4_kernel<true> : (const void*)rot_gemm_w4_kernel<false>;
    MMG_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + ROT_TILE * 8));
    EvScope ev(ctx, EV_ROT);
#define MMG_LAUNCH_ROT_W4(F)                                                                                           \
  hipLaunchKernelGGL(rot_gemm_w4_kernel<F>, dim3((unsigned)(256 * tab.size())), dim3(W4_THREADS),                      \
                     LDS_BYTES + ROT_TILE * 8, ctx->stream, g->d, (int64_t)g->Npad, nSb, Vq, (int64_t)g->Npad, nVT,    \
                     ctx->grp_tab, nch4, per4, g->Npad / BK, dstep, T, (int64_t)nVT * ROT_TILE)
    if (fast) MMG_LAUNCH_ROT_W4(true); else MMG_LAUNCH_ROT_W4(false);
#undef MMG_LAUNCH_ROT_W4
  }
  MMG_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

// coef row i: [ d_0 .. d_{PB-1} | omega_0, g_00 .. g_0(Q-1) | omega_1, g_10 .. | ... ]   (PB + PB * (1 + Q) doubles)
// R = SNPs per lane (R workgroup-blocks of 256 SNPs side by side): the wave-uniform coefficients are fetched once per
// R elements -- at R = 1 the scalar cache, not the fp64 VALU or HBM, sets the pace (measured: 3.9 TB/s of T).
template <int PB, int Q, int R, int ABL = 0>   // ABL (timing ablations, wrong results): 1 = no loads of T, 2 = loads + one FMA
__global__ __launch_bounds__(256) void scan_multi_kernel(const double* __restrict__ T, int64_t nrows, int32_t N, int64_t M,
                                                         const double* __restrict__ coef, const double* __restrict__ h0,
                                                         double nu, double lnbeta, double* __restrict__ rss,
                                                         double* __restrict__ Fst, double* __restrict__ pv, int64_t ldOut) {
  constexpr int NL = PB * (1 + Q), NC = PB + NL;
  const int64_t nblk = (M + 255) / 256;
  double aq[R][PB], al[R][NL];
  const double* tp[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
#pragma unroll
    for (int k = 0; k < PB; ++k) aq[rr][k] = 0.0;
#pragma unroll
    for (int k = 0; k < NL; ++k) al[rr][k] = 0.0;
    // a block index beyond the last one re-reads the last block (its results are not written)
    const int64_t blk = min((int64_t)blockIdx.x * R + rr, nblk - 1);
    tp[rr] = T + blk * nrows * 256 + threadIdx.x;
  }
  // The elements of the next DEPTH eigen-coordinates are in flight while the current one is multiplied out (a ring of
  // registers): with loads issued only at the top of an iteration the memory stream and the fp64 FMAs took turns
  // (10.1 ms per pass = the SUM of 6.5 ms loads-only and 5.4 ms FMAs-only) instead of overlapping.
  constexpr int DEPTH = 4;
  auto accumulate = [&](const double (&tau)[R], int i) {
    const double* c = coef + (int64_t)i * NC;                  // wave-uniform: scalar loads
    if (ABL == 2) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) aq[rr][0] = fma(tau[rr], c[0], aq[rr][0]);
      return;
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const double t2 = tau[rr] * tau[rr];
#pragma unroll
      for (int k = 0; k < PB; ++k) aq[rr][k] = fma(t2, c[k], aq[rr][k]);
#pragma unroll
      for (int k = 0; k < NL; ++k) al[rr][k] = fma(tau[rr], c[PB + k], al[rr][k]);
    }
  };
  double ring[DEPTH][R];
  const int last = (int)nrows - 1;                              // rows N .. nrows-1 exist (zeros); never read past them
#pragma unroll
  for (int u = 0; u < DEPTH; ++u)
#pragma unroll
    for (int rr = 0; rr < R; ++rr) ring[u][rr] = ABL == 1 ? 1e-3 * (u + rr) : tp[rr][(int64_t)min(u, last) * 256];
  const int N4 = N / DEPTH * DEPTH;
  for (int i = 0; i < N4; i += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      double tau[R];
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        tau[rr] = ring[u][rr];
        ring[u][rr] = ABL == 1 ? tau[rr] + 1e-9 : tp[rr][(int64_t)min(i + u + DEPTH, last) * 256];
      }
      accumulate(tau, i + u);
    }
  }
#pragma unroll
  for (int u = 0; u < DEPTH; ++u)
    if (N4 + u < N) {
      double tau[R];
#pragma unroll
      for (int rr = 0; rr < R; ++rr) tau[rr] = ring[u][rr];
      accumulate(tau, N4 + u);
    }
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const int64_t m = ((int64_t)blockIdx.x * R + rr) * 256 + threadIdx.x;
    if (m >= M) continue;
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      double den = aq[rr][k];
#pragma unroll
      for (int c = 0; c < Q; ++c) den = fma(-al[rr][k * (1 + Q) + 1 + c], al[rr][k * (1 + Q) + 1 + c], den);
      const double dot = al[rr][k * (1 + Q)];
      const double h = h0[k];
      double r = h;
      // den ~ 0: the SNP is constant after projecting the covariates out; the reference's lstsq returns no residual
      // and rss stays h0_rss (linear_models.py:1308,1329) -- same rule as scan_finalize_kernel
      if (den > 1e-7 * aq[rr][k] && den > 0.0) r = h - dot * dot / den;
      const double F = (h / r - 1.0) * nu;
      if (rss) rss[(int64_t)k * ldOut + m] = r;
      if (Fst) Fst[(int64_t)k * ldOut + m] = F;
      if (pv) pv[(int64_t)k * ldOut + m] = f_sf_1(F, nu, lnbeta);
    }
  }
}

template <int PB, int Q>
static void launch_multi(mmg_ctx* ctx, const double* T, int64_t nrows, int32_t N, int64_t M, const double* coef,
                         const double* h0, int32_t df2, double lnbeta, double* rss, double* F, double* p, int64_t ldOut) {
  constexpr int R = (PB * (2 + Q) <= 24) ? 2 : 1;            // accumulators: 2 R PB (2 + Q) VGPRs
  int rsel = R;
  if (const char* e = std::getenv("MMG_MULTI_R")) rsel = std::atoi(e) == 1 ? 1 : R;
  const int64_t nblk = (M + 255) / 256;
  int abl = 0;
  if (const char* e = std::getenv("MMG_MULTI_ABL")) abl = std::atoi(e);
  if (abl == 1 && PB == 8 && Q == 1)
    hipLaunchKernelGGL((scan_multi_kernel<8, 1, 2, 1>), dim3((unsigned)((nblk + 1) / 2)), dim3(256), 0, ctx->stream, T,
                       nrows, N, M, coef, h0, (double)df2, lnbeta, rss, F, p, ldOut);
  else if (abl == 2 && PB == 8 && Q == 1)
    hipLaunchKernelGGL((scan_multi_kernel<8, 1, 2, 2>), dim3((unsigned)((nblk + 1) / 2)), dim3(256), 0, ctx->stream, T,
                       nrows, N, M, coef, h0, (double)df2, lnbeta, rss, F, p, ldOut);
  else if (rsel == 2)
    hipLaunchKernelGGL((scan_multi_kernel<PB, Q, R>), dim3((unsigned)((nblk + R - 1) / R)), dim3(256), 0, ctx->stream, T,
                       nrows, N, M, coef, h0, (double)df2, lnbeta, rss, F, p, ldOut);
  else
    hipLaunchKernelGGL((scan_multi_kernel<PB, Q, 1>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, T, nrows, N, M,
                       coef, h0, (double)df2, lnbeta, rss, F, p, ldOut);
}

int run_scan_multi(mmg_ctx* ctx, const double* T, int64_t nrows, int32_t N, int64_t M, int PB, int q, const double* coef,
                   const double* h0, int32_t df2, double lnbeta, double* rss, double* F, double* p, int64_t ldOut) {
  if (M == 0) return MMG_OK;
  EvScope ev(ctx, EV_MULTI);
#define MMG_MULTI(PB_, Q_) launch_multi<PB_, Q_>(ctx, T, nrows, N, M, coef, h0, df2, lnbeta, rss, F, p, ldOut)
#define MMG_MULTI_Q(PB_)                                  \
  do {                                                    \
    if (q == 1) MMG_MULTI(PB_, 1);                        \
    else if (q == 2) MMG_MULTI(PB_, 2);                   \
    else if (q == 3) MMG_MULTI(PB_, 3);                   \
    else MMG_MULTI(PB_, 4);                               \
  } while (0)
  if (q < 1 || q > 4) return set_err(ctx, MMG_E_ARG, "multi-phenotype scan: 1 <= q <= 4 fixed-effect columns");
  if (PB == 1) MMG_MULTI_Q(1);
  else if (PB == 2) MMG_MULTI_Q(2);
  else if (PB == 4) MMG_MULTI_Q(4);
  else if (PB == 8) MMG_MULTI_Q(8);
  else return set_err(ctx, MMG_E_ARG, "multi-phenotype scan: batch must be 1, 2, 4 or 8");
#undef MMG_MULTI_Q
#undef MMG_MULTI
  MMG_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

}  // namespace mmg
