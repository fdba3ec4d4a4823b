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
// so the O(N^2) work per SNP is done ONCE (T = S U', an exact int8-MFMA digit GEMM, 4 unsigned 7-bit digits per
// eigenvector (entries shifted into the non-negative range, gemm_i8_w4s.h ROWS_OFFSET) with a per-eigenvector step -- the operand layout and mainloop of the permutation GEMM, k_perm.hip),
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
    int sb_per_chunk, int nks, const double* __restrict__ step, const double* __restrict__ ssum, double* __restrict__ T,
    int64_t nrows) {
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
      const int ss = (int)ssum[(int64_t)sb * TN + snp];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        Tb[(int64_t)(vt * ROT_TILE + pl) * TN + snp] =
            digits4_to_f64<false>(acc[0][nn][e], acc[1][nn][e], acc[2][nn][e], acc[3][nn][e], ss) * ex[pl];
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
    const double* __restrict__ ssum, double* __restrict__ T, int64_t nrows) {
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
  int ss[4];
  w4s_stream(
      sb0, sb1, ldV, ldS, lds, [&](int sb) { return W4Job{P, S + (int64_t)sb * TN * ldS, nks}; },
      [&](int sb) {
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) ss[nn] = (int)ssum[(int64_t)sb * TN + wn * 128 + nn * 32 + r];
      },
      [&](int sb, v16i (&acc)[4][4]) {
        double* Tb = T + (int64_t)sb * nrows * TN;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int snp = wn * 128 + nn * 32 + r;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int pl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            Tb[(int64_t)(vt * ROT_TILE + pl) * TN + snp] =
                digits4_to_f64<FAST>(acc[0][nn][e], acc[1][nn][e], acc[2][nn][e], acc[3][nn][e], ss[nn]) * ex[pl];
          }
        }
      });
}

int run_rotate(mmg_ctx* ctx, const mmg_geno* g, const int8_t* Vq, const double* dstep, const double* d_ssum, int nVT,
               double* T) {
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
                       dstep, d_ssum, T, (int64_t)nVT * ROT_TILE);
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
                     ctx->grp_tab, nch4, per4, g->Npad / BK, dstep, d_ssum, T, (int64_t)nVT * ROT_TILE)
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

// ---- second generation of the pass: every column on the fp64 MATRIX pipe, operands streamed by LDS-DMA.
// Per rotated coordinate a phenotype needs d * tau^2 (one column) and omega * tau, G_c * tau (1 + Q columns).  With
// SGPR coefficients on the VALU (scan_multi_kernel) a batch of 8 phenotypes is VALU-bound, and the scalar loads
// (2 + Q doubles per phenotype and coordinate) cannot be hidden at all once fewer FMAs stand behind each of them
// (measured: the 17 VALU operations per element of a 16-phenotype quadratic part alone took 14.5 ms per pass).
// On v_mfma_f64_16x16x4_f64 the coefficients are an operand instead: A = 16 coefficient columns x 4 coordinates
// (lane 16 k + c: column c of coordinate i0 + k), B = tau (lane 16 k + r: coordinate i0 + k of one SNP of the wave's
// 64; the squares for the quadratic tile are the only VALU work left, one multiply per element).  Tile 0 = the PB
// quadratic columns against tau^2, tiles 1.. = the PB (1 + Q) linear columns against tau.
// Both operands come through LDS: a GROUP is 4 coordinates = 2 KB of T (4 rows x 64 SNPs of this wave) + the 4
// coefficient rows (contiguous in the table), fetched by 2 + NCI buffer_load ... lds instructions into a per-wave ring
// of DEPTH groups.  No register is the destination of a global load, so the only vmcnt waits are the ones written
// here (a first version with register loads and a register ring spent as long waiting as computing: the compiler
// placed one vmcnt for all groups in flight at the top of the unrolled body, 14.7 ms = 9.4 ms MFMA + 5.3 ms exposed
// latency).  Nothing is shared between waves: no barrier in the loop.
// LDS image of a group: tau row k at k * 512 (SNP pair p at + 16 p), coefficient row k at 2048 + k * NC * 8.
// Fragment reads are conflict-free: a 16-lane ds_read_b128 group reads 256 contiguous bytes of one tau row; the two
// coefficient rows of a 32-lane ds_read_b64 group lie 384 (mod 256: 128) bytes apart for NC = 48.
// SNP of MFMA column r in SNP group g = (h, e): 32 h + 2 r + e (the two doubles of one 16-byte read feed two groups).
// D layout of v_mfma_f64_16x16x4_f64 (probed, tools/probe/mfma_f64_layout.hip): lane l, register r holds row
// 4 r + l / 16, column l % 16.
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <int PB, int Q>
struct MultiMfmaCfg {
  static constexpr int NL = PB * (1 + Q), NC = PB + NL, NTL = (NL + 15) / 16, NT = 1 + NTL;
  static constexpr int NCI = (32 * NC + 1023) / 1024;            // DMA instructions per coefficient group
  static constexpr int GROUP_BYTES = 2048 + 1024 * NCI;
#ifndef MMG_MULTI_DEPTH
#define MMG_MULTI_DEPTH 4
#endif
  static constexpr int DEPTH = GROUP_BYTES <= 4096 ? MMG_MULTI_DEPTH : 4;      // groups in flight per wave
  static constexpr int WAVE_BYTES = DEPTH * GROUP_BYTES;
  static constexpr int LDS_BYTES = 4 * WAVE_BYTES > 32768 ? 4 * WAVE_BYTES : 32768;
};

template <int PB, int Q, int ABL = 0>   // ABL (timing ablations, wrong results): 1 = no MFMA, 2 = no DMA in the loop
__global__ __launch_bounds__(256) void scan_multi_mfma_kernel(const double* __restrict__ T, int64_t nrows, int32_t N,
                                                              int64_t M, const double* __restrict__ coef,
                                                              const double* __restrict__ h0, double nu, double lnbeta,
                                                              double* __restrict__ rss, double* __restrict__ Fst,
                                                              double* __restrict__ pv, int64_t ldOut) {
  using Cfg = MultiMfmaCfg<PB, Q>;
  constexpr int NL = Cfg::NL, NC = Cfg::NC, NTL = Cfg::NTL, NT = Cfg::NT, NCI = Cfg::NCI, GB = Cfg::GROUP_BYTES,
                DEPTH = Cfg::DEPTH;
  static_assert(PB <= 16, "one quadratic tile");
  static_assert((2 + NCI) * DEPTH < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char mlds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kq = lane >> 4, r16 = lane & 15;
  char* ring = mlds + wave * Cfg::WAVE_BYTES;
  // out-of-range rows (coordinates beyond nrows, coefficient rows beyond N) read as zeros: raw buffers with exact sizes
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(T + (int64_t)blockIdx.x * nrows * 256), 0, (int)(nrows * 2048), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)coef, 0, N * NC * 8, 0x00020000);
  const int vT = (lane >> 5) * 2048 + wave * 512 + (lane & 31) * 16;
  const int vC = lane * 16;
  auto issue = [&](int grp, int slot) {
    char* dst = ring + slot * GB;
#pragma unroll
    for (int qd = 0; qd < 2; ++qd)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (MMG_AS3 void*)(dst + qd * 1024), 16, vT, (4 * grp + 2 * qd) * 2048, 0, 0);
#pragma unroll
    for (int qd = 0; qd < NCI; ++qd)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsC, (MMG_AS3 void*)(dst + 2048 + qd * 1024), 16, vC,
                                               4 * grp * NC * 8 + qd * 1024, 0, 0);
  };
  v4d acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[t][g] = v4d{0.0, 0.0, 0.0, 0.0};
  double sink = 0.0;                                             // ABL 1 only
  static_assert(DEPTH % 2 == 0, "two fragment register sets alternate over an unrolled ring turn");
  const int NG = ((N + 3) / 4 + DEPTH - 1) / DEPTH * DEPTH;      // whole ring turns; groups beyond N are zeros
  struct Frag { v2d tau[2]; double a[NT]; };
  auto read_frag = [&](Frag& f, int slot) {
    const char* src = ring + slot * GB;
#pragma unroll
    for (int h = 0; h < 2; ++h) f.tau[h] = *(const v2d*)(src + kq * 512 + (r16 + 16 * h) * 16);
    f.a[0] = *(const double*)(src + 2048 + (kq * NC + r16) * 8);
#pragma unroll
    for (int t = 0; t < NTL; ++t) f.a[1 + t] = *(const double*)(src + 2048 + (kq * NC + PB + 16 * t + r16) * 8);
  };
  // Schedule of step s (group s; fragments one group ahead, DMA DEPTH groups ahead):
  //   lgkmcnt(0)                        fragments of group s are in registers -> its slot is free
  //   DMA of group s + DEPTH            into that slot
  //   vmcnt((2 + NCI)(DEPTH - 1))       group s + 1 (issued DEPTH - 1 steps ago) has landed
  //   fragment reads of group s + 1     (overlap the MFMAs below)
  //   4 multiplies + 4 NT MFMAs of group s
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) issue(u, u);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((2 + NCI) * (DEPTH - 1)) : "memory");
  Frag fr[2];
  read_frag(fr[0], 0);
  for (int gi = 0; gi < NG; gi += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      Frag& cur = fr[u & 1];
      Frag& nxt = fr[(u + 1) & 1];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ABL != 2) {
        issue(gi + u + DEPTH, u);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((2 + NCI) * (DEPTH - 1)) : "memory");
      }
      read_frag(nxt, (u + 1) % DEPTH);
      if (ABL != 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double x = cur.tau[g >> 1][g & 1];
          acc[0][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a[0], x * x, acc[0][g], 0, 0, 0);
#pragma unroll
          for (int t = 1; t < NT; ++t) acc[t][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a[t], x, acc[t][g], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) sink = fma(cur.tau[g >> 1][g & 1], cur.a[0] + cur.a[NT - 1], sink);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (ABL == 1) acc[0][0][0] = sink;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the groups issued past the end must land before the ring is reused
  // ---- matrix-pipe layout -> lane = SNP, one 16-column tile at a time through LDS (the ring's memory)
  double (*xl)[16][64] = (double (*)[16][64])mlds;               // [wave][column][SNP]
  double aq[PB], al[NL];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) xl[wave][4 * r + kq][32 * (g >> 1) + 2 * r16 + (g & 1)] = acc[t][g][r];
    __syncthreads();
    if (t == 0) {
#pragma unroll
      for (int c = 0; c < PB; ++c) aq[c] = xl[wave][c][lane];
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (16 * (t - 1) + c < NL) al[16 * (t - 1) + c] = xl[wave][c][lane];
    }
  }
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
#pragma unroll
  for (int k = 0; k < PB; ++k) {
    double den = aq[k];
#pragma unroll
    for (int c = 0; c < Q; ++c) den = fma(-al[k * (1 + Q) + 1 + c], al[k * (1 + Q) + 1 + c], den);
    const double dot = al[k * (1 + Q)];
    const double h = h0[k];
    double r = h;
    if (den > 1e-7 * aq[k] && den > 0.0) r = h - dot * dot / den;      // same rule as scan_multi_kernel
    const double F = (h / r - 1.0) * nu;
    if (rss) rss[(int64_t)k * ldOut + m] = r;
    if (Fst) Fst[(int64_t)k * ldOut + m] = F;
    if (pv) pv[(int64_t)k * ldOut + m] = f_sf_1(F, nu, lnbeta);
  }
}

template <int PB, int Q>
static void launch_multi(mmg_ctx* ctx, const double* T, int64_t nrows, int32_t N, int64_t M, const double* coef,
                         const double* h0, int32_t df2, double lnbeta, double* rss, double* F, double* p, int64_t ldOut) {
  // MMG_MULTI_KERNEL=valu: the first-generation kernel (all columns on the VALU); default for batches of 8 and 16:
  // linear columns on the fp64 matrix pipe
  static const bool valu_only = [] { const char* e = std::getenv("MMG_MULTI_KERNEL"); return e && std::string(e) == "valu"; }();
  if constexpr (PB >= 8) if (PB == 16 || !valu_only || Q > 4) {
    const int64_t nb = (M + 255) / 256;
    int ab = 0;
#ifdef MMG_EXPERIMENTS
    if (const char* e = std::getenv("MMG_MULTI_ABL")) ab = std::atoi(e);     // timing ablations (wrong results): `make EXPERIMENTS=1` only
#endif
    constexpr int LB = MultiMfmaCfg<PB, Q>::LDS_BYTES;
#define MMG_LAUNCH_MFMA(ABL_)                                                                                         \
  do {                                                                                                                \
    hipFuncSetAttribute((const void*)scan_multi_mfma_kernel<PB, Q, ABL_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                        LB);                                                                                          \
    hipLaunchKernelGGL((scan_multi_mfma_kernel<PB, Q, ABL_>), dim3((unsigned)nb), dim3(256), LB, ctx->stream, T,      \
                       nrows, N, M, coef, h0, (double)df2, lnbeta, rss, F, p, ldOut);                                 \
  } while (0)
    if constexpr (Q == 1) {
      if (ab == 1) { MMG_LAUNCH_MFMA(1); return; }
      if (ab == 2) { MMG_LAUNCH_MFMA(2); return; }
    }
    MMG_LAUNCH_MFMA(0);
#undef MMG_LAUNCH_MFMA
    return;
  }
  if constexpr (PB <= 8 && Q <= 4) {
  constexpr int R = (PB * (2 + Q) <= 24) ? 2 : 1;            // accumulators: 2 R PB (2 + Q) VGPRs
  int rsel = R;
  if (const char* e = std::getenv("MMG_MULTI_R")) rsel = std::atoi(e) == 1 ? 1 : R;
  const int64_t nblk = (M + 255) / 256;
  int abl = 0;
#ifdef MMG_EXPERIMENTS
  if (const char* e = std::getenv("MMG_MULTI_ABL")) abl = std::atoi(e);
#endif
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
  if (q < 1 || q > 8) return set_err(ctx, MMG_E_ARG, "multi-phenotype scan: 1 <= q <= 8 fixed-effect columns");
  if (q > 4) {
    // wider models (round 4: more than 3 cofactors): 8 phenotypes per pass on the matrix-pipe kernel, whose tile count
    // follows the column count (1 + ceil(8 (1 + q) / 16) tiles: 5 for q = 5, 6 for q = 8); smaller batches are padded
    if (PB != 8) return set_err(ctx, MMG_E_ARG, "multi-phenotype scan: q > 4 runs in batches of 8");
    if (q == 5) MMG_MULTI(8, 5); else if (q == 6) MMG_MULTI(8, 6); else if (q == 7) MMG_MULTI(8, 7); else MMG_MULTI(8, 8);
  }
  else if (PB == 1) MMG_MULTI_Q(1);
  else if (PB == 2) MMG_MULTI_Q(2);
  else if (PB == 4) MMG_MULTI_Q(4);
  else if (PB == 8) MMG_MULTI_Q(8);
  else if (PB == 16 && q <= 2) {                              // 1 + 2 / 1 + 3 tiles; wider models go 8 at a time
    if (q == 1) MMG_MULTI(16, 1); else MMG_MULTI(16, 2);
  }
  else return set_err(ctx, MMG_E_ARG, "multi-phenotype scan: batch must be 1, 2, 4, 8 or (q <= 2) 16");
#undef MMG_MULTI_Q
#undef MMG_MULTI
  MMG_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

}  // namespace mmg
