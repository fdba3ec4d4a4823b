// gemm_i8_w4s.h -- the hand-laid 4-wave int8 MFMA pipeline of k_scan_w4s.hip as a reusable JOB STREAM:
// one workgroup (4 waves, one per SIMD, wave tile 128 x 128 of a 256 x 256 output tile) runs a list of tile jobs
// (P tile, Q tile, K steps) back to back through ONE software pipeline -- the LDS-DMA prefetch of the next job's
// first stages is in flight while the current job finishes and its epilogue runs, so nothing drains between tiles.
// Users: perm_gemm_w4_kernel (k_perm.hip), rot_gemm_w4_kernel (k_rot.hip), kinship_i8_w4_kernel (k_kinship.hip);
// the EMMAX quadratic form keeps its own specialisation (operand capture for its epilogue) in k_scan_w4s.hip and
// shares the slice primitives below.
//
// Pipeline of a K step (128 bytes of k = 4 slices of 32 bytes; per wave and slice 16 MFMA 32x32x32, 8 ds_read_b128
// into the other half of a register double buffer, and LDS-DMA pieces of the stage two ahead):
//       step t, slice 0:   pieces N3..15 of stage t+1 -> slot (t+1)&1
//       step t, slices 0-2: MFMA on slot t&1, fragments one slice ahead
//       s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier             <- the only barrier of the step
//       step t, slice 3:   MFMA on registers; fragments of step t+1 slice 0 from slot (t+1)&1;
//                          pieces 0..N3-1 of stage t+2 -> slot t&1
// RAW / WAR argument: see k_scan_w4s.hip.  Beyond the end of the stream the cursor re-issues the last stage into
// slots nobody reads; the stream drains vmcnt before it returns.
#pragma once
#include <algorithm>
#include <vector>
#include "gemm_i8_w4.h"

namespace mmg {

struct Frag4 {
  v4i a[4], b[4];
};

__device__ __forceinline__ v16i mfma8(v4i a, v4i b, v16i c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }

// interleave hint for one slice: (MFMA, ds_read, MFMA, [DMA]) x 8
template <int NDMA>
__device__ __forceinline__ void sched_slice() {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (i < NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  }
}

// Fragments are fetched in the order a0 b0 a1 b1 a2 b2 a3 b3 (one per two MFMAs) and the MFMAs of the next
// slice consume them in the order of their arrival (ORD), so every fragment has at least 12 MFMA slots
// (~400 cycles) between its ds_read and its first use.
__device__ constexpr int ORD_M[16] = {0, 1, 0, 1, 2, 2, 0, 1, 2, 3, 3, 3, 0, 1, 2, 3};
__device__ constexpr int ORD_N[16] = {0, 0, 1, 1, 0, 1, 2, 2, 2, 0, 1, 2, 3, 3, 3, 3};

struct W4Job {
  const int8_t* P;     // row 0 of the job's 256-row P tile (k = 0 of the job's contraction range)
  const int8_t* Q;     // row 0 of the job's 256-row Q tile
  int nks;             // K steps of 128 bytes (>= 1)
};

// One slice of the generic stream: 16 MFMA on `cur`; fragment reads of (slot `src`, chunk) into `nxt`; DMA pieces
// [P0, P1) of the cursor's stage (pieces 0-7: P rows, 8-15: Q rows of this wave) into slot `dst`.
template <int P0, int P1, bool ZERO>
__device__ __forceinline__ void w4_slice(v16i (&acc)[4][4], const Frag4& cur, Frag4& nxt, const char* src, int arow,
                                         int brow, int chunk, const StageOp4& sp, const StageOp4& sq, int k0, char* dst,
                                         int wave) {
  static_assert(P1 - P0 <= 8, "at most one DMA piece per MFMA pair");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m0 = ORD_M[2 * i], n0 = ORD_N[2 * i], m1 = ORD_M[2 * i + 1], n1 = ORD_N[2 * i + 1];
    if (ZERO) acc[m0][n0] = mfma8(cur.a[m0], cur.b[n0], v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
    else acc[m0][n0] = mfma8(cur.a[m0], cur.b[n0], acc[m0][n0]);
    if ((i & 1) == 0) nxt.a[i >> 1] = lds_frag(src, arow + (i >> 1) * 32, chunk);
    else nxt.b[i >> 1] = lds_frag(src + TILE_BYTES, brow + (i >> 1) * 32, chunk);
    if (ZERO) acc[m1][n1] = mfma8(cur.a[m1], cur.b[n1], v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
    else acc[m1][n1] = mfma8(cur.a[m1], cur.b[n1], acc[m1][n1]);
    if (P0 + i < P1) {
      const int pc = P0 + i;
      if (pc < 8) stage_piece4(sp, k0, dst, wave, pc);
      else stage_piece4(sq, k0, dst + TILE_BYTES, wave, pc - 8);
    }
  }
  sched_slice<(P1 > P0 ? P1 - P0 : 0)>();
}

// Runs jobs j0 .. j1-1 of this workgroup.  job(j) -> W4Job (wave-uniform); ldP / ldQ: row strides of the operands
// (the same for every job; 256 * ld < 2^31, the 32-bit buffer offsets of the staging loads).  pre(j) is called before the LAST K step of job j (global loads issued there are
// covered by that step's own vmcnt(0) wait -- the place to fetch what the epilogue needs); epi(j, acc) after it,
// with acc[m][n] = the wave's 4 x 4 accumulator tiles (rows wm*128 + m*32.., columns wn*128 + n*32..; wm = wave >> 1,
// wn = wave & 1; C layout of gemm_i8_core.h).  epi must not touch lds[0, LDS_BYTES).
template <class JobFn, class PreFn, class EpiFn>
__device__ __forceinline__ void w4s_stream(int j0, int j1, int64_t ldP, int64_t ldQ, char* lds, JobFn&& job, PreFn&& pre,
                                           EpiFn&& epi) {
  constexpr int N3 = 8;                                  // pieces issued right after the barrier (slice 3)
  if (j1 <= j0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int arow = wm * 128 + r, brow = wn * 128 + r;

  // ---- issue cursor over the flattened stage stream (wave-uniform scalars)
  int cj = j0;
  W4Job cjb = job(cj);
  int cks = 0, cnks = cjb.nks;
  StageOp4 sp = make_stage_op4(cjb.P, ldP, wave, lane);
  StageOp4 sq = make_stage_op4(cjb.Q, ldQ, wave, lane);
  auto advance = [&]() {
    if (cks + 1 < cnks) { ++cks; return; }
    if (cj + 1 < j1) {
      ++cj;
      cjb = job(cj);
      cks = 0;
      cnks = cjb.nks;
      sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)cjb.P, 0, 0x7fffffff, 0x00020000);
      sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)cjb.Q, 0, 0x7fffffff, 0x00020000);
    }                                                    // else: stay on the last stage (harmless re-issue)
  };

  // ---- prologue: stage 0 complete, the first N3 pieces of stage 1 in flight, fragments of step 0 slice 0
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece4(sp, 0, lds, wave, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece4(sq, 0, lds + TILE_BYTES, wave, i);
  advance();                                             // -> stage 1
#pragma unroll
  for (int i = 0; i < N3; ++i) stage_piece4(sp, cks * BK, lds + BUF_BYTES, wave, i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N3) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  Frag4 f0, f1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f0.a[i] = lds_frag(lds, arow + i * 32, h);
    f0.b[i] = lds_frag(lds + TILE_BYTES, brow + i * 32, h);
  }

  v16i acc[4][4];                                        // written (not accumulated) by the first slice of every job
  int t = 0;
  auto step = [&](bool first) {
    char* cur = lds + (t & 1) * BUF_BYTES;
    char* oth = lds + ((t + 1) & 1) * BUF_BYTES;
    const int k1 = cks * BK;
    if (first) w4_slice<N3, 16, true>(acc, f0, f1, cur, arow, brow, 2 + h, sp, sq, k1, oth, wave);
    else w4_slice<N3, 16, false>(acc, f0, f1, cur, arow, brow, 2 + h, sp, sq, k1, oth, wave);
    w4_slice<16, 16, false>(acc, f1, f0, cur, arow, brow, 4 + h, sp, sq, k1, oth, wave);
    w4_slice<16, 16, false>(acc, f0, f1, cur, arow, brow, 6 + h, sp, sq, k1, oth, wave);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    advance();                                           // -> stage t+2
    w4_slice<0, N3, false>(acc, f1, f0, oth, arow, brow, h, sp, sq, cks * BK, cur, wave);
    ++t;
  };
  for (int jj = j0; jj < j1; ++jj) {
    const int nks = job(jj).nks;
    for (int ks = 0; ks < nks - 1; ++ks) step(ks == 0);
    pre(jj);
    step(nks == 1);
    epi(jj, acc);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-issued tail stages must land before LDS is released
}


// Digit rows of the permutation / rotation GEMM operands (k_perm.hip:perm_quantize_kernel): a row's entries are
// rounded to Z in [-(2^27 - 1), 2^27 - 1], shifted by ROWS_OFFSET = 2^27 into the non-negative range and written as four
// UNSIGNED 7-bit digits (round 2: four balanced base-256 digits; non-negative bytes run 6-9 % faster on this
// power-capped part, tools/probe/mfma_digit_range.hip, at 3 bits of the operand).  The GEMM then accumulates
// sum_k s_k (Z_k + 2^27) = sum_k s_k Z_k + 2^27 sum(s): the epilogue takes the second term out with the SNP's exact
// genotype sum.
constexpr int ROWS_DIGIT_BITS = 7;
constexpr long long ROWS_OFFSET = 1ll << 27;
constexpr double ROWS_ZMAX = 134217727.0;                // 2^27 - 1

// Four unsigned 7-bit digit accumulators and the SNP's genotype sum -> the exact integer
// a0 + 2^7 a1 + 2^14 a2 + 2^21 a3 - 2^27 ssum = sum_k s_k Z_k as a double.
// FAST (host-checked: every |a_d| * 257 < 2^31, i.e. Npad * 128 * max|s| * 257 < 2^31): two 32-bit digit pairs -- the
// offset leaves with the upper pair as (ssum << 13) -- and one exact fp64 FMA instead of the 64-bit shift/add chain and
// its int64 -> double conversion.  Both forms are exact (|sum| < 2^53), hence bit-identical.
template <bool FAST>
__device__ __forceinline__ double digits4_to_f64(int a0, int a1, int a2, int a3, int ssum) {
  if (FAST) {
    const int lo = a0 + a1 * 128, hi = a2 + a3 * 128 - (ssum << 13);
    return fma((double)hi, 16384.0, (double)lo);
  }
  const long long gi = (long long)a0 + ((long long)a1 << 7) + ((long long)a2 << 14) + ((long long)a3 << 21) -
                       ((long long)ssum << 27);
  return (double)gi;
}

inline bool w4_digits_fast(int64_t smax, int64_t Npad) { return Npad * 128 * std::max<int64_t>(smax, 1) * 257 < (int64_t(1) << 31); }

// ---- XCD-aware placement of (operand tile, SNP chunk) pairs for the GEMMs that stream the genotype store against a
// set of 256-row operand tiles (permutation columns, eigenvectors).  Workgroup b runs on XCD b & 7 (round-robin
// placement) and, at one workgroup per CU, the 32 workgroups i = b >> 3 of a GROUP (i >> 5) are resident on that XCD
// together.  A group is gv operand tiles (tile slots slot0 .. slot0+gv-1 of this XCD; tile = x + 8 * slot) times
// 32 / gv SNP chunks: the workgroups of a group walk their chunks in step, so that every genotype block is fetched
// into the XCD's L2 once per gv tiles and every operand tile once per 32 / gv chunks -- per 32 tile products
// gv + 32/gv tile streams instead of the 1 + 32 of one tile per XCD (first version: the whole store crossed the
// fabric once per operand tile, 400 GB per rotation at N = 5000, M = 1e6).
// Table entry: x = slot0 | gv << 16, y = first chunk.
inline std::vector<int2> w4_group_table(int rounds, int nch, int GV) {
  std::vector<int2> t;
  for (int s0 = 0; s0 < rounds;) {
    int gv = 1;
    while (gv * 2 <= std::min(GV, rounds - s0)) gv *= 2;
    const int cpg = 32 / gv;
    for (int c0 = 0; c0 < nch; c0 += cpg) t.push_back(int2{s0 | (gv << 16), c0});
    s0 += gv;
  }
  return t;
}

// tile / chunk of workgroup b; false: nothing to do
__device__ __forceinline__ bool w4_group_place(const int2* __restrict__ tab, int b, int ntiles, int nch, int& tile,
                                               int& chunk) {
  const int x = b & 7, i = b >> 3;
  const int2 ge = tab[i >> 5];
  const int within = i & 31, gv = ge.x >> 16, slot0 = ge.x & 0xffff;
  tile = x + 8 * (slot0 + (within & (gv - 1)));
  chunk = ge.y + within / gv;
  return tile < ntiles && chunk < nch;
}

}  // namespace mmg
