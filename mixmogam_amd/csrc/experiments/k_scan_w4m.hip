// k_scan_w4m.hip -- the 4-wave software-pipelined quadratic-form GEMM of k_scan_w4s.hip on the OTHER int8 MFMA
// shape, v_mfma_i32_16x16x64_i8.
//
// Why: the chip is power-limited under this kernel, and the clock it holds depends on the MFMA
// shape.  tools/probes/mfma_shape_probe.hip (bare MFMA loops on random operands, one wave per SIMD, the same
// 128x128 tile per wave): 32x32x32 holds 2.07 GHz = 4.23 POP/s, 16x16x64 holds 2.22 GHz = 4.50 POP/s at equal
// cycles per MAC (on zeros both hold 2.38 GHz).  Same LDS image, same LDS bytes per MAC, same 256 accumulator
// registers; only the fragment / accumulator layouts and the epilogue change.
//
// Outcome (opt-in, MMG_SCAN_KERNEL=w4m; bit-identical): the kernel does hold a higher clock (2.14 vs 1.89 GHz) but
// needs 27 % more cycles -- a 16-cycle MFMA leaves half the issue slack of a 32-cycle one for the same ds_read /
// LDS-DMA / s_waitcnt stream of a single wave per SIMD -- and ends 13 % slower than k_scan_w4s.hip (16.0 vs 14.1 ms
// at M = 400k).  Kept for the record and for A/B runs.
//
//   wave tile 128 x 128 = 8 x 8 tiles of 16 x 16, acc[m][n] 4 registers: lane l holds column n*16 + (l & 15) and
//   rows m*16 + 4*(l >> 4) + e.  A/B fragment of a 64-byte k slice: lane l holds row (l & 15), bytes 16*(l >> 4) .. +16.
//   K step = 2 k slices = 4 quarters of 32 MFMA (16 cycles each):
//       Q0: slice 0, m 0-3      Q1: slice 0, m 4-7      Q2: slice 1, m 0-3      Q3: slice 1, m 4-7
//   registers: ONE set of 8 B fragments, replaced in place (MFMAs run n-major, so b[n] is dead after its fourth
//   MFMA of a slice's second quarter and its successor is read right there: 29 MFMA slots until its first use);
//   A fragments of the current and the next quarter (2 x 4).
//   Stage pipeline, barrier before the last quarter, flattened job stream, captured epilogue operands: exactly
//   as in k_scan_w4s.hip (its header has the RAW / WAR argument); the accumulators are cleared in the epilogue.
// Results are bit-identical to the other generations.
#include <algorithm>
#include <cstdlib>
#include <string>
#include "gemm_i8_core.h"
#include "gemm_i8_w4.h"
#include "mmg_internal.h"

namespace mmg {

// The MFMA is issued through inline asm with the accumulator tied to an AGPR quad ("+a"): with the builtin and 64
// separate 4-register accumulators the register allocator parks tiles in VGPRs and rotates them through a staging
// AGPR quad around every MFMA (8 v_accvgpr moves per MFMA; 22 ms instead of 14).  Consequences handled here:
// the scheduler does not know these are MFMAs, so the interleave is pinned with sched_barrier between groups, and
// the MFMA -> VALU read hazard before the epilogue is covered by explicit s_nops (epilogue_fence).
__device__ __forceinline__ void mfma16(v4i& c, const v4i& a, const v4i& b) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void epilogue_fence() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);          // the accumulator reads must not be scheduled above the nops
}

// One quarter: 32 MFMA (m in [4*MH, 4*MH+4) x n in 0..7, n-major) on A fragments `a` and B fragments `b`.
//   * 4 ds_read_b128 of the next quarter's A fragments into `na` (rows arow + (4*NMH + i)*16 of tile `src`, chunk
//     nchunk), after MFMAs 1, 5, 9, 13;
//   * LOADB: the B fragments are replaced IN PLACE for the next slice: b[n] is dead after its fourth MFMA
//     (i = 4n + 3), which is where its successor (rows brow + n*16 of the Q tile of `src`, chunk nchunk) is read --
//     every B fragment then has 29 MFMA slots (464 cycles) until its first use, and one B register set suffices;
//   * DMA pieces [P0, P1) after MFMAs 2, 6, 10, ...
template <int MH, int NMH, bool LOADB, int P0, int P1>
__device__ __forceinline__ void quarter(v4i (&acc)[8][8], const v4i (&a)[4], v4i (&b)[8], v4i (&na)[4], const char* src,
                                        int arow, int brow, int nchunk, const StageOp4& sp, const StageOp4& sq, int k0,
                                        char* dst, int wave) {
  static_assert(P1 - P0 <= 8, "DMA pieces per quarter");
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int n = i >> 2, m = i & 3;
    mfma16(acc[4 * MH + m][n], a[m], b[n]);
    if ((i & 3) == 1 && i < 16) na[i >> 2] = lds_frag(src, arow + (4 * NMH + (i >> 2)) * 16, nchunk);
    if (LOADB && (i & 3) == 3) b[n] = lds_frag(src + TILE_BYTES, brow + n * 16, nchunk);
    if ((i & 3) == 2 && P0 + (i >> 2) < P1) {
      const int pc = P0 + (i >> 2);
      if (pc < 8) stage_piece4(sp, k0, dst, wave, pc);
      else stage_piece4(sq, k0, dst + TILE_BYTES, wave, pc - 8);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// FAST: see k_scan_w4s.hip (24-bit accumulators, 32-bit per-lane partial sums proven on the host).
template <bool FAST>
__global__ __launch_bounds__(W4_THREADS) void scan_quad_w4m_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
    int64_t digit_stride, const int* __restrict__ job_off, const int2* __restrict__ jobs, int AS,
    unsigned long long* __restrict__ q) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = blockIdx.x;
  const int x = b & 7, bi = b >> 3;
  const int cohort = bi >> 5, within = bi & 31;
  const int a_ = within % AS, grp = within / AS;
  const int sb = (cohort * 8 + x) * AS + a_;
  if (sb >= nSb) return;
  const int j0 = job_off[grp], j1 = job_off[grp + 1];
  if (j1 <= j0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1, r = lane & 15, g = lane >> 4;
  const int8_t* Q = S + (int64_t)sb * TN * ldS;
  const int arow = wm * 128 + r, brow = wn * 128 + r;

  // ---- issue cursor over the flattened stage stream (wave-uniform scalars)
  int cj = j0;
  int2 cjb = jobs[cj];
  int cks = 0, cnks = 2 * (cjb.y + 1);
  StageOp4 sp = make_stage_op4(Bq + (int64_t)cjb.x * digit_stride + (int64_t)cjb.y * TM * ldB, ldB, wave, lane);
  const StageOp4 sq = make_stage_op4(Q, ldS, wave, lane);
  auto advance = [&]() {
    if (cks + 1 < cnks) { ++cks; return; }
    if (cj + 1 < j1) {
      ++cj;
      cjb = jobs[cj];
      cks = 0;
      cnks = 2 * (cjb.y + 1);
      sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Bq + (int64_t)cjb.x * digit_stride + (int64_t)cjb.y * TM * ldB),
                                                0, 0x7fffffff, 0x00020000);
    }                                                    // else: stay on the last stage (harmless re-issue)
  };

  // ---- prologue: stage 0 complete, the P half of stage 1 in flight, fragments of step 0 quarter 0
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece4(sp, 0, lds, wave, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece4(sq, 0, lds + TILE_BYTES, wave, i);
  advance();                                             // -> stage 1
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece4(sp, cks * BK, lds + BUF_BYTES, wave, i);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  v4i a0[4], a1[4], bf[8];                               // A: current / next quarter; B: one set, replaced in place
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = lds_frag(lds, arow + i * 16, g);
#pragma unroll
  for (int i = 0; i < 8; ++i) bf[i] = lds_frag(lds + TILE_BYTES, brow + i * 16, g);

  v4i acc[8][8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = v4i{0, 0, 0, 0};
  unsigned long long qacc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  int cap[8][8];                                         // [n][m]: genotype bytes of rows m*16 + 4g .. +3

  int t = 0;
  // capture: dword g of 16-byte chunk m of this wave's 8 x 16 SNP rows, from the Q tile of the step about to run
  // (conflict-free ds_read_b32: 16 rows x 4 consecutive dwords under the chunk swizzle cover all 64 banks)
  auto capture = [&]() {
    const char* qt = lds + (t & 1) * BUF_BYTES + TILE_BYTES;
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int row = brow + n * 16;
        cap[n][m] = *(const int*)(qt + row * BK + ((m ^ ((row >> 1) & 7)) << 4) + 4 * g);
      }
  };
  auto step = [&](int ks) {
    char* cur = lds + (t & 1) * BUF_BYTES;
    char* oth = lds + ((t + 1) & 1) * BUF_BYTES;
    const int k1 = cks * BK;                             // the cursor's stage (t+1): its Q half goes out in Q0
    // Q0: slice 0, m 0-3; fetch A(m 4-7, slice 0); Q half of stage t+1 -> other slot
    quarter<0, 1, false, 8, 16>(acc, a0, bf, a1, cur, arow, brow, g, sp, sq, k1, oth, wave);
    // Q1: slice 0, m 4-7; fetch A(m 0-3, slice 1); B -> slice 1 in place
    quarter<1, 0, true, 0, 0>(acc, a1, bf, a0, cur, arow, brow, 4 + g, sp, sq, 0, oth, wave);
    // Q2: slice 1, m 0-3; fetch A(m 4-7, slice 1)
    quarter<0, 1, false, 0, 0>(acc, a0, bf, a1, cur, arow, brow, 4 + g, sp, sq, 0, oth, wave);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    advance();                                           // -> stage t+2
    // Q3: slice 1, m 4-7 on registers; fetch A(m 0-3) and B of step t+1 slice 0; P half of stage t+2 -> the slot just retired
    quarter<1, 0, true, 0, 8>(acc, a1, bf, a0, oth, arow, brow, g, sp, sq, cks * BK, cur, wave);
    ++t;
  };
  for (int jj = j0; jj < j1; ++jj) {
    const int2 jb = jobs[jj];
    const int d = jb.x, nks = 2 * (jb.y + 1);
    // ONE copy of the step body per kernel: with several, the accumulator tiles get different homes in each copy
    // and the compiler shuffles all 256 registers between them (and, not knowing that the asm statements are
    // MFMAs, without the wait states those moves need).  Step nks-2 holds the epilogue operands of row half
    // wm = 0, step nks-1 those of wm = 1.
    for (int ks = 0; ks < nks; ++ks) {
      if (ks == nks - 2 + wm) capture();
      step(ks);
    }
    epilogue_fence();
    // ---- epilogue of job jj: qacc[n] += (sum_j T[j][snp] * s[snp][256J + j]) << 8d
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      long long part = 0;
      if (FAST) {
        int p32 = 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int wd = cap[n][m];
#pragma unroll
          for (int e = 0; e < 4; ++e) p32 += __mul24(acc[m][n][e], (int)(int8_t)((wd >> (8 * e)) & 0xff));
        }
        part = p32;
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int wd = cap[n][m];
#pragma unroll
          for (int e = 0; e < 4; ++e) part += (long long)acc[m][n][e] * (long long)(int)(int8_t)((wd >> (8 * e)) & 0xff);
        }
      }
      qacc[n] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m][n] = v4i{0, 0, 0, 0};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-issued tail stages must land before LDS is released
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    unsigned long long v = qacc[n];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (g == 0) atomicAdd(q + (int64_t)sb * TN + wn * 128 + n * 16 + r, v);
  }
}

void launch_scan_quad_w4m(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, unsigned long long* q) {
  const int nSb = (int)(g->Mpad / TN);
  const int per = 8 * md.AS;
  const int ncoh = (nSb + per - 1) / per;
  const int64_t smax = g->smax;
  const bool fast = smax * md.Npad < (1 << 16) && smax * smax * md.Npad < (1 << 18) && !std::getenv("MMG_W4S_SLOW_EPI");
#define MMG_LAUNCH_W4M(F)                                                                                              \
  do {                                                                                                                 \
    hipFuncSetAttribute((const void*)scan_quad_w4m_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL(scan_quad_w4m_kernel<F>, dim3((unsigned)(ncoh * 256)), dim3(W4_THREADS), LDS_BYTES, ctx->stream, \
                       g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad, md.job_off,  \
                       md.jobs, md.AS, q);                                                                             \
  } while (0)
  if (fast) MMG_LAUNCH_W4M(true);
  else MMG_LAUNCH_W4M(false);
#undef MMG_LAUNCH_W4M
}

}  // namespace mmg
