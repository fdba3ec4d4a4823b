// k_scan_w4b.hip -- the 4-wave software-pipelined quadratic-form GEMM of k_scan_w4s.hip for BINARY genotype
// stores (0/1: the reference's 'binary' format and simulations.py data), with the SNP operand staged as bits.
//
//   stage        = digit tile P (256 rows x 128 B = 32 KiB) + genotype tile Qb (256 SNP rows x 16 B = 4 KiB):
//                  36 KiB instead of 64 KiB through L2, the LDS-DMA path and the LDS write port
//   LDS          4 slots x 36 KiB = 144 KiB; stage u lives in slot u & 3 and is issued TWO steps ahead
//                  (9 DMA pieces per wave and stage: 8 x P, 1 x Qb), 3 pieces per slice in slices 0-2
//   B fragments  a lane rebuilds its 16 int8 of a slice from 16 bits with 4 x {bfe, mul, and}
//                  ((nibble * 0x00204081) & 0x01010101) one slice ahead, in the shadow of the MFMAs
//   LDS reads    per wave and K step: 16 ds_read_b128 (A fragments) + 4 ds_read_b128 (row bits) instead of 32
//   step t       slices 0-2: MFMA on slot t&3, A fragments one slice ahead, stage t+2 -> slot (t+2)&3
//                s_waitcnt vmcnt(9) lgkmcnt(0) ; s_barrier          (stage t+1 landed on every wave)
//                slice 3: MFMA on registers; row bits + first A fragments of step t+1 from slot (t+1)&3
//     RAW: stage t+1 was issued during step t-1; at the barrier of step t each wave has at most the 9 pieces of
//          stage t+2 in flight (in-order vmcnt).   WAR: slot (t+2)&3 was last read in step t-2, and a wave can
//          only be in step t after every wave passed the barrier of step t-1.
//   epilogue     operands are the row bits of the job's last two K steps (16 registers per lane).
// Results are bit-identical to scan_quad_kernel.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>
#include "gemm_i8_core.h"
#include "gemm_i8_w4.h"
#include "mmg_internal.h"

namespace mmg {

constexpr int WB_QB = 256 * 16;                    // packed genotype tile
constexpr int WB_SLOT = TILE_BYTES + WB_QB;        // 36 KiB
constexpr int WB_LDS = 4 * WB_SLOT;                // 144 KiB

struct FragB {
  v4i a[4], b[4];
};

__device__ __forceinline__ v4i expand_bits16(uint32_t x16) {
  v4i o;
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (int)((((x16 >> (4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u);
  return o;
}

__device__ __forceinline__ v16i mfma8b(v4i a, v4i b, v16i c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }

template <bool FAST>
__global__ __launch_bounds__(W4_THREADS) void scan_quad_w4b_kernel(
    const uint8_t* __restrict__ Sb, int64_t ldSb, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
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
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int arow = wm * 128 + r, brow = wn * 128 + r;

  // ---- issue cursor over the flattened stage stream (wave-uniform scalars)
  int cj = j0;
  int2 cjb = jobs[cj];
  int cks = 0, cnks = 2 * (cjb.y + 1);
  StageOp4 sp = make_stage_op4(Bq + (int64_t)cjb.x * digit_stride + (int64_t)cjb.y * TM * ldB, ldB, wave, lane);
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Sb + (int64_t)sb * TN * ldSb), 0, 0x7fffffff, 0x00020000);
  const int vq = (wave * 64 + lane) * (int)ldSb;     // piece `wave` of the Qb tile: rows wave*64 .. +63, 16 B per row
  auto advance = [&]() {
    if (cks + 1 < cnks) { ++cks; return; }
    if (cj + 1 < j1) {
      ++cj;
      cjb = jobs[cj];
      cks = 0;
      cnks = 2 * (cjb.y + 1);
      sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Bq + (int64_t)cjb.x * digit_stride + (int64_t)cjb.y * TM * ldB),
                                                0, 0x7fffffff, 0x00020000);
    }                                                // else: stay on the last stage (harmless re-issue)
  };
  // piece 0-7: P rows, 8: this wave's Qb rows, of the cursor's stage into `slot`
  auto piece = [&](int pc, char* slot) {
    if (pc < 8) stage_piece4(sp, cks * BK, slot, wave, pc);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (MMG_AS3 void*)(slot + TILE_BYTES + wave * 1024), 16, vq, cks * 16, 0, 0);
  };

  // ---- prologue: stages 0 and 1 issued, stage 0 landed
#pragma unroll
  for (int i = 0; i < 9; ++i) piece(i, lds);
  advance();
#pragma unroll
  for (int i = 0; i < 9; ++i) piece(i, lds + WB_SLOT);
  asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  FragB f0, f1;
  v4i rb0[4], rb1[4];                                // row bits of the current / next K step (4 dwords = 4 slices)
#pragma unroll
  for (int i = 0; i < 4; ++i) rb0[i] = *(const v4i*)(lds + TILE_BYTES + (brow + i * 32) * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f0.a[i] = lds_frag(lds, arow + i * 32, h);
    f0.b[i] = expand_bits16(((uint32_t)rb0[i][0] >> (16 * h)) & 0xFFFFu);
  }

  v16i acc[4][4];                                    // written (not accumulated) by the first slice of every job
  unsigned long long qacc[4] = {0ull, 0ull, 0ull, 0ull};
  v4i cap[4];

  int t = 0;
  // dword e (4 genotypes) of a B fragment from the 16 row bits x16
  auto exp_dw = [&](uint32_t x16, int e) { return (int)((((x16 >> (4 * e)) & 0xFu) * 0x00204081u) & 0x01010101u); };
  // slices 0-2 of a step: MFMAs on `cur`, fragments of slice S+1 into `nxt`, three DMA pieces of stage t+2.
  // Source order IS the schedule: one sched_barrier per MFMA keeps the (MFMA, 3-4 VALU, ds_read | DMA) groups
  // apart -- left alone the compiler runs the whole fragment expansion as one VALU block with the MFMA pipe idle.
  auto mid_slice = [&](auto zero_tag, auto s_tag, const FragB& cur, FragB& nxt, const v4i (&rb)[4], const char* pt,
                       char* dst) {
    constexpr bool ZERO = decltype(zero_tag)::value;
    constexpr int SL = decltype(s_tag)::value;         // 0, 1, 2
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = i >> 2, n = i & 3;
      if (ZERO) acc[m][n] = mfma8b(cur.a[m], cur.b[n], v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
      else acc[m][n] = mfma8b(cur.a[m], cur.b[n], acc[m][n]);
      nxt.b[i >> 2][i & 3] = exp_dw(((uint32_t)rb[i >> 2][SL + 1] >> (16 * h)) & 0xFFFFu, i & 3);
      if ((i & 1) == 0 && i < 8) nxt.a[i >> 1] = lds_frag(pt, arow + (i >> 1) * 32, 2 * (SL + 1) + h);
      if ((i & 1) == 1 && i < 6) piece(3 * SL + (i >> 1), dst);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto step = [&](int ks, v4i (&rbc)[4], v4i (&rbn)[4]) {
    char* cur = lds + (t & 3) * WB_SLOT;
    char* nxs = lds + ((t + 1) & 3) * WB_SLOT;
    char* dst = lds + ((t + 2) & 3) * WB_SLOT;
    advance();                                       // -> stage t+2, issued during this step (the cursor's branches
                                                     //    stay out of the slice code: one basic block per step)
    if (ks == 0) mid_slice(std::true_type{}, std::integral_constant<int, 0>{}, f0, f1, rbc, cur, dst);
    else mid_slice(std::false_type{}, std::integral_constant<int, 0>{}, f0, f1, rbc, cur, dst);
    mid_slice(std::false_type{}, std::integral_constant<int, 1>{}, f1, f0, rbc, cur, dst);
    mid_slice(std::false_type{}, std::integral_constant<int, 2>{}, f0, f1, rbc, cur, dst);
    asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // slice 3: MFMA on registers; row bits and first A fragments of step t+1, then its first B fragments
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = i >> 2, n = i & 3;
      acc[m][n] = mfma8b(f1.a[m], f1.b[n], acc[m][n]);
      if (i < 4) rbn[i] = *(const v4i*)(nxs + TILE_BYTES + (brow + i * 32) * 16);
      else if (i < 8) f0.a[i - 4] = lds_frag(nxs, arow + (i - 4) * 32, h);
      else {
        const int fr = (i - 8) >> 1, e0 = 2 * ((i - 8) & 1);
        const uint32_t x16 = ((uint32_t)rbn[fr][0] >> (16 * h)) & 0xFFFFu;
        f0.b[fr][e0] = exp_dw(x16, e0);
        f0.b[fr][e0 + 1] = exp_dw(x16, e0 + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    ++t;
  };
  // every job has an even number of K steps, so the row-bit registers alternate rb0 / rb1 with the parity of ks
  for (int jj = j0; jj < j1; ++jj) {
    const int2 jb = jobs[jj];
    const int d = jb.x, nks = 2 * (jb.y + 1);
    for (int ks = 0; ks < nks - 2; ks += 2) {
      step(ks, rb0, rb1);
      step(ks + 1, rb1, rb0);
    }
    // epilogue operands: the row bits of step nks-2 (row half wm = 0) / nks-1 (wm = 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) cap[i] = rb0[i];
    step(nks - 2, rb0, rb1);
    if (wm == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) cap[i] = rb1[i];
    }
    step(nks - 1, rb1, rb0);
    // ---- epilogue of job jj: qacc[n] += (sum_j T[j][snp] * s[snp][256J + j]) << 8d
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      long long part = 0;
      if (FAST) {
        int p32 = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t nib = ((uint32_t)cap[n][m] >> (8 * g4 + 4 * h)) & 0xFu;
#pragma unroll
            for (int e = 0; e < 4; ++e) p32 += __mul24(acc[m][n][g4 * 4 + e], (int)((nib >> e) & 1u));
          }
        part = p32;
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t nib = ((uint32_t)cap[n][m] >> (8 * g4 + 4 * h)) & 0xFu;
#pragma unroll
            for (int e = 0; e < 4; ++e) part += ((nib >> e) & 1u) ? (long long)acc[m][n][g4 * 4 + e] : 0ll;
          }
      }
      qacc[n] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-issued tail stages must land before LDS is released
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    unsigned long long v = qacc[n];
    v += __shfl_xor(v, 32);
    if (h == 0) atomicAdd(q + (int64_t)sb * TN + wn * 128 + n * 32 + r, v);
  }
}

void launch_scan_quad_w4b(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, unsigned long long* q) {
  const int nSb = (int)(g->Mpad / TN);
  const int per = 8 * md.AS;
  const int ncoh = (nSb + per - 1) / per;
  // binary store: |s| <= 1, so the 24-bit epilogue only needs Npad < 2^16
  const bool fast = md.Npad < (1 << 16) && !std::getenv("MMG_W4S_SLOW_EPI");
#define MMG_LAUNCH_W4B(F)                                                                                              \
  do {                                                                                                                 \
    hipFuncSetAttribute((const void*)scan_quad_w4b_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, WB_LDS);    \
    hipLaunchKernelGGL(scan_quad_w4b_kernel<F>, dim3((unsigned)(ncoh * 256)), dim3(W4_THREADS), WB_LDS, ctx->stream,  \
                       g->bits, (int64_t)(g->Npad >> 3), nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,    \
                       md.job_off, md.jobs, md.AS, q);                                                                 \
  } while (0)
  if (fast) MMG_LAUNCH_W4B(true);
  else MMG_LAUNCH_W4B(false);
#undef MMG_LAUNCH_W4B
}

}  // namespace mmg
