// k_scan_w4s.hip -- quadratic-form GEMM of the EMMAX scan, third generation: 4 waves per workgroup
// (one per SIMD, the whole 512-entry register file), wave tile 128 x 128, and a software pipeline
// that is written out slice by slice instead of being left to the compiler:
//
//   K step = 128 bytes = 4 slices of 32 bytes; per wave and slice: 16 MFMA (4 x 4 tiles of
//   v_mfma_i32_32x32x32_i8), 8 ds_read_b128 that fetch the NEXT slice's fragments into the other half
//   of a register double buffer, and (slices 0 and 3 only) 8 LDS-DMA pieces -- interleaved
//   MFMA / ds_read / MFMA / DMA, so that the matrix pipe never waits for an address-path burst.
//
//   LDS: the 2-slot, 128-byte-row image of gemm_i8_core.h (same swizzle).  Stage u = K step u of the
//   workgroup's flattened job stream (the prefetch never drains at a job boundary):
//       step t, slice 0:  Q half of stage t+1  -> slot (t+1)&1
//       step t, slices 0-2: MFMA on slot t&1 (fragments one slice ahead)
//       s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier        <- the only barrier of the step
//       step t, slice 3:  MFMA on registers; fragments of step t+1 slice 0 from slot (t+1)&1;
//                         P half of stage t+2 -> slot t&1
//     RAW: stage t+1 (P half issued in step t-1, Q half in slice 0 of step t) is complete on every wave
//          at that wave's vmcnt(0) before the barrier; it is first read after the barrier.
//     WAR: slot t&1 is rewritten (P: slice 3 of step t, Q: slice 0 of step t+1) only after the barrier,
//          which every wave reaches with lgkmcnt(0), i.e. with all its reads of slot t&1 returned.
//   The barrier sits before the LAST slice, whose fragments are already in registers, so no wave waits
//   for LDS latency after it.  Beyond the end of the stream the cursor re-issues the last stage into
//   slots nobody reads (branch-free step body); the kernel drains vmcnt before it exits.
//
//   Epilogue operands (the genotype bytes s[snp][256J + j] that multiply row j of the accumulators) are
//   the Q tiles of the job's last two K steps: every wave captures 64 dwords from LDS before step nks-2 (row
//   half wm = 0 keeps them, wm = 1 overwrites them before step nks-1 -- an unconditional first capture keeps the
//   64 registers from being live across jobs) and holds them in registers.  The first slice of a job writes the
//   accumulators with C = 0 instead of clearing them; the epilogue runs on v_mad_i32_i24 when the host can prove
//   from the store's tracked max |s| that nothing overflows (FAST), on 64-bit multiply-adds otherwise.
//
// Which digit planes a launch covers is the caller's business (job list): the adaptive schedule of
// api.hip:mmg_emmax_scan_device launches this kernel twice per scan (planes 1-3 on everything, plane 0 on the
// compact store of the SNPs that need it).
// Results are bit-identical to scan_quad_kernel (exact integers, 64-bit integer atomics).
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <string>
#include <vector>
#include "gemm_i8_core.h"
#include "gemm_i8_w4s.h"
#include "mmg_internal.h"

// cache policy of the digit-tile (P) / genotype-tile (Q) LDS-DMA loads: 0 = default, 2 = nt.  Build-time experiment
// knobs (make DEFS="-DMMG_SCAN_AUX_P=2"); see DESIGN.md 4.1 for what they measured.
#ifndef MMG_SCAN_AUX_P
#define MMG_SCAN_AUX_P 0
#endif
#ifndef MMG_SCAN_AUX_Q
#define MMG_SCAN_AUX_Q 0
#endif

namespace mmg {

// MMG_W4S_PIN=1 (A/B builds, `make DEFS=-DMMG_W4S_PIN=1`): every (MFMA, read, MFMA, read/DMA) group ends in a full
// scheduling barrier and the slice in front of the step's barrier issues its eight fragment reads in the first four
// MFMA pairs.  Tried in round 2 because the ISA of the default build shows one slice's reads issued in reverse order
// (the first MFMA of the next slice waits for the two reads issued last) and the last read of a step issued one MFMA
// before `s_waitcnt lgkmcnt(0)`; measured on the same box: 24.93 vs 24.88 ms, per-K-step stamps 2730 vs 2702 cycles --
// no gain, the LDS latency was already covered by the MFMAs in flight.  Bit-identical results either way.
#ifndef MMG_W4S_PIN
#define MMG_W4S_PIN 0
#endif
constexpr bool PIN_SLICES = MMG_W4S_PIN != 0;

// One slice: 16 MFMA on `cur`; fragment reads of (slot `src`, chunk) into `nxt`; DMA pieces [P0, P1) of the
// cursor's stage (pieces 0-7: P rows, 8-15: Q rows of this wave) into slot `dst`.
// FRONT (the slice in front of the step's barrier): all eight fragment reads are issued in the first four MFMA
// pairs (two per pair), so that they have returned when the wave reaches `s_waitcnt lgkmcnt(0)` before the barrier
// -- spread one per pair, the last read is issued one MFMA before that wait and its whole LDS latency is exposed
// once per K step.  Every (MFMA, read, MFMA, read/DMA) group ends in a full scheduling barrier: the class-level
// sched_group_barrier hints of the first version left the compiler free to choose WHICH read fills a slot, and it
// issued one slice's reads in reverse (bottom-up) order, so that the first MFMA of the following slice waited for
// the two reads issued last.
template <bool LOAD, int P0, int P1, bool ZERO = false, int HOT = 0, bool FRONT = false>   // HOT: 1 = all pieces, 2 = Q pieces, 3 = P pieces
__device__ __forceinline__ void slice(v16i (&acc)[4][4], const Frag4& cur, Frag4& nxt, const char* src, int arow,
                                      int brow, int chunk, const StageOp4& sp, const StageOp4& sq, int k0, char* dst,
                                      int wave) {
  static_assert(P1 - P0 <= 8, "at most one DMA piece per MFMA pair");
  static_assert(!FRONT || P1 == P0, "the front-loaded slice carries no DMA pieces");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m0 = ORD_M[2 * i], n0 = ORD_N[2 * i], m1 = ORD_M[2 * i + 1], n1 = ORD_N[2 * i + 1];
    if (ZERO) acc[m0][n0] = mfma8(cur.a[m0], cur.b[n0], v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
    else acc[m0][n0] = mfma8(cur.a[m0], cur.b[n0], acc[m0][n0]);
    if (LOAD) {
      if (FRONT) {
        if (i < 4) nxt.a[i] = lds_frag(src, arow + i * 32, chunk);
      } else {
        if ((i & 1) == 0) nxt.a[i >> 1] = lds_frag(src, arow + (i >> 1) * 32, chunk);
        else nxt.b[i >> 1] = lds_frag(src + TILE_BYTES, brow + (i >> 1) * 32, chunk);
      }
    }
    if (ZERO) acc[m1][n1] = mfma8(cur.a[m1], cur.b[n1], v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
    else acc[m1][n1] = mfma8(cur.a[m1], cur.b[n1], acc[m1][n1]);
    if (LOAD && FRONT && i < 4) nxt.b[i] = lds_frag(src + TILE_BYTES, brow + i * 32, chunk);
    if (P0 + i < P1) {
      const int pc = P0 + i;
      if (HOT == 1 || (HOT == 2 && pc >= 8) || (HOT == 3 && pc < 8)) {   // ablation: same LDS-DMA traffic, source always the same 2 KiB
        if (pc < 8) stage_piece4<MMG_SCAN_AUX_Q>(sq, 0, dst, wave, pc & 1);
        else stage_piece4<MMG_SCAN_AUX_Q>(sq, 0, dst + TILE_BYTES, wave, pc & 1);
      } else if (pc < 8) {
        stage_piece4<MMG_SCAN_AUX_P>(sp, k0, dst, wave, pc);
      } else {
        stage_piece4<MMG_SCAN_AUX_Q>(sq, k0, dst + TILE_BYTES, wave, pc - 8);
      }
    }
    if (PIN_SLICES) __builtin_amdgcn_sched_barrier(0);
  }
  if (!PIN_SLICES) sched_slice<(P1 > P0 ? P1 - P0 : 0)>();
}

// ABL: 0 = production; timing ablations with WRONG results: 1 = no DMA in the loop, 2 = no fragment reads in
// the loop, 3 = trivial epilogue, 4 = in-kernel stamps, 5 = all DMA pieces read the same 2 KiB (L1-resident), 6 / 7 = only the genotype / only the digit pieces do.
// N3/N0/N1: DMA pieces of a stage issued in slice 3 (right after the barrier that frees the slot) and in slices
// 0 / 1 of the following step; the remaining 16 - N3 - N0 - N1 go into slice 2.
// FAST: |s| * Npad < 2^16 and s^2 * Npad < 2^18 (checked on the host from the store's tracked max |s|): every
// accumulator fits 24 bits and a lane's 64 products per SNP fit 32 bits, so the epilogue runs on v_mad_i32_i24.
// LIN: rows 240 .. 255 of the LAST tile row of the TOP digit plane are not matrix rows but digit images of the
// vectors of the linear terms (api.hip:add_linear_rows; the genotype columns they would multiply in this kernel's
// own epilogue are padding individuals, i.e. zeros): seven balanced base-256 digits of w in rows 240-243, 248-250,
// ones in row 251, seven digits of diag(A) in rows 244-247, 252-254.  Their accumulators are s.w, sum_i s_i and
// sum_i A_ii s_i of the tile's SNPs as exact integers -- in the C layout registers 8-15 of tile m = 3 of the waves with
// wm = 1, lanes with h = 0 (w, ones) and h = 1 (diag) -- and are written out when the job (top plane, last tile row)
// ends: for a binary store (s^2 = s) that is everything scan_finalize_kernel reads the 5 GB genotype store a second
// time for.
// The kernel only drops the raw accumulators (16 ints per SNP: [h][8]) -- anything more in this epilogue and the
// register allocator, already at 488 of 512, starts spilling; scan_finalize_lin_kernel recombines the digits.
struct LinArgs {
  int* raw;                     // [Mpad][16]
  int dtop, jlast;              // the job that carries the rows
};

template <int ABL, int N3, int N0, int N1, bool FAST, bool LIN = false>
__global__ __launch_bounds__(W4_THREADS) void scan_quad_w4s_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
    int64_t digit_stride, const int* __restrict__ job_off, const int2* __restrict__ jobs, int AS, int sb_base,
    unsigned long long* __restrict__ q, unsigned long long* __restrict__ dbg, LinArgs lin) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr bool LD = ABL != 2;
  unsigned long long seg[4] = {0, 0, 0, 0};              // ABL 4: cycles in {vmcnt+lgkm wait, barrier, epilogue, total}
  const unsigned long long T0 = ABL == 4 ? stamp() : 0;
  constexpr int E3 = ABL == 1 ? 0 : N3, E0 = ABL == 1 ? 0 : N3 + N0, E1 = ABL == 1 ? 0 : N3 + N0 + N1,
                E2 = ABL == 1 ? 0 : 16;
  const int b = blockIdx.x;
  const int x = b & 7, bi = b >> 3;
  const int cohort = bi >> 5, within = bi & 31;
  const int a_ = within % AS, grp = within / AS;
  const int sb = sb_base + (cohort * 8 + x) * AS + a_;
  if (sb >= nSb) return;
  const int j0 = job_off[grp], j1 = job_off[grp + 1];
  if (j1 <= j0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int8_t* Q = S + (int64_t)sb * TN * ldS;
  const int arow = wm * 128 + r, brow = wn * 128 + r;

  // ---- issue cursor over the flattened stage stream (wave-uniform scalars)
  int cj = j0;                                           // job of the stage the cursor points at
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

  // ---- prologue: stage 0 complete, the first N3 pieces of stage 1 in flight, fragments of step 0 slice 0
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece4<MMG_SCAN_AUX_P>(sp, 0, lds, wave, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece4<MMG_SCAN_AUX_Q>(sq, 0, lds + TILE_BYTES, wave, i);
  advance();                                             // -> stage 1
#pragma unroll
  for (int i = 0; i < N3; ++i) {
    if (i < 8) stage_piece4<MMG_SCAN_AUX_P>(sp, cks * BK, lds + BUF_BYTES, wave, i);
    else stage_piece4<MMG_SCAN_AUX_Q>(sq, cks * BK, lds + BUF_BYTES + TILE_BYTES, wave, i - 8);
  }
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
  unsigned long long qacc[4] = {0ull, 0ull, 0ull, 0ull};
  int cap[4][4][2][2];                                   // [n][m][chunk of the 32-byte group][dword pair]

  int t = 0;
  // capture: dwords (4h + 8q) / 4 of every 32-byte row group of this wave's 4 x 32 SNP rows, from the Q tile
  // of the step about to run (must be issued before that step's barrier)
  auto capture = [&]() {
    const char* qt = lds + (t & 1) * BUF_BYTES + TILE_BYTES;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          // whole 16-byte chunks (conflict-free like the fragment reads; ds_read2_b32 of just the two dwords
          // cost 17 % LDS bank-conflict cycles), then the lane's half: dwords h and h + 2
          const v4i ch = lds_frag(qt, brow + n * 32, 2 * m + cc);
          cap[n][m][cc][0] = h ? ch[1] : ch[0];
          cap[n][m][cc][1] = h ? ch[3] : ch[2];
        }
  };
  auto step = [&](int ks) {
    char* cur = lds + (t & 1) * BUF_BYTES;
    char* oth = lds + ((t + 1) & 1) * BUF_BYTES;
    // slices 0-2: MFMA on slot t&1; the rest of stage t+1 (the cursor's stage) -> the other slot
    const int k1 = cks * BK;
    constexpr int HOT = ABL == 5 ? 1 : ABL == 6 ? 2 : ABL == 7 ? 3 : 0;
    if (ks == 0) slice<LD, E3, E0, true, HOT>(acc, f0, f1, cur, arow, brow, 2 + h, sp, sq, k1, oth, wave);
    else slice<LD, E3, E0, false, HOT>(acc, f0, f1, cur, arow, brow, 2 + h, sp, sq, k1, oth, wave);
    slice<LD, E0, E1, false, HOT>(acc, f1, f0, cur, arow, brow, 4 + h, sp, sq, k1, oth, wave);
    slice<LD, E1, E2, false, HOT, PIN_SLICES && (E2 == E1)>(acc, f0, f1, cur, arow, brow, 6 + h, sp, sq, k1, oth, wave);
    const unsigned long long tb0 = ABL == 4 ? stamp() : 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long tb1 = ABL == 4 ? stamp() : 0;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ABL == 4) { const unsigned long long tb2 = stamp(); seg[0] += tb1 - tb0; seg[1] += tb2 - tb1; }
    advance();                                           // -> stage t+2
    // slice 3: MFMA on registers; fragments of step t+1 slice 0; first pieces of stage t+2 -> the slot just retired
    slice<LD, 0, E3, false, HOT>(acc, f1, f0, oth, arow, brow, h, sp, sq, cks * BK, cur, wave);
    ++t;
  };
  for (int jj = j0; jj < j1; ++jj) {
    const int2 jb = jobs[jj];
    const int d = jb.x, nks = 2 * (jb.y + 1);
    for (int ks = 0; ks < nks - 2; ++ks) step(ks);
    capture();                                           // step nks-2 holds the operands of row half wm = 0 ...
    step(nks - 2);
    if (wm == 1) capture();                              // ... and step nks-1 those of row half wm = 1
    step(nks - 1);
    // ---- epilogue of job jj: qacc[n] += (sum_j T[j][snp] * s[snp][256J + j]) << 8d, then clear
    const unsigned long long te0 = ABL == 4 ? stamp() : 0;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      long long part = 0;
      if (ABL == 3) {
#pragma unroll
        for (int m = 0; m < 4; ++m) part += acc[m][n][0] + cap[n][m][0][0];
      } else if (FAST) {
        int p32 = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              const int wd = cap[n][m][cc][gp];
              const int g4 = 2 * cc + gp;
#pragma unroll
              for (int e = 0; e < 4; ++e) p32 += __mul24(acc[m][n][g4 * 4 + e], (int)(int8_t)((wd >> (8 * e)) & 0xff));
              __builtin_amdgcn_sched_barrier(0);           // keep the AGPR reads next to their use (register pressure)
            }
        part = p32;
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              const int wd = cap[n][m][cc][gp];
              const int g4 = 2 * cc + gp;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                part += (long long)acc[m][n][g4 * 4 + e] * (long long)(int)(int8_t)((wd >> (8 * e)) & 0xff);
            }
      }
      qacc[n] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
    }
    if (LIN) {
      // No branch here: with one, the register allocator (488 of 512 in use) spilled 230 registers.  Every wave issues
      // the eight stores at the end of every job, through a buffer descriptor that is EMPTY unless this is the job that
      // carries the linear rows and the wave owns rows 128-255: out-of-range buffer stores are dropped by the address
      // unit, no memory traffic.
      const bool hit = d == lin.dtop && jb.y == lin.jlast && wm == 1;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(lin.raw + ((int64_t)sb * TN + wn * 128) * 16), 0, hit ? 128 * 16 * 4 : 0, 0x00020000);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int off = ((n * 32 + r) * 16 + h * 8) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(
            v4i{acc[3][n][8], acc[3][n][9], acc[3][n][10], acc[3][n][11]}, rs, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(
            v4i{acc[3][n][12], acc[3][n][13], acc[3][n][14], acc[3][n][15]}, rs, off + 16, 0, 0);
      }
    }
    if (ABL == 4) {
#pragma unroll
      for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(qacc[n]));
      seg[2] += stamp() - te0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-issued tail stages must land before LDS is released
  if (ABL == 4 && lane == 0 && blockIdx.x < 16384) {
    seg[3] = stamp() - T0;
    unsigned long long* o = dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3]; o[4] = (unsigned long long)t;
  }
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    unsigned long long v = qacc[n];
    v += __shfl_xor(v, 32);
    if (h == 0) atomicAdd(q + (int64_t)sb * TN + wn * 128 + n * 32 + r, v);
  }
}

bool scan_lin_usable(const mmg_geno* g, const mmg_scan_model& md) {
  const char* e = std::getenv("MMG_SCAN_FUSED_LINEAR");     // =0: keep the separate finalize pass over the store (A/B, tests)
  const bool off = e && e[0] == '0';
  // (diagnostic builds: only the production kernel writes the by-products)
  // the by-products are linear in s: valid for any store; what a binary store adds is s^2 = s.  A store of 0/1/2 codes takes
  // them once its [s = 2] bit image exists (mmg_internal.h:mmg_geno::hi2, api.hip:hi2_prepare) -- the finalize step then
  // corrects sum A_ii s_i^2 and sum s_i^2 from the image
  const bool values_ok = g->sneg == 0 && (g->smax <= 1 || (g->smax == 2 && md.lin_tab != nullptr && geno_hi2_ready(g)));
  return !off && md.lin_rows && values_ok && std::getenv("MMG_W4S_ABL") == nullptr &&
         std::getenv("MMG_SCAN_KERNEL") == nullptr && std::getenv("MMG_ABLATE") == nullptr &&
         std::getenv("MMG_W4S_DIST") == nullptr;
}

void launch_scan_quad_w4s(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, unsigned long long* q,
                          const LinOut* lin) {
  const int nSb = (int)(g->Mpad / TN);
  const int per = 8 * md.AS;
  // full cohorts with the model's grouping; the remaining r < 8 AS blocks as a second launch with the jobs split over 2x or
  // 4x as many groups wherever that takes fewer (shorter) rounds: ceil(r / (8 AS')) rounds of relative length AS' / AS.
  // (N = 50,000, 50,000-SNP chunks = 196 blocks: 6 rounds + 1 round for 4 blocks was 144 ms; MMG_SCAN_TAIL=0 keeps that.)
  int nfull = nSb / per, tail_k = -1;
  {
    const char* e = std::getenv("MMG_SCAN_TAIL");
    const int r = nSb - nfull * per;
    if (r > 0 && !(e && e[0] == '0') && std::getenv("MMG_W4S_ABL") == nullptr) {
      double best = 1.0;
      for (int k = 0; k < 2; ++k) {
        const int as = md.AS >> (k + 1);
        if (as < 1 || !md.tail_off[md.range][k]) continue;
        const double cost = (double)((r + 8 * as - 1) / (8 * as)) * as / md.AS;
        if (cost < best - 1e-9) { best = cost; tail_k = k; }
      }
    }
    if (tail_k < 0) nfull = (nSb + per - 1) / per;          // one launch covers everything
  }
  const int ncoh = nfull;
  int abl = 0, dist = 0;
  const int64_t smax = g->smax;
  bool fast = smax * md.Npad < (1 << 16) && smax * smax * md.Npad < (1 << 18);
  if (std::getenv("MMG_W4S_SLOW_EPI")) fast = false;
#ifdef MMG_EXPERIMENTS
  // timing ablations (WRONG results) and the piece-distribution A/B: only in a `make EXPERIMENTS=1` library -- no environment
  // variable can switch the shipped library to a kernel that does not compute the scan
  if (const char* e = std::getenv("MMG_W4S_ABL")) abl = std::atoi(e);
  if (const char* e = std::getenv("MMG_W4S_DIST")) dist = std::atoi(e);
#endif
#define MMG_LAUNCH_W4S(...)                                                                                             \
  do {                                                                                                                  \
    hipFuncSetAttribute((const void*)scan_quad_w4s_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                        LDS_BYTES);                                                                                     \
    if (ncoh > 0)                                                                                                       \
      hipLaunchKernelGGL((scan_quad_w4s_kernel<__VA_ARGS__>), dim3((unsigned)(ncoh * 256)), dim3(W4_THREADS), LDS_BYTES, \
                         ctx->stream, g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad,                             \
                         (int64_t)md.Npad * md.Npad, md.job_off, md.jobs, md.AS, 0, q, dbg, la);                        \
    if (tail_k >= 0) {                                                                                                  \
      const int as_t = md.AS >> (tail_k + 1), r_t = nSb - ncoh * per;                                                   \
      const int ncoh_t = (r_t + 8 * as_t - 1) / (8 * as_t);                                                             \
      hipLaunchKernelGGL((scan_quad_w4s_kernel<__VA_ARGS__>), dim3((unsigned)(ncoh_t * 256)), dim3(W4_THREADS),         \
                         LDS_BYTES, ctx->stream, g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad,                  \
                         (int64_t)md.Npad * md.Npad, md.tail_off[md.range][tail_k], md.tail_jobs[md.range][tail_k],     \
                         as_t, ncoh * per, q, dbg, la);                                                                 \
    }                                                                                                                   \
  } while (0)
  LinArgs la{nullptr, -1, -1};
  const bool use_lin = lin != nullptr && lin->raw != nullptr && scan_lin_usable(g, md);
  if (use_lin) la = LinArgs{lin->raw, md.D - 1, md.Npad / TM - 1};
  static unsigned long long* dbg = nullptr;
  (void)abl; (void)dist;
#ifdef MMG_EXPERIMENTS
  if (abl == 4) {
    const size_t n = (size_t)16384 * 4 * 8;
    if (!dbg) hipMalloc(&dbg, n * sizeof(unsigned long long));
    hipMemsetAsync(dbg, 0, n * sizeof(unsigned long long), ctx->stream);
    if (fast) MMG_LAUNCH_W4S(4, 8, 8, 0, true); else MMG_LAUNCH_W4S(4, 8, 8, 0, false);
    std::vector<unsigned long long> hbuf(n);
    hipMemcpyAsync(hbuf.data(), dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    double sm[5] = {0, 0, 0, 0, 0};
    long cnt = 0;
    for (size_t w = 0; w < (size_t)16384 * 4; ++w) {
      if (hbuf[w * 8 + 4] == 0) continue;
      for (int k = 0; k < 5; ++k) sm[k] += (double)hbuf[w * 8 + k];
      ++cnt;
    }
    if (cnt)
      fprintf(stderr, "[w4s stamps] waves %ld  K-steps/wave %.0f  per K-step cycles: total %.0f  mem wait %.0f  barrier %.0f  "
                      "epilogue (amortised) %.0f\n", cnt, sm[4] / cnt, sm[3] / sm[4], sm[0] / sm[4], sm[1] / sm[4], sm[2] / sm[4]);
    return;
  }
  if (abl == 1) MMG_LAUNCH_W4S(1, 8, 8, 0, true);
  else if (abl == 2) MMG_LAUNCH_W4S(2, 8, 8, 0, true);
  else if (abl == 3) MMG_LAUNCH_W4S(3, 8, 8, 0, true);
  else if (abl == 5) MMG_LAUNCH_W4S(5, 8, 8, 0, true);
  else if (abl == 6) MMG_LAUNCH_W4S(6, 8, 8, 0, true);
  else if (abl == 7) MMG_LAUNCH_W4S(7, 8, 8, 0, true);
  else if (dist == 1 && fast) MMG_LAUNCH_W4S(0, 6, 5, 5, true);
  else
#endif
  if (fast && use_lin) MMG_LAUNCH_W4S(0, 8, 8, 0, true, true);
  else if (fast) MMG_LAUNCH_W4S(0, 8, 8, 0, true);
  else if (use_lin) MMG_LAUNCH_W4S(0, 8, 8, 0, false, true);
  else MMG_LAUNCH_W4S(0, 8, 8, 0, false);
#undef MMG_LAUNCH_W4S
}

int run_scan_quad(mmg_ctx* ctx, mmg_geno* g, const mmg_scan_model& md, unsigned long long* q, int ev_slot,
                  const LinOut* lin) {
  // Production: the 4-wave x 128x128 hand-laid pipeline above.  A library built with `make EXPERIMENTS=1`
  // (csrc/experiments/: the superseded generations q8 / w4m / w4b / bits / timed / m16 / flat / ring / pp, all
  // bit-identical) honours MMG_SCAN_KERNEL / MMG_ABLATE for A/B runs; the shipped library ignores them.
#ifdef MMG_EXPERIMENTS
  const char* kv = std::getenv("MMG_SCAN_KERNEL");
  const std::string k = kv ? kv : "";
  const bool ablate = std::getenv("MMG_ABLATE") != nullptr;
  const bool want_w4b = k == "w4b" && !ablate;
  const bool want_bits = (k == "bits" || want_w4b) && !ablate;
  if (want_bits) {
    int rc = ensure_bits(ctx, g);                     // once per store content
    if (rc) return rc;
  }
  EvScope ev(ctx, ev_slot);
  if (want_w4b && g->binary) launch_scan_quad_w4b(ctx, g, md, q);
  else if (want_bits && g->binary) launch_scan_quad_bits(ctx, g, md, q);
  else if (!ablate && k == "w4m") launch_scan_quad_w4m(ctx, g, md, q);
  else if (!ablate && (k.empty() || k == "w4s" || k == "w4b" || k == "bits")) launch_scan_quad_w4s(ctx, g, md, q, lin);
  else launch_scan_quad(ctx, g, md, q);
#else
  EvScope ev(ctx, ev_slot);
  launch_scan_quad_w4s(ctx, g, md, q, lin);
#endif
  return MMG_OK;
}

}  // namespace mmg

extern "C" int mmg_has_experiments(void) {
#ifdef MMG_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}
