// gemm_i8_grm4.h -- the FOUR digit planes of the exact GRM of a binary store in one pass over the genotypes.
//
// api.hip writes the weighted Gram matrix sum_m omega_m s_m s_m' as sum_d 128^d C_d, C_d = (diag(dig_d) S)' S with the
// 7-bit digits dig_d(m) of the weight of SNP m: four int8 GEMMs over the same 5 GB store, each with its own digit-scaled
// image of it (round 3: SNP-major, read through the transposed LDS reads of gemm_i8_w4tr.h).  Those GEMMs are limited by
// what goes INTO LDS as much as by the matrix pipe (DESIGN.md 4.1 "fill bandwidth"), and the four scaled images differ
// from the store only by a factor per k row.  Here a workgroup fills LDS with the PLAIN genotype tiles once per K step
// and forms the four scaled operands in registers: genotype bytes are 0 / 1, so (dword + 0x7f7f7f7f) ^ 0x7f7f7f7f turns them
// into byte masks and one v_and with the digits of the lane's four k rows is the product -- 48 VALU ops per 16 MFMA, in the
// shadow of the matrix pipe.  Per K step (128 SNP rows) a workgroup computes a 128 x 128 tile of ALL FOUR planes:
//   LDS fill   32 KiB per 4 x 128 x 128 x 128 MAC   = half the bytes per MAC of the 256 x 256 single-plane tile
//   LDS reads  4 fragments per 16 MFMA              = half of gemm_i8_w4tr.h's 8
//   HBM        the store once instead of the store + four images (and no image pass: 6.6 ms at N = 5000 x M = 1e6)
// Registers: wave tile 64 x 64 x 4 planes = 16 accumulator tiles of 32 x 32 (256 AGPR), as many as the 128 x 128 tile of
// one plane.
//
// LDS image of an operand tile per K step: [128 k rows][128 cols] bytes (16 KiB), filled by lane-linear LDS-DMA (16 B per
// lane, one instruction = 8 k rows x 128 B); the 16-byte chunks of row k sit at position chunk ^ (((k >> 1) & 3) << 1): a
// transposed read of a 32-lane half covers 8 consecutive k rows x 32 bytes (an aligned chunk pair), rows k, k + 2 would
// share banks (128 B apart) -- with the swizzle the 8 rows fall into the 8 different bank octets.  Behind the two tiles:
// the digits of the step, [4 planes][128 k] bytes.
//
// Pipeline: the K step is 4 slices of 32 k rows; raw fragments and digits are read two slices ahead and scaled one slice
// ahead (inline asm reads, one lgkmcnt(0) per slice when only reads issued a whole slice ago are outstanding); two LDS
// slots, one barrier per K step, stage t + 2 issued behind it.  Timing ablations at N = 5000 x M = 1e6 (MMG_GRM4_ABL):
// 36.3 ms as is, 30.1 without the DMA, 28.8 without LDS reads and scaling -- the matrix pipe alone; a third slot (stage
// t + 3 issued at step t, counted vmcnt at the barrier) made it slower (37.6): what the DMA costs is not its latency.
#pragma once
#include "gemm_i8_w4tr.h"

namespace mmg {

constexpr int G4_T = 128;                               // tile edge (individuals)
constexpr int G4_TILE = G4_T * BK;                      // 16 KiB per operand tile and K step
constexpr int G4_DIG = 2 * G4_TILE;                     // offset of the digit block in a slot
constexpr int G4_BUF = G4_DIG + 1024;                   // slot: P tile, Q tile, digits [4][128]
constexpr int G4_LDS = 2 * G4_BUF;

struct StageG4 {
  __amdgpu_buffer_rsrc_t rs;
  int voff;                                             // per-lane source offset (bytes)
  int ld8;                                              // 8 * ld
};

__device__ __forceinline__ StageG4 make_stage_g4(const int8_t* base, int64_t ld, int lane) {
  StageG4 s;
  s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  const int rp = lane >> 3, cp = lane & 7;              // k row within the 8-row piece, chunk POSITION in LDS
  s.voff = rp * (int)ld + ((cp ^ (((rp >> 1) & 3) << 1)) << 4);
  s.ld8 = 8 * (int)ld;
  return s;
}

// piece i in 0..3 of this wave: k rows wave*32 + i*8 .. +8 (1 KiB of LDS)
__device__ __forceinline__ void stage_piece_g4(const StageG4& s, char* lds_tile, int wave, int i) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rs, (MMG_AS3 void*)(lds_tile + (wave * 4 + i) * 1024), 16, s.voff,
                                           (wave * 4 + i) * s.ld8, 0, 0);
}

// lane-constant part of a fragment address: operand columns tile_col0 + (lane & 31) .. , k half (lane >> 5)
__device__ __forceinline__ int frag_base_g4(int tile_col0, int lane) {
  const int i16 = lane & 15, g = lane >> 4, h = g >> 1, q = i16 >> 1;
  const int chunk = (tile_col0 >> 4) + (g & 1);
  return (h * 16 + q) * G4_T + ((chunk ^ (((q >> 1) & 3) << 1)) << 4) + (i16 & 1) * 8;
}

// fragment of a 32-column operand block for K slice `slice` (32 k rows): two transposed reads (k rows +0..7, +8..15)
__device__ __forceinline__ void lds_frag_g4(v4i& f, uint32_t addr, int slice) {
  v2i lo, hi;
  switch (slice) {
    case 0: asm volatile("ds_read_b64_tr_b8 %0, %2\n\tds_read_b64_tr_b8 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
    case 1: asm volatile("ds_read_b64_tr_b8 %0, %2 offset:4096\n\tds_read_b64_tr_b8 %1, %2 offset:5120" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
    case 2: asm volatile("ds_read_b64_tr_b8 %0, %2 offset:8192\n\tds_read_b64_tr_b8 %1, %2 offset:9216" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
    default: asm volatile("ds_read_b64_tr_b8 %0, %2 offset:12288\n\tds_read_b64_tr_b8 %1, %2 offset:13312" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
  }
  f = v4i{lo.x, lo.y, hi.x, hi.y};
}

// digits of the lane's 16 k rows (k = 32 slice + 16 (lane >> 5) ..) for the four planes; addr = slot + 16 (lane >> 5)
__device__ __forceinline__ void lds_digits_g4(v4i (&dg)[4], uint32_t addr, int slice) {
#define MMG_G4_DIGS(O)                                                                                            \
  asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t" \
               "ds_read_b128 %3, %4 offset:%8"                                                                     \
               : "=&v"(dg[0]), "=&v"(dg[1]), "=&v"(dg[2]), "=&v"(dg[3])                                            \
               : "v"(addr), "n"(G4_DIG + (O)), "n"(G4_DIG + 128 + (O)), "n"(G4_DIG + 256 + (O)), "n"(G4_DIG + 384 + (O)))
  switch (slice) {
    case 0: MMG_G4_DIGS(0); break;
    case 1: MMG_G4_DIGS(32); break;
    case 2: MMG_G4_DIGS(64); break;
    default: MMG_G4_DIGS(96); break;
  }
#undef MMG_G4_DIGS
}

// operands of one slice, ready for the MFMAs: the digit-scaled A fragments of both 32-column blocks for the four
// planes, the raw B fragments
struct OpsG4 {
  v4i as[4], at[4], b[2];
};

// bytes 0 / 1 -> 0x00 / 0xff (no carry between bytes)
__device__ __forceinline__ v4i byte_mask_g4(v4i x) { return (x + 0x7f7f7f7f) ^ 0x7f7f7f7f; }

struct G4Job {
  const int8_t* P;     // column 0 of the job's 128-column P window at k row 0 of the job
  const int8_t* Q;     // likewise for the Q window
  const int8_t* dig;   // digit of plane 0 at k row 0 of the job; plane d at + d * dig_stride
  int dig_stride;
  int nks;             // K steps of 128 rows (>= 1)
};

// raw operands of a slice as they come out of LDS
struct RawG4 {
  v4i dg[4], a[2], b[2];
};

__device__ __forceinline__ void wait_raw_g4(RawG4& r) {       // all LDS reads issued so far have landed
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r.dg[0]), "+v"(r.dg[1]), "+v"(r.dg[2]), "+v"(r.dg[3]), "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.b[0]), "+v"(r.b[1]));
}

// One slice g: 16 MFMA (4 planes x 2 x 2 tiles) on `cur` = operands of slice g; the raw fragments of slice g + 1 (`rin`,
// read during slice g - 1) become `nxt`; the raw fragments of slice g + 2 are read from (slot `src`, k-slice `slice`)
// into `rout`; with DMA_P / DMA_Q: half of the stage the cursor points at goes into slot `dst`.
// The matrix pipe takes a 32x32x32 int8 MFMA every 32 cycles = 8 issue slots per MFMA for everything else, the MFMA
// itself included; the 48 VALU operations of the scaling, the 12 LDS reads and the DMA are dealt out over the 16 gaps,
// at most 5 per gap: gaps 0-3 the byte masks of a0 / a1 (+ a DMA piece), 4-6 the LDS reads, 7-14 the eight scaled
// fragments.  (With all of the scaling at the head of the slice that consumes it, or bunched in 8 per gap, the matrix
// pipe stood at 64 % busy.)  Two slices of distance between a read and its use: ONE wait per slice, lgkmcnt(0) on entry,
// when only reads issued a whole slice ago are outstanding.
template <bool DMA_P, bool DMA_Q, bool ZERO, int ABL = 0>
__device__ __forceinline__ void g4_slice(v16i (&acc)[4][2][2], const OpsG4& cur, OpsG4& nxt, RawG4& rin, RawG4& rout,
                                         const char* src, const int (&ab)[2], const int (&bb)[2], int dgb, int slice,
                                         const StageG4& sp, const StageG4& sq, const __amdgpu_buffer_rsrc_t& rdig, int dig_voff,
                                         int dig_stride2, char* dst, int wave) {
  const uint32_t s32 = (uint32_t)(uintptr_t)src;
  const v16i zero = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v4i m0 = v4i{0, 0, 0, 0}, m1 = v4i{0, 0, 0, 0};   // (dead stores outside the ablations that skip a mask)
  if (ABL != 2) wait_raw_g4(rin);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = (i >> 2) & 1, n = i >> 3, d = i & 3;    // (as, b0) x 4, (at, b0) x 4, (as, b1) x 4, (at, b1) x 4
    acc[d][m][n] = mfma8(m ? cur.at[d] : cur.as[d], cur.b[n], ZERO ? zero : acc[d][m][n]);
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 2) {
      if (i == 0) m0 = rin.a[0] + 0x7f7f7f7f;              // bytes 0 / 1 -> 0x7f / 0x80 -> (xor) 0x00 / 0xff
      if (i == 1) m0 = m0 ^ 0x7f7f7f7f;
      if (i == 2 && ABL != 4 && ABL != 5 && ABL != 6) m1 = rin.a[1] + 0x7f7f7f7f;
      if (i == 3 && ABL != 4 && ABL != 5 && ABL != 6) m1 = m1 ^ 0x7f7f7f7f;
      if (i == 3 && (ABL == 4 || ABL == 5)) m1 = rin.a[1];  // ablations 4 / 5: half / none of the scaling VALU work
      if (i == 4) lds_digits_g4(rout.dg, s32 + (uint32_t)dgb, slice);
      if (i == 5) { lds_frag_g4(rout.a[0], s32 + (uint32_t)ab[0], slice); lds_frag_g4(rout.a[1], s32 + (uint32_t)ab[1], slice); }
      if (i == 6) { lds_frag_g4(rout.b[0], s32 + (uint32_t)bb[0], slice); lds_frag_g4(rout.b[1], s32 + (uint32_t)bb[1], slice); }
      if (i >= 7 && i <= 10) { if (ABL == 5) { if (i == 7) { nxt.as[0] = m0; nxt.as[1] = m0; nxt.as[2] = rin.dg[0]; nxt.as[3] = rin.dg[1] | rin.dg[2] | rin.dg[3]; } } else nxt.as[i - 7] = m0 & rin.dg[i - 7]; }
      if (i >= 11 && i <= 14 && ABL == 6) nxt.at[i - 11] = nxt.as[i - 11];   // ablation 6: the second row block reuses the first one's scaled operands
      else if (i >= 11 && i <= 14) { if (ABL == 4 || ABL == 5) { if (i == 11) { nxt.at[0] = m1; nxt.at[1] = m1; nxt.at[2] = m1; nxt.at[3] = m1; } } else nxt.at[i - 11] = m1 & rin.dg[i - 11]; }
      if (i == 15) { nxt.b[0] = rin.b[0]; nxt.b[1] = rin.b[1]; }
    }
    if (ABL != 1) {
      if (DMA_P && i < 4) stage_piece_g4(sp, dst, wave, i);
      if (DMA_Q && i < 4) stage_piece_g4(sq, dst + G4_TILE, wave, i);
      if (DMA_Q && wave == 0 && (i == 4 || i == 5))        // digits: [plane 2 (i - 4) + (lane >> 5)][4 (lane & 31) ..]
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(dst + G4_DIG + (i - 4) * 256), 4, dig_voff,
                                                 (i - 4) * dig_stride2, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (ABL == 2) nxt = cur;                                // ablation: no LDS reads, no scaling in the loop (wrong results)
}

// Runs one job; epi(acc): acc[d][m][n] = plane d, rows = P columns wm*64 + m*32 .., columns = Q columns wn*64 + n*32 ..
// (C layout of gemm_i8_core.h inside a 32 x 32 tile).
// ABL (timing ablations, WRONG results): 1 = no DMA in the loop, 2 = no LDS reads in the loop, 3 = no barrier in the loop
// ABL 8 + k, k = 0..3 (round 5, `make EXPERIMENTS=1`): right results + s_memtime stamps (shader clock, SGPRs) at the start of a K
// step, behind position k of it (0: slice 2, 1: slice 3, 2: the vmcnt / lgkmcnt wait + barrier, 3: slice 0 with its P pieces) and
// at its end: stamps[k] += cycles start -> position, stamps[4] += position -> end, stamps[5] = K steps, stamps[6] = the loop;
// lane 0 of every wave writes its row before the epilogue.  One position per variant: all five in one kernel cost 134 spilled
// VGPRs (round 4 met the same wall), a single one none.
template <int ABL = 0, class EpiFn>
__device__ __forceinline__ void g4_stream(const G4Job& job, int64_t ld, char* lds, EpiFn&& epi, unsigned long long* stamps = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int ab[2], bb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ab[i] = frag_base_g4(wm * 64 + i * 32, lane);
    bb[i] = frag_base_g4(wn * 64 + i * 32, lane) + G4_TILE;
  }
  const int dgb = (lane >> 5) * 16;
  const int64_t kstep_bytes = (int64_t)BK * ld;
  const int dig_voff = (lane >> 5) * job.dig_stride + (lane & 31) * 4, dig_stride2 = 2 * job.dig_stride;

  // ---- issue cursor (wave-uniform): the descriptor bases move with the stage
  int cks = 0;
  StageG4 sp = make_stage_g4(job.P, ld, lane), sq = make_stage_g4(job.Q, ld, lane);
  __amdgpu_buffer_rsrc_t rdig = __builtin_amdgcn_make_buffer_rsrc((void*)job.dig, 0, 0x7fffffff, 0x00020000);
  auto rebase = [&]() {
    sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(job.P + cks * kstep_bytes), 0, 0x7fffffff, 0x00020000);
    sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(job.Q + cks * kstep_bytes), 0, 0x7fffffff, 0x00020000);
    rdig = __builtin_amdgcn_make_buffer_rsrc((void*)(job.dig + (int64_t)cks * BK), 0, 0x7fffffff, 0x00020000);
  };
  auto advance = [&]() {
    if (cks + 1 < job.nks) { ++cks; rebase(); }           // else: stay on the last stage (harmless re-issue)
  };
  auto issue_stage = [&](char* slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece_g4(sp, slot, wave, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece_g4(sq, slot + G4_TILE, wave, i);
    if (wave == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(slot + G4_DIG), 4, dig_voff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(slot + G4_DIG + 256), 4, dig_voff, dig_stride2, 0, 0);
    }
  };

  // ---- prologue: stages 0 and 1 complete, fragments of step 0 slice 0
  issue_stage(lds);
  advance();
  issue_stage(lds + G4_BUF);
  advance();                                             // -> stage 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  OpsG4 f0, f1;
  RawG4 r0, r1;
  {
    const uint32_t l32 = (uint32_t)(uintptr_t)lds;
    lds_digits_g4(r0.dg, l32 + (uint32_t)dgb, 0);        // slice 0 -> f0
    lds_frag_g4(r0.a[0], l32 + (uint32_t)ab[0], 0);
    lds_frag_g4(r0.a[1], l32 + (uint32_t)ab[1], 0);
    lds_frag_g4(r0.b[0], l32 + (uint32_t)bb[0], 0);
    lds_frag_g4(r0.b[1], l32 + (uint32_t)bb[1], 0);
    lds_digits_g4(r1.dg, l32 + (uint32_t)dgb, 1);        // slice 1 stays raw: the first slice of the loop scales it
    lds_frag_g4(r1.a[0], l32 + (uint32_t)ab[0], 1);
    lds_frag_g4(r1.a[1], l32 + (uint32_t)ab[1], 1);
    lds_frag_g4(r1.b[0], l32 + (uint32_t)bb[0], 1);
    lds_frag_g4(r1.b[1], l32 + (uint32_t)bb[1], 1);
    wait_raw_g4(r0);
    const v4i m0 = byte_mask_g4(r0.a[0]), m1 = byte_mask_g4(r0.a[1]);
#pragma unroll
    for (int d = 0; d < 4; ++d) { f0.as[d] = m0 & r0.dg[d]; f0.at[d] = m1 & r0.dg[d]; }
    f0.b[0] = r0.b[0]; f0.b[1] = r0.b[1];
  }

  v16i acc[4][2][2];                                     // written (not accumulated) by the first slice of the job

  // Step t multiplies stage t (slot t & 1).  Slice s computes on the operands of k-slice s, scales the raw fragments of
  // k-slice s + 1 and reads those of k-slice s + 2 -- for s = 2, 3 that is k-slice 0, 1 of stage t + 1 in the other slot,
  // hence the barrier (and the vmcnt(0) for that stage) between slices 1 and 3.  Behind it every read of slot t & 1 has
  // been issued and waited for: stage t + 2 is issued into it during slices 2 and 3.
  unsigned long long seg[5] = {0, 0, 0, 0, 0};
  const unsigned long long T0 = ABL >= 8 ? __builtin_amdgcn_s_memtime() : 0;
  unsigned long long tp = T0;
#define MMG_G4_STAMP(K) do { if (ABL >= 8 && (K == 4 || K == ABL - 8)) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); seg[K] += now_ - tp; tp = now_; } } while (0)
  for (int t = 0; t < job.nks; ++t) {
    char* cur = lds + (t & 1) * G4_BUF;
    char* oth = lds + ((t + 1) & 1) * G4_BUF;
#define MMG_G4_ARGS(SRC, SL) SRC, ab, bb, dgb, SL, sp, sq, rdig, dig_voff, dig_stride2, cur, wave
    if (t == 0) g4_slice<false, false, true, ABL>(acc, f0, f1, r1, r0, MMG_G4_ARGS(cur, 2));
    else g4_slice<false, false, false, ABL>(acc, f0, f1, r1, r0, MMG_G4_ARGS(cur, 2));
    MMG_G4_STAMP(0);
    g4_slice<false, false, false, ABL>(acc, f1, f0, r0, r1, MMG_G4_ARGS(cur, 3));
    MMG_G4_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (ABL != 3) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    MMG_G4_STAMP(2);
    g4_slice<true, false, false, ABL>(acc, f0, f1, r1, r0, MMG_G4_ARGS(oth, 0));
    MMG_G4_STAMP(3);
    g4_slice<false, true, false, ABL>(acc, f1, f0, r0, r1, MMG_G4_ARGS(oth, 1));
    MMG_G4_STAMP(4);
#undef MMG_G4_ARGS
    advance();                                           // -> stage t+3
  }
#undef MMG_G4_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-issued tail stages must land before LDS is released
  wait_raw_g4(r0);                                       // ... and the raw fragments read for slices that do not exist stay
  wait_raw_g4(r1);                                       // live up to here (gemm_i8_w4tr.h: frag_drain)
  if (ABL >= 8 && stamps != nullptr && lane == 0) {
    const unsigned long long total = __builtin_amdgcn_s_memtime() - T0;
    unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = seg[k];
    o[5] = (unsigned long long)job.nks;
    o[6] = total;
  }
  epi(acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the same product with the waves as four ROW STRIPS of the 128 x 128 tile (wave w: P columns 32 w .. + 32 against
// all 128 Q columns) instead of 2 x 2 quadrants.  In the quadrant layout the two waves of a tile row scale the SAME two A
// fragments: 8 mask + 32 and-operations per slice and wave, and halving them (MMG_GRM4_ABL=6: the second A block reuses the
// first one's scaled fragments -- wrong results, same operand statistics) takes 34.1 -> 30.9 ms at C3: the slice is bound
// by what ONE wave per SIMD can issue between 16 MFMAs.  A strip has one A fragment to scale for its 16 MFMAs (4 + 16 VALU
// operations) and reads 4 raw B fragments instead of 2 (14 LDS reads per slice instead of 12).  B needs no scaling, so it is
// read ONE slice ahead straight into the other half of a register double buffer (no copies); A and the digits keep their two
// slices of lookahead (read, scale, use).  That puts the last read of a stage's Q tile (k slice 3, read in slice 2) BEHIND the
// barrier of its step, so the Q tiles rotate through THREE LDS slots (stage t + 2 lands in the slot stage t - 1 left a whole
// barrier ago); P tiles and digits keep two.  LDS: 2 x 16 + 3 x 16 KiB + 2 x 1 KiB = 82 KiB, one workgroup per CU as before.
constexpr int G4R_P = 0;                                 // P slots at 0, 16 KiB
constexpr int G4R_Q = 2 * G4_TILE;                       // Q slots at 32, 48, 64 KiB
constexpr int G4R_D = 5 * G4_TILE;                       // digit slots [4 planes][128 k] at 80 KiB, + 1 KiB
constexpr int G4R_LDS = G4R_D + 2 * 1024;

__device__ __forceinline__ void lds_digits_g4r(v4i (&dg)[4], uint32_t addr, int slice) {
#define MMG_G4R_DIGS(O)                                                                                           \
  asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t" \
               "ds_read_b128 %3, %4 offset:%8"                                                                     \
               : "=&v"(dg[0]), "=&v"(dg[1]), "=&v"(dg[2]), "=&v"(dg[3])                                            \
               : "v"(addr), "n"(O), "n"(128 + (O)), "n"(256 + (O)), "n"(384 + (O)))
  switch (slice) {
    case 0: MMG_G4R_DIGS(0); break;
    case 1: MMG_G4R_DIGS(32); break;
    case 2: MMG_G4R_DIGS(64); break;
    default: MMG_G4R_DIGS(96); break;
  }
#undef MMG_G4R_DIGS
}

struct RawG4R { v4i dg[4], a; };                         // what comes out of LDS for the A side of a slice
struct BufG4R { v4i b[4]; };                             // the raw B fragments of a slice

__device__ __forceinline__ void wait_lds_g4r(RawG4R& r, BufG4R& b) {   // every LDS read issued so far has landed
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r.dg[0]), "+v"(r.dg[1]), "+v"(r.dg[2]), "+v"(r.dg[3]), "+v"(r.a), "+v"(b.b[0]), "+v"(b.b[1]), "+v"(b.b[2]), "+v"(b.b[3]));
}

// Slice g: 16 MFMA (4 planes x 4 column blocks) on as_cur (scaled during slice g - 1) x bcur (read during slice g - 1);
// rin (raw A + digits of slice g + 1, read during g - 1) is scaled into as_nxt; the raw A + digits of slice g + 2 are read
// from (pa, da, k-slice sa) into rout, the B fragments of slice g + 1 from (qa[], k-slice sb) into bnxt.
template <bool DMA_P, bool DMA_Q, bool ZERO, int ABL = 0>
__device__ __forceinline__ void g4r_slice(v16i (&acc)[4][4], const v4i (&as_cur)[4], v4i (&as_nxt)[4], const BufG4R& bcur,
                                          BufG4R& bnxt, RawG4R& rin, RawG4R& rout, uint32_t pa, uint32_t da, int sa,
                                          const uint32_t (&qa)[4], int sb, const StageG4& sp, const StageG4& sq,
                                          const __amdgpu_buffer_rsrc_t& rdig, int dig_voff, int dig_stride2, char* dst_p,
                                          char* dst_q, char* dst_d, int wave) {
  const v16i zero = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v4i m;
  wait_lds_g4r(rin, const_cast<BufG4R&>(bcur));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = i >> 2, d = i & 3;                      // column block by column block: (b0: 4 planes), (b1: ...), ...
    acc[d][n] = mfma8(as_cur[d], bcur.b[n], ZERO ? zero : acc[d][n]);
    __builtin_amdgcn_sched_barrier(0);
    if (i < 4) lds_frag_g4(bnxt.b[i], qa[i], sb);          // B first: it is wanted one slice from now, A / digits two
    if (i == 4) m = rin.a + 0x7f7f7f7f;                   // bytes 0 / 1 -> 0x7f / 0x80; the xor below makes them 0x00 / 0xff
    if (i == 5) { if (ABL == 7) { rout.dg[0] = rin.dg[0]; rout.dg[1] = rin.dg[1]; rout.dg[2] = rin.dg[2]; rout.dg[3] = rin.dg[3]; } else lds_digits_g4r(rout.dg, da, sa); }   // ablation 7: no digit reads (wrong results)
    if (i == 6) lds_frag_g4(rout.a, pa, sa);
    if (i >= 8 && i <= 11) as_nxt[i - 8] = (m ^ 0x7f7f7f7f) & rin.dg[i - 8];
    if (DMA_P && i < 4) stage_piece_g4(sp, dst_p, wave, i);
    if (DMA_Q && i < 4) stage_piece_g4(sq, dst_q, wave, i);
    if (DMA_Q && wave == 0 && (i == 12 || i == 13))
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(dst_d + (i - 12) * 256), 4, dig_voff, (i - 12) * dig_stride2, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// epi(acc): acc[d][n] = plane d, rows = P columns wave*32 .., columns = Q columns n*32 .. (C layout of a 32 x 32 tile)
template <int ABL = 0, class EpiFn>
__device__ __forceinline__ void g4r_stream(const G4Job& job, int64_t ld, char* lds, EpiFn&& epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t l32 = (uint32_t)(uintptr_t)lds;
  const uint32_t fa = (uint32_t)frag_base_g4(wave * 32, lane);
  uint32_t fb[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) fb[n] = (uint32_t)frag_base_g4(n * 32, lane);
  const uint32_t dgb = (uint32_t)((lane >> 5) * 16);
  const int64_t kstep_bytes = (int64_t)BK * ld;
  const int dig_voff = (lane >> 5) * job.dig_stride + (lane & 31) * 4, dig_stride2 = 2 * job.dig_stride;

  // ---- issue cursor (wave-uniform): the descriptor bases move with the stage
  int cks = 0, cst = 0;                                   // K step the descriptors point at, stages issued so far
  StageG4 sp = make_stage_g4(job.P, ld, lane), sq = make_stage_g4(job.Q, ld, lane);
  __amdgpu_buffer_rsrc_t rdig = __builtin_amdgcn_make_buffer_rsrc((void*)job.dig, 0, 0x7fffffff, 0x00020000);
  auto advance = [&]() {
    ++cst;
    if (cks + 1 < job.nks) {                              // else: stay on the last stage (harmless re-issue into a free slot)
      ++cks;
      sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(job.P + cks * kstep_bytes), 0, 0x7fffffff, 0x00020000);
      sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(job.Q + cks * kstep_bytes), 0, 0x7fffffff, 0x00020000);
      rdig = __builtin_amdgcn_make_buffer_rsrc((void*)(job.dig + (int64_t)cks * BK), 0, 0x7fffffff, 0x00020000);
    }
  };
  auto p_slot = [&](int st) { return lds + G4R_P + (st & 1) * G4_TILE; };
  auto d_slot = [&](int st) { return lds + G4R_D + (st & 1) * 1024; };
  auto q_slot = [&](int st) { return lds + G4R_Q + (st % 3) * G4_TILE; };
  auto issue_stage = [&]() {                              // the whole stage `cst` at once (prologue)
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece_g4(sp, p_slot(cst), wave, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece_g4(sq, q_slot(cst), wave, i);
    if (wave == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(d_slot(cst)), 4, dig_voff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(d_slot(cst) + 256), 4, dig_voff, dig_stride2, 0, 0);
    }
  };

  // ---- prologue: stages 0 and 1 complete; raw A of k-slices 0 and 1, B of k-slice 0
  issue_stage();
  advance();
  issue_stage();
  advance();                                             // -> stage 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // LDS addresses of this lane's reads in the CURRENT slots; they move on by adds once per step (below)
  uint32_t pa = l32 + G4R_P + fa, da = l32 + G4R_D + dgb;
  uint32_t qa[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) qa[n] = l32 + G4R_Q + fb[n];
  int pst = 0, qst = 0;                                   // slot index the addresses point at (stage mod 2 / mod 3)

  v4i f0[4], f1[4];
  RawG4R r0, r1;
  BufG4R b0, b1;
  lds_digits_g4r(r0.dg, da, 0);
  lds_frag_g4(r0.a, pa, 0);
  lds_digits_g4r(r1.dg, da, 1);
  lds_frag_g4(r1.a, pa, 1);
#pragma unroll
  for (int n = 0; n < 4; ++n) lds_frag_g4(b0.b[n], qa[n], 0);
  wait_lds_g4r(r0, b0);
  {
    const v4i m = byte_mask_g4(r0.a);
#pragma unroll
    for (int d = 0; d < 4; ++d) f0[d] = m & r0.dg[d];
  }

  v16i acc[4][4];                                        // written (not accumulated) by the first slice of the job

  // Step t (stage t: P / digit slot t & 1, Q slot t % 3).  Slice s multiplies k-slice s, scales k-slice s + 1, reads raw A +
  // digits of k-slice s + 2 and B of k-slice s + 1: slices 2, 3 read A / digits of stage t + 1, slice 3 its B -- behind the
  // barrier, in front of which this wave's DMA of stage t + 1 has landed (vmcnt(0)) and every read of the P / digit slot of
  // stage t has been waited for (lgkmcnt(0)): stage t + 2's P tile and digits go into that slot in slices 2 / 3, its Q tile
  // into the slot stage t - 1 used (last read in slice 2 of step t - 1).
  for (int t = 0; t < job.nks; ++t) {
#define MMG_G4R_DMA sp, sq, rdig, dig_voff, dig_stride2, p_slot(cst), q_slot(cst), d_slot(cst), wave
    if (t == 0) g4r_slice<false, false, true, ABL>(acc, f0, f1, b0, b1, r1, r0, pa, da, 2, qa, 1, MMG_G4R_DMA);
    else g4r_slice<false, false, false, ABL>(acc, f0, f1, b0, b1, r1, r0, pa, da, 2, qa, 1, MMG_G4R_DMA);
    g4r_slice<false, false, false, ABL>(acc, f1, f0, b1, b0, r0, r1, pa, da, 3, qa, 2, MMG_G4R_DMA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {                                                     // A / digit reads move to the slot of stage t + 1
      const int dlt = pst ? -1 : 1;
      pa += (uint32_t)(dlt * G4_TILE); da += (uint32_t)(dlt * 1024); pst ^= 1;
    }
    g4r_slice<true, false, false, ABL>(acc, f0, f1, b0, b1, r1, r0, pa, da, 0, qa, 3, MMG_G4R_DMA);
    {                                                     // B reads move to the Q slot of stage t + 1
      const int dlt = qst == 2 ? -2 : 1;
#pragma unroll
      for (int n = 0; n < 4; ++n) qa[n] += (uint32_t)(dlt * G4_TILE);
      qst = qst == 2 ? 0 : qst + 1;
    }
    g4r_slice<false, true, false, ABL>(acc, f1, f0, b1, b0, r0, r1, pa, da, 1, qa, 0, MMG_G4R_DMA);
#undef MMG_G4R_DMA
    advance();                                           // -> stage t + 3
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-issued tail stages must land before LDS is released
  wait_lds_g4r(r0, b0);                                  // ... and the registers of the reads for slices that do not exist
  wait_lds_g4r(r1, b1);                                  // stay live up to here (gemm_i8_w4tr.h: frag_drain)
  epi(acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the quadrant kernel with the scaling where the compiler had put it anyway -- in the slice that CONSUMES it.
// tools/grm4_stamps.py (s_memtime stamps, one position per variant) on g4_stream: the first slice of a K step took 1124
// cycles, the other three 614-629, the wait + barrier 40.  The ISA shows why: `nxt.as[d] = mask & digit` is plain arithmetic,
// sched_barrier orders a basic block only, the wave-0 digit pieces and the stage cursor split a slice into several blocks, and
// machine sinking carried every scaled operand down to the block of the MFMA that uses it -- for slices 1-3 of a step into
// the gap in front of that MFMA (fine), for the first slice into the LOOP LATCH (its operands are loop-carried, the header
// has two predecessors): 8 v_add + 32 v_bitop3 in a row with the matrix pipe idle.  Pinning the operands where g4_slice computes
// them (an empty volatile asm per result) restores the designed pipeline and spills 149 VGPRs: two scaled + two raw operand
// sets never fitted beside the addresses; the "one slice ahead" scaling was never what ran.
// Here a slice scales its own raw operands in the gaps in front of their first use (as[0] before MFMA 0, as[1..3] in gaps
// 0-2, at[0..3] in gaps 3-6; MFMAs 8-15 reuse them), reads the raw fragments of the NEXT slice in gaps 7-9 into the other
// raw set (one slice of lookahead: 6 MFMAs = 200 cycles for 12 LDS reads) and issues its DMA pieces in gaps 10-15.  Nothing
// scaled is carried from slice to slice, so nothing can sink across a block boundary.  Slot use of a step t (stage t in
// slot t & 1): slices 0-2 read k-slices 1-3 of the stage, the barrier, slice 3 reads k-slice 0 of stage t + 1 from the other
// slot and issues the P pieces of stage t + 2 into the slot just retired; slice 0 of step t + 1 issues its Q pieces and digits.
template <bool DMA_P, bool DMA_Q, bool ZERO>
__device__ __forceinline__ void g4j_slice(v16i (&acc)[4][2][2], RawG4& cur, RawG4& nxt, const char* src, const int (&ab)[2],
                                          const int (&bb)[2], int dgb, int slice, const StageG4& sp, const StageG4& sq,
                                          const __amdgpu_buffer_rsrc_t& rdig, int dig_voff, int dig_stride2, char* dst, int wave) {
  const uint32_t s32 = (uint32_t)(uintptr_t)src;
  const v16i zero = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  wait_raw_g4(cur);                                      // the reads of `cur` were issued in gaps 7-9 of the slice before
  v4i as[4], at[4];
  const v4i m0 = byte_mask_g4(cur.a[0]);
  v4i m1 = m0;
  as[0] = m0 & cur.dg[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = (i >> 2) & 1, n = i >> 3, d = i & 3;    // (as, b0) x 4, (at, b0) x 4, (as, b1) x 4, (at, b1) x 4
    acc[d][m][n] = mfma8(m ? at[d] : as[d], cur.b[n], ZERO ? zero : acc[d][m][n]);
    __builtin_amdgcn_sched_barrier(0);
    if (i < 3) as[i + 1] = m0 & cur.dg[i + 1];
    if (i == 3) { m1 = byte_mask_g4(cur.a[1]); at[0] = m1 & cur.dg[0]; }
    if (i >= 4 && i <= 6) at[i - 3] = m1 & cur.dg[i - 3];
    if (i == 7) lds_digits_g4(nxt.dg, s32 + (uint32_t)dgb, slice);
    if (i == 8) { lds_frag_g4(nxt.a[0], s32 + (uint32_t)ab[0], slice); lds_frag_g4(nxt.a[1], s32 + (uint32_t)ab[1], slice); }
    if (i == 9) { lds_frag_g4(nxt.b[0], s32 + (uint32_t)bb[0], slice); lds_frag_g4(nxt.b[1], s32 + (uint32_t)bb[1], slice); }
    if (DMA_P && i >= 10 && i <= 13) stage_piece_g4(sp, dst, wave, i - 10);     // (four gaps apart instead -- 1, 5, 10, 13 -- no
    if (DMA_Q && i >= 10 && i <= 13) stage_piece_g4(sq, dst + G4_TILE, wave, i - 10);   //  difference: 31.09 against 30.98 ms)
    if (DMA_Q && wave == 0 && i >= 14)                    // digits: [plane 2 (i - 14) + (lane >> 5)][4 (lane & 31) ..]
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(dst + G4_DIG + (i - 14) * 256), 4, dig_voff,
                                               (i - 14) * dig_stride2, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// STAMP_AT in 0..3: s_memtime stamps (`make EXPERIMENTS=1`, tools/grm4_stamps.py) behind slice STAMP_AT and at the end of the step
template <int STAMP_AT = -1, class EpiFn>
__device__ __forceinline__ void g4j_stream(const G4Job& job, int64_t ld, char* lds, EpiFn&& epi, unsigned long long* stamps = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int ab[2], bb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ab[i] = frag_base_g4(wm * 64 + i * 32, lane);
    bb[i] = frag_base_g4(wn * 64 + i * 32, lane) + G4_TILE;
  }
  const int dgb = (lane >> 5) * 16;
  const int64_t kstep_bytes = (int64_t)BK * ld;
  const int dig_voff = (lane >> 5) * job.dig_stride + (lane & 31) * 4, dig_stride2 = 2 * job.dig_stride;

  int cks = 0;                                           // issue cursor (wave-uniform): the stage whose pieces go out next
  StageG4 sp = make_stage_g4(job.P, ld, lane), sq = make_stage_g4(job.Q, ld, lane);
  __amdgpu_buffer_rsrc_t rdig = __builtin_amdgcn_make_buffer_rsrc((void*)job.dig, 0, 0x7fffffff, 0x00020000);
  auto rebase_p = [&](int st) { sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(job.P + st * kstep_bytes), 0, 0x7fffffff, 0x00020000); };
  auto rebase_q = [&](int st) {
    sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(job.Q + st * kstep_bytes), 0, 0x7fffffff, 0x00020000);
    rdig = __builtin_amdgcn_make_buffer_rsrc((void*)(job.dig + (int64_t)st * BK), 0, 0x7fffffff, 0x00020000);
  };
  auto issue_stage = [&](char* slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece_g4(sp, slot, wave, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece_g4(sq, slot + G4_TILE, wave, i);
    if (wave == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(slot + G4_DIG), 4, dig_voff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdig, (MMG_AS3 void*)(slot + G4_DIG + 256), 4, dig_voff, dig_stride2, 0, 0);
    }
  };
  const int last = job.nks - 1;                          // a stage beyond the job is the last one again (harmless re-issue)

  // ---- prologue: stages 0 and 1 complete, raw fragments of step 0 slice 0
  issue_stage(lds);
  cks = 1 < last ? 1 : last;
  rebase_p(cks); rebase_q(cks);
  issue_stage(lds + G4_BUF);
  cks = 2 < last ? 2 : last;                             // -> stage 2: its P pieces in slice 3 of step 0, its Q pieces in slice 0 of step 1
  rebase_p(cks); rebase_q(cks);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  RawG4 r0, r1;
  {
    const uint32_t l32 = (uint32_t)(uintptr_t)lds;
    lds_digits_g4(r0.dg, l32 + (uint32_t)dgb, 0);
    lds_frag_g4(r0.a[0], l32 + (uint32_t)ab[0], 0);
    lds_frag_g4(r0.a[1], l32 + (uint32_t)ab[1], 0);
    lds_frag_g4(r0.b[0], l32 + (uint32_t)bb[0], 0);
    lds_frag_g4(r0.b[1], l32 + (uint32_t)bb[1], 0);
  }
  v16i acc[4][2][2];                                     // written (not accumulated) by the first slice of the job

  unsigned long long seg[5] = {0, 0, 0, 0, 0};
  const unsigned long long T0 = STAMP_AT >= 0 ? __builtin_amdgcn_s_memtime() : 0;
  unsigned long long tp = T0;
#define MMG_G4J_STAMP(K) do { if (STAMP_AT >= 0 && (K == 4 || K == STAMP_AT)) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); seg[K] += now_ - tp; tp = now_; } } while (0)
  for (int t = 0; t < job.nks; ++t) {
    char* cur = lds + (t & 1) * G4_BUF;
    char* oth = lds + ((t + 1) & 1) * G4_BUF;
#define MMG_G4J_ARGS(SRC, SL, DST) SRC, ab, bb, dgb, SL, sp, sq, rdig, dig_voff, dig_stride2, DST, wave
    if (t == 0) g4j_slice<false, false, true>(acc, r0, r1, MMG_G4J_ARGS(cur, 1, oth));
    else {
      g4j_slice<false, true, false>(acc, r0, r1, MMG_G4J_ARGS(cur, 1, oth));      // + Q pieces and digits of stage `cks`
      cks = cks < last ? cks + 1 : last;
      rebase_p(cks);
    }
    MMG_G4J_STAMP(0);
    g4j_slice<false, false, false>(acc, r1, r0, MMG_G4J_ARGS(cur, 2, oth));
    MMG_G4J_STAMP(1);
    g4j_slice<false, false, false>(acc, r0, r1, MMG_G4J_ARGS(cur, 3, oth));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    MMG_G4J_STAMP(2);
    g4j_slice<true, false, false>(acc, r1, r0, MMG_G4J_ARGS(oth, 0, cur));         // + P pieces of stage `cks` into the slot just retired
    rebase_q(cks);
    MMG_G4J_STAMP(4);
#undef MMG_G4J_ARGS
  }
#undef MMG_G4J_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-issued tail stages must land before LDS is released
  wait_raw_g4(r0);                                       // ... and the raw fragments read for a slice that does not exist stay
  wait_raw_g4(r1);                                       // live up to here (gemm_i8_w4tr.h: frag_drain)
  if (STAMP_AT >= 0 && stamps != nullptr && lane == 0) {
    const unsigned long long total = __builtin_amdgcn_s_memtime() - T0;
    unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = seg[k];
    o[5] = (unsigned long long)job.nks;
    o[6] = total;
  }
  epi(acc);
}

}  // namespace mmg
