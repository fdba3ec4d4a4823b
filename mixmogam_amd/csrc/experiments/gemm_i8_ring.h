// gemm_i8_ring.h -- int8 MFMA tile-stream mainloop with a 4-slot LDS ring (second generation of
// gemm_i8_core.h).  A workgroup processes a SEQUENCE of 256x256 output tiles (jobs); the K-steps
// of all its tiles form one flattened software pipeline, so the LDS-DMA prefetch never drains at a
// tile boundary and the per-tile epilogue runs under the next tile's loads.
//
//   K step       64 bytes; stage = P tile (256 x 64 B = 16 KiB) + Q tile (16 KiB)
//   ring         4 stages = 128 KiB; up to 3 stages in flight (counted s_waitcnt vmcnt, never 0 in
//                steady state), one raw s_barrier per K step:
//                    wait(stage t landed) ; barrier ; issue(stage t+3 -> slot (t-1)&3) ; compute(t)
//                RAW: a stage is read one barrier after the counted wait that retires it.
//                WAR: slot (t-1)&3 is refilled only after the barrier every wave reaches after
//                     finishing step t-1's LDS reads.
//   LDS image    64-B rows, 16-B chunk swizzle c ^ ((row>>2)&3) on the DMA source address and on
//                the ds_read_b128 side (conflict-free: a 16-lane group covers 16 rows with distinct
//                (row&3, (row>>2)&3)).
//   waves        8 = 2 (M) x 4 (N), wave tile 128 x 64, v_mfma_i32_32x32x32_i8, 16 MFMA per step.
#pragma once
#include "gemm_i8_core.h"

namespace mmg {

constexpr int RBK = 64;                         // bytes of k per ring step
constexpr int RTILE = 256 * RBK;                // 16 KiB per operand per stage
constexpr int RSTAGE = 2 * RTILE;               // 32 KiB
constexpr int RSLOTS = 4;
static_assert(RSLOTS * RSTAGE == LDS_BYTES, "ring uses the same 128 KiB as the double buffer");

struct TileDesc {
  const int8_t* P;   // row 0 of the 256 P rows
  const int8_t* Q;   // row 0 of the 256 Q rows
  int nks;           // K steps of RBK bytes
};

// per-operand staging state: 1 KiB per wave-instruction = 16 rows x 64 B; each wave moves 2 pieces
// (32 rows) per operand per stage.  The swizzle (row>>2)&3 does not depend on the piece.
struct RingOp {
  __amdgpu_buffer_rsrc_t rs;
  int v;      // per-lane byte offset
  int ld16;   // 16 * ld
};

__device__ __forceinline__ RingOp make_ring_op(const int8_t* base, int64_t ld, int wave, int lane) {
  RingOp s;
  s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  const int base_row = wave * 32 + (lane >> 2);
  const int c = (lane & 3) ^ ((base_row >> 2) & 3);
  s.v = base_row * (int)ld + c * 16;
  s.ld16 = 16 * (int)ld;
  return s;
}

__device__ __forceinline__ void ring_stage_piece(const RingOp& s, int k0, char* lds_tile, int wave, int i) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rs, (MMG_AS3 void*)(lds_tile + (wave * 2 + i) * 1024), 16, s.v,
                                           k0 + i * s.ld16, 0, 0);
}

__device__ __forceinline__ v4i ring_frag(const char* tile, int row, int chunk) {
  return *(const v4i*)(tile + row * RBK + ((chunk ^ ((row >> 2) & 3)) << 4));
}

// tile(i) -> TileDesc for i in [0, ntiles); epi(i, acc) consumes the finished accumulators.
template <class TileFn, class EpiFn>
__device__ __forceinline__ void run_tiles_ring(int ntiles, int64_t ldP, int64_t ldQ, char* lds, TileFn tile,
                                               EpiFn epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
  int total = 0;
  for (int i = 0; i < ntiles; ++i) total += tile(i).nks;
  if (total == 0) return;

  // issue cursor
  int it = 0, iks = 0, issued = 0;
  TileDesc id = tile(0);
  while (id.nks == 0 && it + 1 < ntiles) id = tile(++it);
  RingOp sp = make_ring_op(id.P, ldP, wave, lane);
  RingOp sq = make_ring_op(id.Q, ldQ, wave, lane);
  auto issue_one = [&]() {
    if (issued >= total) return;
    char* slot = lds + (issued & (RSLOTS - 1)) * RSTAGE;
    ring_stage_piece(sp, iks * RBK, slot, wave, 0);
    ring_stage_piece(sp, iks * RBK, slot, wave, 1);
    ring_stage_piece(sq, iks * RBK, slot + RTILE, wave, 0);
    ring_stage_piece(sq, iks * RBK, slot + RTILE, wave, 1);
    ++issued;
    if (++iks == id.nks) {
      iks = 0;
      do { ++it; if (it < ntiles) id = tile(it); } while (it < ntiles && id.nks == 0);
      if (it < ntiles) {
        sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)id.P, 0, 0x7fffffff, 0x00020000);
        sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)id.Q, 0, 0x7fffffff, 0x00020000);
      }
    }
  };
  issue_one(); issue_one(); issue_one();

  int t = 0;
  for (int ct = 0; ct < ntiles; ++ct) {
    const TileDesc cd = tile(ct);
    if (cd.nks == 0) continue;
    v16i acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0;
    for (int ks = 0; ks < cd.nks; ++ks, ++t) {
      const int after = total - 1 - t;              // steps still to come after this one
      if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_one();                                   // stage t+3 -> slot (t+3)&3 == (t-1)&3
      const char* pt = lds + (t & (RSLOTS - 1)) * RSTAGE;
      const char* qt = pt + RTILE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        v4i a[4], b[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = ring_frag(pt, wm * 128 + m * 32 + r, 2 * kk + h);
#pragma unroll
        for (int n = 0; n < 2; ++n) b[n] = ring_frag(qt, wn * 64 + n * 32 + r, 2 * kk + h);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
      }
    }
    epi(ct, acc);
  }
  // every wave has consumed its last stage; make the LDS reusable by the caller
  __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------
// Flattened tile stream over the 2-slot / 128-byte K-step layout of gemm_i8_core.h: one barrier per
// 32 MFMA per wave; the first stage of the next tile is already in flight while the epilogue of the
// current tile runs.  hook(ct, ks, nks, stage_ptr) is called once per K step while that step's
// stage is readable in LDS (the scan captures its epilogue operands from the Q tile there).
template <class TileFn, class HookFn, class EpiFn>
__device__ __forceinline__ void run_tiles_flat2(int ntiles, int64_t ldP, int64_t ldQ, char* lds, TileFn tile,
                                                HookFn hook, EpiFn epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int total = 0;
  for (int i = 0; i < ntiles; ++i) total += tile(i).nks;
  if (total == 0) return;
  int it = 0, iks = 0, issued = 0;
  TileDesc id = tile(0);
  while (id.nks == 0 && it + 1 < ntiles) id = tile(++it);
  StageOp sp = make_stage_op(id.P, ldP, wave, lane);
  StageOp sq = make_stage_op(id.Q, ldQ, wave, lane);
  auto issue_one = [&]() {
    if (issued >= total) return;
    char* slot = lds + (issued & 1) * BUF_BYTES;
    stage_tile(sp, iks * BK, slot, wave);
    stage_tile(sq, iks * BK, slot + TILE_BYTES, wave);
    ++issued;
    if (++iks == id.nks) {
      iks = 0;
      do { ++it; if (it < ntiles) id = tile(it); } while (it < ntiles && id.nks == 0);
      if (it < ntiles) {
        sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)id.P, 0, 0x7fffffff, 0x00020000);
        sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)id.Q, 0, 0x7fffffff, 0x00020000);
      }
    }
  };
  issue_one();
  int t = 0;
  for (int ct = 0; ct < ntiles; ++ct) {
    const TileDesc cd = tile(ct);
    if (cd.nks == 0) continue;
    v16i acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0;
    for (int ks = 0; ks < cd.nks; ++ks, ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage t landed (the only stage in flight)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_one();                                        // stage t+1 -> the slot step t-1 used
      const char* cur = lds + (t & 1) * BUF_BYTES;
      hook(ct, ks, cd.nks, cur);
      mma_kstep(cur, wm, wn, lane, acc);
    }
    epi(ct, acc);
  }
  __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------
// Ping-pong variant: the two waves that share a SIMD (w and w+4) run one barrier apart, so that
// while one of them is in its MFMA section the other issues its LDS reads and LDS-DMA pieces.
// Per phase (one 64-byte ring step, 16 MFMA): R: 12 ds_read_b128 + 4 DMA pieces + counted vmcnt
//                                               + lgkmcnt(0)
//                                            barrier ; 16 MFMA (s_setprio 1) ; barrier
// Waves 4-7 execute one extra barrier up front and waves 0-3 one at the end.
//   RAW: the counted vmcnt that retires stage t+1 sits in the R section of the last phase of step
//        t; the first reads of stage t+1 are one full phase later, after every wave has passed a
//        barrier that follows its own wait.
//   WAR: stage t+3 is written into the slot of stage t-1 during step t; every wave retired its
//        reads of stage t-1 (lgkmcnt(0) before a barrier) at least one barrier earlier.
template <class TileFn, class EpiFn>
__device__ __forceinline__ void run_tiles_pingpong(int ntiles, int64_t ldP, int64_t ldQ, char* lds, TileFn tile,
                                                   EpiFn epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
  int total = 0;
  for (int i = 0; i < ntiles; ++i) total += tile(i).nks;
  if (total == 0) return;

  int it = 0, iks = 0, issued = 0;
  TileDesc id = tile(0);
  while (id.nks == 0 && it + 1 < ntiles) id = tile(++it);
  RingOp sp = make_ring_op(id.P, ldP, wave, lane);
  RingOp sq = make_ring_op(id.Q, ldQ, wave, lane);
  // half 0: the two P pieces of stage `issued`; half 1: the two Q pieces, then advance the cursor
  auto issue_half = [&](int half) {
    if (issued >= total) return;
    char* slot = lds + (issued & (RSLOTS - 1)) * RSTAGE;
    if (half == 0) {
      ring_stage_piece(sp, iks * RBK, slot, wave, 0);
      ring_stage_piece(sp, iks * RBK, slot, wave, 1);
    } else {
      ring_stage_piece(sq, iks * RBK, slot + RTILE, wave, 0);
      ring_stage_piece(sq, iks * RBK, slot + RTILE, wave, 1);
      ++issued;
      if (++iks == id.nks) {
        iks = 0;
        do { ++it; if (it < ntiles) id = tile(it); } while (it < ntiles && id.nks == 0);
        if (it < ntiles) {
          sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)id.P, 0, 0x7fffffff, 0x00020000);
          sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)id.Q, 0, 0x7fffffff, 0x00020000);
        }
      }
    }
  };
  for (int i = 0; i < 3; ++i) { issue_half(0); issue_half(1); }
  if (total >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (total == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();       // waves 4-7 run one barrier behind

  int t = 0;
  for (int ct = 0; ct < ntiles; ++ct) {
    const TileDesc cd = tile(ct);
    if (cd.nks == 0) continue;
    v16i acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0;
    for (int ks = 0; ks < cd.nks; ++ks, ++t) {
      const char* pt = lds + (t & (RSLOTS - 1)) * RSTAGE;
      const char* qt = pt + RTILE;
      // ---- R section: all fragments of this 64-byte step
      v4i a[2][4], b[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int m = 0; m < 4; ++m) a[kk][m] = ring_frag(pt, wm * 128 + m * 32 + r, 2 * kk + h);
#pragma unroll
        for (int n = 0; n < 2; ++n) b[kk][n] = ring_frag(qt, wn * 64 + n * 32 + r, 2 * kk + h);
      }
      issue_half(0);
      issue_half(1);                                  // stage t+3 -> slot (t-1)&3
      if (t + 1 < total) {                            // retire stage t+1 for the next step's reads
        const int keep = total - 2 - t;               // stages allowed to stay in flight (t+2, t+3)
        if (keep >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (keep == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // ---- MFMA section
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kk][m], b[kk][n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    epi(ct, acc);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();         // balance the extra barrier of waves 4-7
  __builtin_amdgcn_s_barrier();
}

}  // namespace mmg
