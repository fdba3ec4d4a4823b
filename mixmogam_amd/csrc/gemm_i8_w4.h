// gemm_i8_w4.h -- experiment: 4 waves per workgroup (one per SIMD), wave tile 128 x 128, flattened
// tile stream over the 2-slot / 128-byte K-step LDS layout of gemm_i8_core.h.
// Accumulators: 4 x 4 x 16 = 256 registers per lane (the unified 512-entry VGPR/AGPR file of one
// wave per SIMD).  LDS reads per K step drop from 192 to 128 wave-instructions and only 4 waves
// meet at the barrier.
#pragma once
#include "gemm_i8_core.h"
#include "gemm_i8_ring.h"

namespace mmg {

constexpr int W4_THREADS = 256;

struct StageOp4 {
  __amdgpu_buffer_rsrc_t rs;
  int v_even, v_odd, ld8;
};

__device__ __forceinline__ StageOp4 make_stage_op4(const int8_t* base, int64_t ld, int wave, int lane) {
  StageOp4 s;
  s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  const int base_row = wave * 64 + (lane >> 3);
  const int c0 = (lane & 7) ^ ((base_row >> 1) & 7);
  s.v_even = base_row * (int)ld + c0 * 16;
  s.v_odd = base_row * (int)ld + (c0 ^ 4) * 16;
  s.ld8 = 8 * (int)ld;
  return s;
}

// piece i in 0..7: rows (wave*8 + i)*8 .. +8
__device__ __forceinline__ void stage_piece4(const StageOp4& s, int k0, char* lds_tile, int wave, int i) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rs, (MMG_AS3 void*)(lds_tile + (wave * 8 + i) * 1024), 16,
                                           (i & 1) ? s.v_odd : s.v_even, k0 + i * s.ld8, 0, 0);
}

template <class TileFn, class EpiFn>
__device__ __forceinline__ void run_tiles_w4(int ntiles, int64_t ldP, int64_t ldQ, char* lds, TileFn tile, EpiFn epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int total = 0;
  for (int i = 0; i < ntiles; ++i) total += tile(i).nks;
  if (total == 0) return;

  int it = 0, iks = 0, issued = 0;
  TileDesc id = tile(0);
  while (id.nks == 0 && it + 1 < ntiles) id = tile(++it);
  StageOp4 sp = make_stage_op4(id.P, ldP, wave, lane);
  StageOp4 sq = make_stage_op4(id.Q, ldQ, wave, lane);
  auto issue_one = [&]() {
    if (issued >= total) return;
    char* slot = lds + (issued & 1) * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_piece4(sp, iks * BK, slot, wave, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_piece4(sq, iks * BK, slot + TILE_BYTES, wave, i);
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
    v16i acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0;
    for (int ks = 0; ks < cd.nks; ++ks, ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage t landed (only stage in flight)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_one();                                        // stage t+1 -> the slot step t-1 used
      const char* pt = lds + (t & 1) * BUF_BYTES;
      const char* qt = pt + TILE_BYTES;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        v4i a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = lds_frag(pt, wm * 128 + m * 32 + r, 2 * kk + h);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = lds_frag(qt, wn * 128 + n * 32 + r, 2 * kk + h);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
      }
    }
    epi(ct, acc);
  }
  __builtin_amdgcn_s_barrier();
}

}  // namespace mmg
