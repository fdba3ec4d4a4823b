// gemm_i8_core.h -- the int8 MFMA "NT" tile mainloop shared by the IBS kinship GEMM, the EMMAX
// quadratic-form GEMM and the permutation GEMM.  gfx950 only.
//
// Geometry (one workgroup = 512 threads = 8 waves, 1 workgroup per CU):
//   output tile  TM x TN = 256 x 256 int32   (P rows x Q rows; both operands are row-major
//                                             with the contraction index k contiguous)
//   waves        2 (M) x 4 (N); wave tile 128 x 64 = 4 x 2 MFMA tiles of 32x32
//   MFMA         v_mfma_i32_32x32x32_i8 (16 B of k per lane per operand)
//   K step       BK = 128 bytes; LDS: 2 buffers x (P tile 32 KiB + Q tile 32 KiB) = 128 KiB
//   staging      global_load_lds_dwordx4 (LDS-DMA, 1 KiB = 8 rows x 128 B per wave-instruction);
//                the LDS image is lane-linear, the 16-B chunk swizzle c ^ ((row>>1)&7) is applied
//                to the per-lane SOURCE address and again on the ds_read_b128 side
//                (conflict-free for the 16-lane ds_read_b128 groups: rows distinct mod 16).
//
// C/D layout of the 32x32 MFMA (dtype independent): lane l holds column n = l & 31 and rows
// m = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), reg = 0..15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TM = 256, TN = 256, BK = 128;
constexpr int TILE_BYTES = TM * BK;            // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // P + Q
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // double buffered: 128 KiB
constexpr int NTHREADS = 512;

#define MMG_AS1 __attribute__((address_space(1)))
#define MMG_AS3 __attribute__((address_space(3)))

// Stage rows [0,256) x bytes [k0, k0+128) of a row-major int8 operand (leading dimension ld
// bytes, 16-B aligned rows) into a 32 KiB LDS tile.  Each wave moves 4 groups of 8 rows.
__device__ __forceinline__ void stage_tile(const int8_t* __restrict__ g, int64_t ld, int64_t k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int grp = wave * 4 + i;
    const int row = grp * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    const int8_t* src = g + (int64_t)row * ld + k0 + c * 16;
    __builtin_amdgcn_global_load_lds((const MMG_AS1 void*)src, (MMG_AS3 void*)(lds_tile + grp * 1024),
                                     16, 0, 0);
  }
}

__device__ __forceinline__ v4i lds_frag(const char* tile, int row, int chunk) {
  return *(const v4i*)(tile + row * BK + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// acc[m][n] += P_tile(rows wm*128 + m*32 ..) x Q_tile(rows wn*64 + n*32 ..)^T over one K step.
__device__ __forceinline__ void mma_kstep(const char* buf, int wm, int wn, int lane, v16i (&acc)[4][2]) {
  const char* pt = buf;
  const char* qt = buf + TILE_BYTES;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    v4i a[4], b[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = lds_frag(pt, wm * 128 + m * 32 + r, 2 * kk + h);
#pragma unroll
    for (int n = 0; n < 2; ++n) b[n] = lds_frag(qt, wn * 64 + n * 32 + r, 2 * kk + h);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
  }
}

// Full K loop over k-steps [ks0, ks1) (units of BK bytes).  P / Q point at row 0 of the tile's
// row range.  On return every wave has passed the final barrier (LDS free for reuse).
// ABLATE (timing experiments only; results are wrong unless 0): 1 = no staging inside the loop,
// 2 = staging only (no LDS reads, no MFMA), 3 = MFMA on registers only (no LDS reads in loop).
template <int ABLATE = 0>
__device__ __forceinline__ void gemm_tile_i8(const int8_t* __restrict__ P, int64_t ldP,
                                             const int8_t* __restrict__ Q, int64_t ldQ,
                                             int ks0, int ks1, char* lds, v16i (&acc)[4][2]) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0;
  if (ks1 <= ks0) return;
  stage_tile(P, ldP, (int64_t)ks0 * BK, lds, wave, lane);
  stage_tile(Q, ldQ, (int64_t)ks0 * BK, lds + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  v4i fa[4], fb[2];
  if (ABLATE == 3) {
#pragma unroll
    for (int m = 0; m < 4; ++m) fa[m] = lds_frag(lds, wm * 128 + m * 32 + (lane & 31), lane >> 5);
#pragma unroll
    for (int n = 0; n < 2; ++n) fb[n] = lds_frag(lds + TILE_BYTES, wn * 64 + n * 32 + (lane & 31), lane >> 5);
  }
  for (int ks = ks0; ks < ks1; ++ks) {
    if (ks + 1 < ks1 && ABLATE != 1 && ABLATE != 3) {
      char* nb = lds + (cur ^ 1) * BUF_BYTES;
      stage_tile(P, ldP, (int64_t)(ks + 1) * BK, nb, wave, lane);
      stage_tile(Q, ldQ, (int64_t)(ks + 1) * BK, nb + TILE_BYTES, wave, lane);
    }
    if (ABLATE == 3) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[m], fb[n], acc[m][n], 0, 0, 0);
    } else if (ABLATE != 2) {
      mma_kstep(lds + cur * BUF_BYTES, wm, wn, lane, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
}

}  // namespace mmg
