// gemm_i8_core.h -- the int8 MFMA "NT" tile mainloop shared by the IBS kinship GEMM, the EMMAX
// quadratic-form GEMM and the permutation GEMM.  gfx950 only.
//
// Geometry (one workgroup = 512 threads = 8 waves, 1 workgroup per CU):
//   output tile  TM x TN = 256 x 256 int32   (P rows x Q rows; both operands are row-major
//                                             with the contraction index k contiguous)
//   waves        2 (M) x 4 (N); wave tile 128 x 64 = 4 x 2 MFMA tiles of 32x32
//   MFMA         v_mfma_i32_32x32x32_i8 (16 B of k per lane per operand)
//   K step       BK = 128 bytes; LDS: 2 buffers x (P tile 32 KiB + Q tile 32 KiB) = 128 KiB
//   staging      buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB = 8 rows x 128 B per wave-instruction):
//                wave-uniform descriptor + SGPR offset (k, row group) + ONE per-lane VGPR offset that
//                is constant for the whole tile, so a piece costs two SALU ops and one VMEM issue.
//                The LDS image is lane-linear; the 16-B chunk swizzle c ^ ((row>>1)&7) is applied
//                to the per-lane SOURCE offset and again on the ds_read_b128 side
//                (conflict-free for the 16-lane ds_read_b128 groups: rows distinct mod 16).
//
// C/D layout of the 32x32 MFMA (dtype independent): lane l holds column n = l & 31 and rows
// m = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), reg = 0..15.
//
// Addressing limit: 256 * ld < 2^31 bytes per operand tile (32-bit buffer offsets); callers chunk
// the contraction axis accordingly (kinship: <= 4M SNPs per pass).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TM = 256, TN = 256, BK = 128;
// Digit planes of the EMMAX scan matrix (k_scan.hip:quantize_kernel): UNSIGNED 7-bit digits of the entries shifted
// into the non-negative range.  On this power-capped part the int8 matrix pipe sustains 4.56 POP/s on non-negative
// digit bytes against 4.32 on balanced (signed) ones of any width (tools/probe/mfma_digit_range.hip; two's-complement
// negatives toggle the sign-extension bits all the way up the adder tree), and the scan GEMM is at the cap: the same
// kernel ran 7 % faster per plane (33.9 -> 31.6 ms, all planes) on non-negative planes.
constexpr int SCAN_DIGIT_BITS = 7;
constexpr int TILE_BYTES = TM * BK;            // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // P + Q
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // double buffered: 128 KiB
constexpr int NTHREADS = 512;
constexpr int64_t MAX_LD = (int64_t(1) << 31) / 256 - 1;

#define MMG_AS1 __attribute__((address_space(1)))
#define MMG_AS3 __attribute__((address_space(3)))

// Per-operand staging state of one wave: buffer descriptor of the tile's 256 rows and the two
// per-lane byte offsets (even / odd 8-row group: the swizzle differs by chunk ^ 4).
struct StageOp {
  __amdgpu_buffer_rsrc_t rs;
  int v_even, v_odd;   // lane offsets (bytes) for row groups wave*4 + {0,2} / {1,3}
  int ld8;             // 8 * ld
};

__device__ __forceinline__ StageOp make_stage_op(const int8_t* base, int64_t ld, int wave, int lane) {
  StageOp s;
  s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  const int base_row = wave * 32 + (lane >> 3);
  const int c0 = (lane & 7) ^ ((base_row >> 1) & 7);
  s.v_even = base_row * (int)ld + c0 * 16;
  s.v_odd = base_row * (int)ld + (c0 ^ 4) * 16;
  s.ld8 = 8 * (int)ld;
  return s;
}

// piece i in 0..3: rows (wave*4 + i)*8 .. +8 of the operand tile, bytes [k0, k0+128)
__device__ __forceinline__ void stage_piece(const StageOp& s, int k0, char* lds_tile, int wave, int i) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rs, (MMG_AS3 void*)(lds_tile + (wave * 4 + i) * 1024), 16,
                                           (i & 1) ? s.v_odd : s.v_even, k0 + i * s.ld8, 0, 0);
}

__device__ __forceinline__ void stage_tile(const StageOp& s, int k0, char* lds_tile, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) stage_piece(s, k0, lds_tile, wave, i);
}

// Loader-wave form: wave w < 4 moves its own 32 rows AND the 32 rows of wave w+4 (rows +128: same
// per-lane offsets, the row displacement goes into the scalar offset), so that waves 4-7 issue no DMA.
__device__ __forceinline__ void stage_tile_pair(const StageOp& s, int k0, char* lds_tile, int wave) {
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rs, (MMG_AS3 void*)(lds_tile + ((wave + 4 * half) * 4 + i) * 1024), 16,
                                               (i & 1) ? s.v_odd : s.v_even, k0 + (i + 16 * half) * s.ld8, 0, 0);
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

// ---- 16x16x64 MFMA flavour of the same wave tile (128 x 64 = 8 x 4 tiles of 16 x 16) ----------
// A/B: lane l holds row l&15, k = 16*(l>>4) + j (j < 16) of a 64-byte k slice; C/D: column l&15,
// rows 4*(l>>4) + reg (reg < 4).  Same LDS bytes per K step as the 32x32x32 form.
__device__ __forceinline__ void mma_kstep_16(const char* buf, int wm, int wn, int lane, v4i (&acc)[8][4]) {
  const char* pt = buf;
  const char* qt = buf + TILE_BYTES;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    v4i b[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) b[n] = lds_frag(qt, wn * 64 + n * 16 + r, 4 * kk + g);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v4i a = lds_frag(pt, wm * 128 + m * 16 + r, 4 * kk + g);
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[n], acc[m][n], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void gemm_tile_i8_16(const int8_t* __restrict__ P, int64_t ldP,
                                                const int8_t* __restrict__ Q, int64_t ldQ, int ks0, int ks1,
                                                char* lds, v4i (&acc)[8][4]) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][n][i] = 0;
  if (ks1 <= ks0) return;
  const StageOp sp = make_stage_op(P, ldP, wave, lane);
  const StageOp sq = make_stage_op(Q, ldQ, wave, lane);
  stage_tile(sp, ks0 * BK, lds, wave);
  stage_tile(sq, ks0 * BK, lds + TILE_BYTES, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int ks = ks0; ks < ks1; ++ks) {
    char* nb = lds + (cur ^ 1) * BUF_BYTES;
    if (ks + 1 < ks1) {
      stage_tile(sp, (ks + 1) * BK, nb, wave);
      stage_tile(sq, (ks + 1) * BK, nb + TILE_BYTES, wave);
    }
    mma_kstep_16(lds + cur * BUF_BYTES, wm, wn, lane, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
}

// In-kernel stamp (diagnostic builds only): shader clock, with the lgkmcnt(0) the guide prescribes.
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// Diagnostic twin of gemm_tile_i8: same loop, accumulates per-wave cycle sums of the four segments
// of a K step into seg[0..3] = {issue DMA, LDS reads + MFMA issue, vmcnt(0) wait, barrier wait}.
__device__ __forceinline__ void gemm_tile_i8_timed(const int8_t* __restrict__ P, int64_t ldP,
                                                   const int8_t* __restrict__ Q, int64_t ldQ, int ks0, int ks1,
                                                   char* lds, v16i (&acc)[4][2], unsigned long long (&seg)[5]) {
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
  const StageOp sp = make_stage_op(P, ldP, wave, lane);
  const StageOp sq = make_stage_op(Q, ldQ, wave, lane);
  unsigned long long tp = stamp();
  stage_tile(sp, ks0 * BK, lds, wave);
  stage_tile(sq, ks0 * BK, lds + TILE_BYTES, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  seg[4] += stamp() - tp;                      // prologue
  int cur = 0;
  for (int ks = ks0; ks < ks1; ++ks) {
    char* nb = lds + (cur ^ 1) * BUF_BYTES;
    const unsigned long long t0 = stamp();
    if (ks + 1 < ks1 && wave < 4) {                 // production staging: loader waves only
      stage_tile_pair(sp, (ks + 1) * BK, nb, wave);
      stage_tile_pair(sq, (ks + 1) * BK, nb + TILE_BYTES, wave);
    }
    const unsigned long long t1 = stamp();
    mma_kstep(lds + cur * BUF_BYTES, wm, wn, lane, acc);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(acc[m][n]));   // pin the MFMAs before the stamp
    const unsigned long long t2 = stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = stamp();
    __syncthreads();
    const unsigned long long t4 = stamp();
    seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3;
    cur ^= 1;
  }
}

// Full K loop over k-steps [ks0, ks1) (units of BK bytes).  P / Q point at row 0 of the tile's
// row range.  On return every wave has passed the final barrier (LDS free for reuse).
// MODE 5 (default): only waves 0-3 issue the LDS-DMA of the next stage -- for their own rows and
//   for the rows of their SIMD partner (wave w+4) -- so that waves 4-7 start their MFMAs right after
//   the barrier while the loader waves are still queueing pieces behind the CU's single address
//   path (in-kernel stamps: with all 8 waves loading, every wave spent ~650 of ~3000 cycles per
//   K step issuing DMA with the MFMA pipe idle).  -4...5 % kernel time.
// MODE 0: every wave stages its own rows (first version).
// MODE 1/2/3 are timing ablations with WRONG results: 1 = no staging inside the loop, 2 = staging
//   only (no LDS reads, no MFMA), 3 = MFMA on registers only.
template <int ABLATE = 5>
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
  const StageOp sp = make_stage_op(P, ldP, wave, lane);
  const StageOp sq = make_stage_op(Q, ldQ, wave, lane);
  stage_tile(sp, ks0 * BK, lds, wave);
  stage_tile(sq, ks0 * BK, lds + TILE_BYTES, wave);
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
    char* nb = lds + (cur ^ 1) * BUF_BYTES;
    if (ABLATE == 5) {
      if (ks + 1 < ks1 && wave < 4) {
        stage_tile_pair(sp, (ks + 1) * BK, nb, wave);
        stage_tile_pair(sq, (ks + 1) * BK, nb + TILE_BYTES, wave);
      }
    } else if (ks + 1 < ks1 && ABLATE != 1 && ABLATE != 3) {
      stage_tile(sp, (ks + 1) * BK, nb, wave);
      stage_tile(sq, (ks + 1) * BK, nb + TILE_BYTES, wave);
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
