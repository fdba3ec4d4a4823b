// k_kinship.hip -- the kinship GEMMs K = X X^T over the individual-major genotype image
// Xt [Npad x Mk] (k = SNP index contiguous).  Replaces kinship.py:29-44 (IBS) and
// kinship.py:63-69 / hdf5_data.py:99-106 (GRM).
//
//  * kinship_i8_kernel : exact IBS counts on the int8 matrix cores (v_mfma_i32_32x32x32_i8),
//                        split-K over SNPs, int32 atomics (order independent -> bit reproducible).
//  * kinship_f32_kernel: dense fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32) with the int8 genotypes
//                        expanded to fp32 from the LDS tile through a per-SNP affine
//                        x = scale*s + shift; per-(tile, K-split) fp32 slabs, reduced in fp64 in
//                        fixed order (deterministic).
// Only upper-triangular tiles (I <= J) are computed.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "gemm_i8_core.h"
#include "gemm_i8_w4s.h"
#include "gemm_i8_grm4.h"
#include "mmg_internal.h"

namespace mmg {

struct KinJob { int I, J, ks0, ks1, slab, pad0, pad1, pad2; };

__device__ __forceinline__ int xcd_job_index(int b) {
  // blocks b and b+8 share an XCD (observed round-robin placement; speed only).  Give each XCD
  // 32 consecutive jobs of every chunk of 256 so that neighbouring tiles share an L2.
  return (b & ~255) + (b & 7) * 32 + ((b >> 3) & 31);
}

// Xp == Xq: IBS / indicator counts X X'.  Xp != Xq: one digit plane of the weighted Gram matrix of the exact GRM
// (rows of tile I from the digit image, rows of tile J from the plain image; the product is symmetric all the same).
__global__ __launch_bounds__(NTHREADS, 2) void kinship_i8_kernel(const int8_t* __restrict__ Xp,
                                                                 const int8_t* __restrict__ Xq, int64_t Mk,
                                                                 int32_t Npad, const KinJob* __restrict__ jobs,
                                                                 int* __restrict__ C32) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  v16i acc[4][2];
  gemm_tile_i8(Xp + (int64_t)job.I * TM * Mk, Mk, Xq + (int64_t)job.J * TN * Mk, Mk, job.ks0, job.ks1, lds,
               acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = job.J * TN + wn * 64 + n * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = job.I * TM + wm * 128 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        atomicAdd(C32 + (int64_t)row * Npad + col, acc[m][n][i]);
      }
    }
}

// The same job on the 4-wave pipeline of gemm_i8_w4s.h (one job per workgroup: a K-split slice of one tile).
__global__ __launch_bounds__(W4_THREADS) void kinship_i8_w4_kernel(const int8_t* __restrict__ Xp,
                                                                   const int8_t* __restrict__ Xq, int64_t Mk,
                                                                   int32_t Npad, const KinJob* __restrict__ jobs,
                                                                   int* __restrict__ C32) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  // (a K-step-major image [Mk / 128][Npad][128] -- 32 KB contiguous per stage instead of 256 rows 1 MB apart -- was
  // timed with this kernel in round 2: no difference, the row-major image stays)
  const int8_t* P = Xp + (int64_t)job.I * TM * Mk + (int64_t)job.ks0 * BK;
  const int8_t* Q = Xq + (int64_t)job.J * TN * Mk + (int64_t)job.ks0 * BK;
  const int nks = job.ks1 - job.ks0;
  w4s_stream(
      0, 1, Mk, Mk, lds, [&](int) { return W4Job{P, Q, nks}; }, [](int) {},
      [&](int, v16i (&acc)[4][4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int col = job.J * TN + wn * 128 + n * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = job.I * TM + wm * 128 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
              atomicAdd(C32 + (int64_t)row * Npad + col, acc[m][n][i]);
            }
          }
      });
}

// The same job straight off SNP-major images (the genotype store itself): both tiles are 256-column windows of rows
// [ks0*128, ks1*128), read through the transposed LDS reads of gemm_i8_w4tr.h -- no individual-major copy.
// Sp != Sq: digit-weighted rows x plain rows (exact GRM planes).
template <int N3, int PFD>
__global__ __launch_bounds__(W4_THREADS) void kinship_i8_tr_kernel(const int8_t* __restrict__ Sp,
                                                                   const int8_t* __restrict__ Sq, int64_t ld,
                                                                   int32_t Npad, const KinJob* __restrict__ jobs,
                                                                   int* __restrict__ C32) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  const int8_t* P = Sp + (int64_t)job.ks0 * BK * ld + (int64_t)job.I * TM;
  const int8_t* Q = Sq + (int64_t)job.ks0 * BK * ld + (int64_t)job.J * TN;
  const int nks = job.ks1 - job.ks0;
  w4tr_stream<N3, PFD>(
      0, 1, ld, lds, [&](int) { return W4JobTr{P, Q, nks}; }, [](int) {},
      [&](int, v16i (&acc)[4][4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int col = job.J * TN + wn * 128 + n * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = job.I * TM + wm * 128 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
              atomicAdd(C32 + (int64_t)row * Npad + col, acc[m][n][i]);
            }
          }
      });
}

// All four digit planes of the exact GRM of a binary store in one pass over the genotypes (gemm_i8_grm4.h): job = a
// 128 x 128 tile pair (I, J) of individuals over SNP rows [ks0*128, ks1*128); C32[d] += (diag(dig_d) S)' S on that tile.
template <int ABL>
__global__ __launch_bounds__(W4_THREADS) void kinship_grm4_kernel(const int8_t* __restrict__ S, int64_t ld, int32_t Npad,
                                                                  const int8_t* __restrict__ dig, int dig_stride,
                                                                  const KinJob* __restrict__ jobs, int* __restrict__ C32,
                                                                  unsigned long long* __restrict__ stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  G4Job gj;
  gj.P = S + (int64_t)job.ks0 * BK * ld + (int64_t)job.I * G4_T;
  gj.Q = S + (int64_t)job.ks0 * BK * ld + (int64_t)job.J * G4_T;
  gj.dig = dig + (int64_t)job.ks0 * BK;
  gj.dig_stride = dig_stride;
  gj.nks = job.ks1 - job.ks0;
  const int64_t plane = (int64_t)Npad * Npad;
  g4_stream<ABL>(gj, ld, lds, [&](v16i (&acc)[4][2][2]) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int col = job.J * G4_T + wn * 64 + n * 32 + r;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = job.I * G4_T + wm * 64 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(C32 + d * plane + (int64_t)row * Npad + col, acc[d][m][n][i]);
          }
        }
  }, stamps);
}

// The quadrant layout with every slice scaling its own operands (gemm_i8_grm4.h, round 5: g4j_stream); STAMP_AT >= 0: + stamps.
template <int STAMP_AT>
__global__ __launch_bounds__(W4_THREADS) void kinship_grm4j_kernel(const int8_t* __restrict__ S, int64_t ld, int32_t Npad,
                                                                   const int8_t* __restrict__ dig, int dig_stride,
                                                                   const KinJob* __restrict__ jobs, int* __restrict__ C32,
                                                                   unsigned long long* __restrict__ stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  G4Job gj;
  gj.P = S + (int64_t)job.ks0 * BK * ld + (int64_t)job.I * G4_T;
  gj.Q = S + (int64_t)job.ks0 * BK * ld + (int64_t)job.J * G4_T;
  gj.dig = dig + (int64_t)job.ks0 * BK;
  gj.dig_stride = dig_stride;
  gj.nks = job.ks1 - job.ks0;
  const int64_t plane = (int64_t)Npad * Npad;
  g4j_stream<STAMP_AT>(gj, ld, lds, [&](v16i (&acc)[4][2][2]) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int col = job.J * G4_T + wn * 64 + n * 32 + r;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = job.I * G4_T + wm * 64 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(C32 + d * plane + (int64_t)row * Npad + col, acc[d][m][n][i]);
          }
        }
  }, stamps);
}

// The same product with the four waves as row strips of the tile (gemm_i8_grm4.h, round 4): half the digit-scaling VALU work.
template <int ABL>
__global__ __launch_bounds__(W4_THREADS) void kinship_grm4r_kernel(const int8_t* __restrict__ S, int64_t ld, int32_t Npad,
                                                                   const int8_t* __restrict__ dig, int dig_stride,
                                                                   const KinJob* __restrict__ jobs, int* __restrict__ C32) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  G4Job gj;
  gj.P = S + (int64_t)job.ks0 * BK * ld + (int64_t)job.I * G4_T;
  gj.Q = S + (int64_t)job.ks0 * BK * ld + (int64_t)job.J * G4_T;
  gj.dig = dig + (int64_t)job.ks0 * BK;
  gj.dig_stride = dig_stride;
  gj.nks = job.ks1 - job.ks0;
  const int64_t plane = (int64_t)Npad * Npad;
  g4r_stream<ABL>(gj, ld, lds, [&](v16i (&acc)[4][4]) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int col = job.J * G4_T + n * 32 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = job.I * G4_T + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          atomicAdd(C32 + d * plane + (int64_t)row * Npad + col, acc[d][n][i]);
        }
      }
  });
}

// The raw-genotype product of a BINARY store on FP4 operands (gemm_i8_w4tr.h FmtF4): v_mfma_scale_f32_32x32x64_f8f6f4
// runs 0/1 x 0/1 at 9.05 POP/s under the power cap against 4.85 for int8 bytes (tools/probe/mfma_f8f6f4_rate.hip) and
// every LDS fill carries half the bytes -- both bounds of the int8 kernel move.  X4: the FP4 image
// (pack_fp4_kernel), row stride ld4 = Npad / 2; a K step is 256 SNP rows.  The fp32 accumulators of a job hold exact
// counts (< 2^24: the host bounds the K range per job).
// ABL: 0 = production; 1-3: the stream's timing ablations (gemm_i8_w4tr.h), 4 = the epilogue writes nothing -- WRONG results, `make
// EXPERIMENTS=1` builds only (MMG_F4_ABL; tools/kin_sweep.py --abl).
template <int ABL = 0, int N3 = 8, int PFD = 0>
__global__ __launch_bounds__(W4_THREADS) void kinship_f4_tr_kernel(const int8_t* __restrict__ X4, int64_t ld4,
                                                                   int32_t Npad, const KinJob* __restrict__ jobs,
                                                                   int* __restrict__ C32) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  const int8_t* P = X4 + (int64_t)job.ks0 * FmtF4::KROWS * ld4 + (int64_t)job.I * (TM / 2);
  const int8_t* Q = X4 + (int64_t)job.ks0 * FmtF4::KROWS * ld4 + (int64_t)job.J * (TN / 2);
  const int nks = job.ks1 - job.ks0;
  w4tr_stream<N3, PFD, FmtF4, (ABL >= 1 && ABL <= 3) ? ABL : 0>(
      0, 1, ld4, lds, [&](int) { return W4JobTr{P, Q, nks}; }, [](int) {},
      [&](int, v16f (&acc)[4][4]) {
        if (ABL == 4) {                                        // one atomic per lane instead of 256: the accumulators stay live
          float t = 0.f;
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
              for (int i = 0; i < 16; ++i) t += acc[m][n][i];
          atomicAdd(C32 + (int64_t)(job.I * TM + wm * 128 + r) * Npad + job.J * TN + wn * 128 + h, (int)t);
          return;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int col = job.J * TN + wn * 128 + n * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = job.I * TM + wm * 128 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
              atomicAdd(C32 + (int64_t)row * Npad + col, (int)acc[m][n][i]);
            }
          }
      });
}

__device__ __forceinline__ float byte_to_f32(int word, int j) {
  return (float)(int)(int8_t)((word >> (8 * j)) & 0xff);
}

__global__ __launch_bounds__(NTHREADS, 2) void kinship_f32_kernel(const int8_t* __restrict__ Xt, int64_t Mk,
                                                                  int32_t Npad, const KinJob* __restrict__ jobs,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const KinJob job = jobs[xcd_job_index(blockIdx.x)];
  if (job.ks1 <= job.ks0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
  const int8_t* P = Xt + (int64_t)job.I * TM * Mk;
  const int8_t* Q = Xt + (int64_t)job.J * TN * Mk;
  v16f acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  const StageOp sp = make_stage_op(P, Mk, wave, lane);
  const StageOp sq = make_stage_op(Q, Mk, wave, lane);
  stage_tile(sp, job.ks0 * BK, lds, wave);
  stage_tile(sq, job.ks0 * BK, lds + TILE_BYTES, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int ks = job.ks0; ks < job.ks1; ++ks) {
    if (ks + 1 < job.ks1) {
      char* nb = lds + (cur ^ 1) * BUF_BYTES;
      stage_tile(sp, (ks + 1) * BK, nb, wave);
      stage_tile(sq, (ks + 1) * BK, nb + TILE_BYTES, wave);
    }
    const char* pt = lds + cur * BUF_BYTES;
    const char* qt = pt + TILE_BYTES;
#pragma unroll 1
    for (int kk = 0; kk < 4; ++kk) {
      // this lane's 16 contraction indices of the sub-step: k = ks*128 + kk*32 + h*16 + j
      const int64_t kbase = (int64_t)ks * BK + kk * 32 + h * 16;
      v4i ab[4], bb[2];
#pragma unroll
      for (int m = 0; m < 4; ++m) ab[m] = lds_frag(pt, wm * 128 + m * 32 + r, 2 * kk + h);
#pragma unroll
      for (int n = 0; n < 2; ++n) bb[n] = lds_frag(qt, wn * 64 + n * 32 + r, 2 * kk + h);
      float sc[16], sh[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 s4 = *(const float4*)(scale + kbase + q4 * 4);
        const float4 t4 = *(const float4*)(shift + kbase + q4 * 4);
        sc[q4 * 4 + 0] = s4.x; sc[q4 * 4 + 1] = s4.y; sc[q4 * 4 + 2] = s4.z; sc[q4 * 4 + 3] = s4.w;
        sh[q4 * 4 + 0] = t4.x; sh[q4 * 4 + 1] = t4.y; sh[q4 * 4 + 2] = t4.z; sh[q4 * 4 + 3] = t4.w;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float af[4], bf[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = fmaf(byte_to_f32(ab[m][j >> 2], j & 3), sc[j], sh[j]);
#pragma unroll
        for (int n = 0; n < 2; ++n) bf[n] = fmaf(byte_to_f32(bb[n][j >> 2], j & 3), sc[j], sh[j]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[n], acc[m][n], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  float* slab = slabs + (int64_t)job.slab * Npad * Npad;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = job.J * TN + wn * 64 + n * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = job.I * TM + wm * 128 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        slab[(int64_t)row * Npad + col] = acc[m][n][i];
      }
    }
}

// C[i][j] (i,j < N) = sum over slabs in fixed order of the upper-triangular tile entry, in fp64.
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int ksplit, int32_t Npad, int32_t N,
                                    double* __restrict__ C, int accumulate) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)N * N) return;
  const int i = (int)(gid / N), j = (int)(gid % N);
  // tile (I,J) with I<=J was computed; inside a diagonal tile both halves are present
  int a = i, b = j;
  if ((i / TM) > (j / TN)) { a = j; b = i; }
  double s = 0.0;
  for (int k = 0; k < ksplit; ++k) s += (double)slabs[(int64_t)k * Npad * Npad + (int64_t)a * Npad + b];
  C[gid] = accumulate ? C[gid] + s : s;
}

__global__ void mirror_i32_kernel(const int* __restrict__ C32, int32_t Npad, int32_t N, int64_t* __restrict__ C) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)N * N) return;
  const int i = (int)(gid / N), j = (int)(gid % N);
  int a = i, b = j;
  if ((i / TM) > (j / TN)) { a = j; b = i; }
  C[gid] = (int64_t)C32[(int64_t)a * Npad + b];
}

// IBS counts from the Gram matrix of the RAW genotypes: sum_m (2 s_i - 1)(2 s_j - 1) = 4 (S'S)_ij - 2 (r_i + r_j) + M,
// r = column sums of S, M = SNPs (exact integers).
__global__ void mirror_ibs_kernel(const int* __restrict__ C32, int32_t Npad, int32_t N, const long long* __restrict__ r,
                                  long long Mtot, int64_t* __restrict__ C) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)N * N) return;
  const int i = (int)(gid / N), j = (int)(gid % N);
  int a = i, b = j;
  if ((i / TM) > (j / TN)) { a = j; b = i; }
  C[gid] = 4 * (int64_t)C32[(int64_t)a * Npad + b] - 2 * (r[i] + r[j]) + Mtot;
}

void launch_mirror_ibs(mmg_ctx* ctx, const int* C32, int32_t Npad, int32_t N, const long long* r, long long Mtot,
                       int64_t* C) {
  const int64_t total = (int64_t)N * N;
  hipLaunchKernelGGL(mirror_ibs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, C32, Npad, N, r,
                     Mtot, C);
}

void launch_reduce_slabs(mmg_ctx* ctx, const float* slabs, int ksplit, int32_t Npad, int32_t N, double* C,
                         int accumulate) {
  const int64_t total = (int64_t)N * N;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     slabs, ksplit, Npad, N, C, accumulate);
}
// K[e] = counts[e] / (2 M) + 0.5 (kinship.py:44-46 on the exact counts): the same two IEEE operations the host mirror does
__global__ void ibs_counts_to_f64_kernel(const int64_t* __restrict__ C, int64_t n, double two_m, double* __restrict__ K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) K[e] = (double)C[e] / two_m + 0.5;
}
// K_ij = 1 - (r_i + r_j - 2 c12_ij) / (2 M) off the diagonal (= (M - 1/2 sum |a - b|) / M, evaluated as the host mirror does:
// M - 0.5 * absdiff, then / M), 1 on it; c12 = c1 + c2, r its diagonal
__global__ void ibs_diploid_combine_kernel(const int64_t* __restrict__ c1, const int64_t* __restrict__ c2, int64_t N, double M,
                                           double* __restrict__ K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * N) return;
  const int64_t i = e / N, j = e - i * N;
  if (i == j) { K[e] = 1.0; return; }
  const double c12 = (double)(c1[e] + (c2 ? c2[e] : 0));
  const double ri = (double)(c1[i * N + i] + (c2 ? c2[i * N + i] : 0)), rj = (double)(c1[j * N + j] + (c2 ? c2[j * N + j] : 0));
  const double absdiff = ri + rj - 2.0 * c12;
  K[e] = (M - 0.5 * absdiff) / M;
}
void launch_ibs_diploid_combine(mmg_ctx* ctx, const int64_t* c1, const int64_t* c2, int64_t N, double M, double* K) {
  hipLaunchKernelGGL(ibs_diploid_combine_kernel, dim3((unsigned)((N * N + 255) / 256)), dim3(256), 0, ctx->stream, c1, c2, N, M, K);
}

void launch_ibs_counts_to_f64(mmg_ctx* ctx, const int64_t* C, int64_t n, double two_m, double* K) {
  hipLaunchKernelGGL(ibs_counts_to_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, C, n, two_m, K);
}

void launch_mirror_i32_to_i64(mmg_ctx* ctx, const int* C32, int32_t Npad, int32_t N, int64_t* C) {
  const int64_t total = (int64_t)N * N;
  hipLaunchKernelGGL(mirror_i32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, C32,
                     Npad, N, C);
}

// Upper-triangular tiles in 4 x 8 patches (neighbouring jobs share operand panels), K-split
// chosen so that the job count fills whole cohorts of 256 workgroups.
static int choose_ksplit(int ntiles, int nk, int min_steps, int max_split) {
  int best = 1;
  double best_score = -1.0;
  for (int ks = 1; ks <= max_split && ks <= nk; ++ks) {
    if (ks > 1 && nk / ks < min_steps) break;
    const int64_t jobs = (int64_t)ntiles * ks;
    if (jobs > 4096 && ks > 1) break;
    const double score = (double)jobs / (double)round_up(jobs, 256);
    if (score > best_score + 1e-9) { best_score = score; best = ks; }
  }
  return best;
}

static std::vector<KinJob> build_jobs(int nT, int nk, int ksplit) {
  // patch shape of the tile order (MMG_KIN_PATCH=IxJ, A/B runs): consecutive jobs share operand panels in an XCD's L2
  static const std::pair<int, int> patch = [] {
    int pi = 4, pj = 8;
    if (const char* e = std::getenv("MMG_KIN_PATCH")) { int a = 0, b = 0; if (std::sscanf(e, "%dx%d", &a, &b) == 2 && a > 0 && b > 0) { pi = a; pj = b; } }
    return std::make_pair(pi, pj);
  }();
  const int PI = patch.first, PJ = patch.second;
  std::vector<std::pair<int, int>> tiles;
  for (int Ib = 0; Ib < nT; Ib += PI)
    for (int Jb = 0; Jb < nT; Jb += PJ)
      for (int I = Ib; I < std::min(Ib + PI, nT); ++I)
        for (int J = Jb; J < std::min(Jb + PJ, nT); ++J)
          if (I <= J) tiles.push_back({I, J});
  std::vector<KinJob> jobs;
  for (int s = 0; s < ksplit; ++s) {
    const int k0 = (int)((int64_t)nk * s / ksplit), k1 = (int)((int64_t)nk * (s + 1) / ksplit);
    for (auto& t : tiles) jobs.push_back(KinJob{t.first, t.second, k0, k1, s, 0, 0, 0});
  }
  while (jobs.size() % 256) jobs.push_back(KinJob{0, 0, 0, 0, 0, 0, 0, 0});
  return jobs;
}

int kinship_pick_ksplit(int32_t Npad, int64_t Mk, bool f32) {
  const int nT = Npad / TM;
  const int nk = (int)(Mk / BK);
  return choose_ksplit(nT * (nT + 1) / 2, nk, f32 ? 2 : 8, f32 ? 16 : 64);
}

// dC (fp64 [N x N]) (+)= step * sum_d base^d C32[d] (upper tiles mirrored) + c1[i] + c1[j] + c0
// One block per 64 x 64 sub-tile of a valid (upper) 256 x 256 tile of C32: the planes are read once, along rows; the
// value goes to dC[i][j] and -- off the tile diagonal -- through LDS to dC[j][i], both along rows.  (The first version
// took one thread per element of dC and read the mirrored half of every plane down its columns: 121 ms of a 388 ms
// GRM chunk at N = 50,000.)
__global__ __launch_bounds__(256) void grm_combine_kernel(const int* __restrict__ C32, int D, int32_t Npad, int32_t N, double step,
                                                          double base, const double* __restrict__ c1, double c0,
                                                          double* __restrict__ C, int accumulate) {
  const int si = blockIdx.y, sj = blockIdx.x;
  if ((si * 64) / TM > (sj * 64) / TN) return;                 // not a tile the GEMM wrote
  const bool mirror = (si * 64) / TM < (sj * 64) / TN;        // diagonal tiles hold both halves themselves
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = sj * 64 + tx;
  const double c1j = j < N ? c1[j] : 0.0;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int i = si * 64 + ty + 4 * r;
    double sum = 0.0, pw = 1.0;
    const int* src = C32 + (int64_t)i * Npad + j;
    for (int d = 0; d < D; ++d) {
      sum = fma((double)src[(int64_t)d * Npad * Npad], pw, sum);   // exact: integers below 2^53
      pw *= base;
    }
    const double v = (i < N && j < N) ? fma(step, sum, c1[i] + c1j + c0) : 0.0;
    tile[ty + 4 * r][tx] = v;
    if (i < N && j < N) {
      const int64_t at = (int64_t)i * N + j;
      C[at] = accumulate ? C[at] + v : v;
    }
  }
  if (!mirror) return;
  __syncthreads();
  const int i2 = si * 64 + tx;                                 // column of the mirrored element
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int j2 = sj * 64 + ty + 4 * r;                       // its row
    if (i2 < N && j2 < N) {
      const int64_t at = (int64_t)j2 * N + i2;
      const double v = tile[tx][ty + 4 * r];
      C[at] = accumulate ? C[at] + v : v;
    }
  }
}

void launch_grm_combine(mmg_ctx* ctx, const int* C32, int D, int32_t Npad, int32_t N, double step, double base,
                        const double* c1, double c0, double* C, int accumulate) {
  const unsigned nS = (unsigned)(Npad / 64);
  hipLaunchKernelGGL(grm_combine_kernel, dim3(nS, nS), dim3(256), 0, ctx->stream, C32, D, Npad, N, step, base, c1, c0, C,
                     accumulate);
}

constexpr int KIN_PF_DEFAULT = 0;     // L2 prefetch distance of the transposed-read kinship kernel (see run_kinship_i8_tr)

// C32 (upper tiles) += Sp' Sq over rows [0, nk * 128) of two SNP-major images with row stride ld (Sp == Sq: the store)
// Device copy of a launch's job list in the context's cached buffer (no hipMalloc / hipFree per launch: those serialise
// the device against the upload stream, and an early return between them leaked the list -- advisor r3).  The copy is
// asynchronous on the library stream; the callers synchronise before `jobs` goes out of scope.
static int job_buffer(mmg_ctx* ctx, const std::vector<KinJob>& jobs, KinJob** out) {
  const size_t bytes = jobs.size() * sizeof(KinJob);
  if (ctx->jobs_cap < bytes) {
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));        // an earlier launch may still read the old buffer
    (void)hipFree(ctx->jobs);
    ctx->jobs = nullptr; ctx->jobs_cap = 0;
    MMG_HIP(ctx, hipMalloc(&ctx->jobs, bytes + bytes / 2));
    ctx->jobs_cap = bytes + bytes / 2;
  }
  MMG_HIP(ctx, hipMemcpyAsync(ctx->jobs, jobs.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
  *out = (KinJob*)ctx->jobs;
  return MMG_OK;
}

int run_kinship_i8_tr(mmg_ctx* ctx, const int8_t* Sp, const int8_t* Sq, int64_t ld, int32_t Npad, int64_t nk, int* C32) {
  const int nT = Npad / TM;
  const int ksplit = kinship_pick_ksplit(Npad, nk * BK, false);
  std::vector<KinJob> jobs = build_jobs(nT, (int)nk, ksplit);
  KinJob* djobs = nullptr;
  { int rcj = job_buffer(ctx, jobs, &djobs); if (rcj) return rcj; }
  // MMG_KIN_N3 = 8 | 16: DMA pieces issued right behind the barrier; MMG_KIN_PF = 0 | 1 | 2: L2 prefetch distance
  // (gemm_i8_w4tr.h) -- A/B timing
  static const int n3 = [] { const char* e = std::getenv("MMG_KIN_N3"); return e && std::atoi(e) == 16 ? 16 : 8; }();
  static const int pf = [] { const char* e = std::getenv("MMG_KIN_PF"); const int v = e ? std::atoi(e) : KIN_PF_DEFAULT; return v < 0 || v > 2 ? KIN_PF_DEFAULT : v; }();
  const size_t lds_bytes = LDS_BYTES + (pf ? W4TR_PF_LDS : 0);
  {
#define MMG_LAUNCH_KTR(N3_, PF_)                                                                                       \
  do {                                                                                                                 \
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_i8_tr_kernel<N3_, PF_>,                                      \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                     \
    EvScope ev(ctx, EV_KIN);                                                                                           \
    hipLaunchKernelGGL((kinship_i8_tr_kernel<N3_, PF_>), dim3((unsigned)jobs.size()), dim3(W4_THREADS), lds_bytes,    \
                       ctx->stream, Sp, Sq, ld, Npad, djobs, C32);                                                     \
  } while (0)
    if (n3 == 16) { if (pf == 2) MMG_LAUNCH_KTR(16, 2); else if (pf == 1) MMG_LAUNCH_KTR(16, 1); else MMG_LAUNCH_KTR(16, 0); }
    else { if (pf == 2) MMG_LAUNCH_KTR(8, 2); else if (pf == 1) MMG_LAUNCH_KTR(8, 1); else MMG_LAUNCH_KTR(8, 0); }
#undef MMG_LAUNCH_KTR
  }
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

// C32 (upper tiles) += X'X over rows [0, nk4 * 256) of the FP4 image X4 (row stride Npad / 2), enqueued on ctx->stream
// WITHOUT synchronising (the caller overlaps the image pass of the next SNP chunk on a second stream); the job list lives
// in `sc` until the caller's scope ends.  MMG_E_STATE: the contraction range of a job could exceed the exact range of the
// fp32 accumulators (caller takes the int8 kernel).
int run_kinship_f4_tr(mmg_ctx* ctx, Scratch& sc, const uint8_t* X4, int32_t Npad, int64_t nk4, int* C32) {
  const int nT = Npad / TM;
  int ksplit = choose_ksplit(nT * (nT + 1) / 2, (int)nk4, 4, 64);
  if (const char* e = std::getenv("MMG_KIN_KSPLIT")) ksplit = std::max(1, std::min((int)nk4, std::atoi(e)));   // A/B runs (tools/kin_sweep.sh)
  if ((nk4 + ksplit - 1) / ksplit * FmtF4::KROWS >= (int64_t(1) << 24)) return MMG_E_STATE;
  std::vector<KinJob> jobs = build_jobs(nT, (int)nk4, ksplit);
  KinJob* djobs = nullptr;
  MMG_HIP(ctx, sc.alloc(&djobs, jobs.size() * sizeof(KinJob)));
  MMG_HIP(ctx, hipMemcpy(djobs, jobs.data(), jobs.size() * sizeof(KinJob), hipMemcpyHostToDevice));   // `jobs` dies with this call
#define MMG_LAUNCH_F4(...)                                                                                                                      \
  do {                                                                                                                                          \
    const size_t lb = LDS_BYTES + W4TR_PF_LDS;                                                                                                  \
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_f4_tr_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb)); \
    hipLaunchKernelGGL((kinship_f4_tr_kernel<__VA_ARGS__>), dim3((unsigned)jobs.size()), dim3(W4_THREADS), lb, ctx->stream,                   \
                       (const int8_t*)X4, (int64_t)(Npad / 2), Npad, djobs, C32);                                                              \
  } while (0)
  int abl = 0;
#ifdef MMG_EXPERIMENTS
  if (const char* e = std::getenv("MMG_F4_ABL")) abl = std::atoi(e);                    // 1-4: timing ablations; 12 / 16: N3; 101 / 102: L2 prefetch distance
  if (abl == 1) MMG_LAUNCH_F4(1); else if (abl == 2) MMG_LAUNCH_F4(2); else if (abl == 3) MMG_LAUNCH_F4(3); else if (abl == 4) MMG_LAUNCH_F4(4);
  else if (abl == 12) MMG_LAUNCH_F4(0, 12); else if (abl == 16) MMG_LAUNCH_F4(0, 16);
  else if (abl == 101) MMG_LAUNCH_F4(0, 8, 1); else if (abl == 102) MMG_LAUNCH_F4(0, 8, 2); else if (abl == 113) MMG_LAUNCH_F4(0, 12, 1); else
#endif
  MMG_LAUNCH_F4(0);
  (void)abl;
#undef MMG_LAUNCH_F4
  MMG_HIP(ctx, hipGetLastError());
  return MMG_OK;
}

// C32[d] (tiles of the upper 256-tile triangle, d = 0..3) += (diag(dig_d) S)' S over rows [0, nk * 128) of the store;
// dig: device [4][dig_stride] digit bytes.  Binary stores only (the scaling is a byte mask).
int run_kinship_grm4(mmg_ctx* ctx, const int8_t* S, int64_t ld, int32_t Npad, int64_t nk, const int8_t* dig, int64_t dig_stride,
                     int* C32) {
  const int nT = Npad / G4_T;
  std::vector<std::pair<int, int>> tiles;                 // every 128-tile of the 256-tiles (I <= J) the combine pass reads
  int PI = 8, PJ = 16;                                    // patch of the tile order: consecutive jobs share operand panels in L2
  if (const char* e = std::getenv("MMG_GRM4_PATCH")) { int a = 0, b = 0; if (std::sscanf(e, "%dx%d", &a, &b) == 2 && a > 0 && b > 0) { PI = a; PJ = b; } }
  for (int Ib = 0; Ib < nT; Ib += PI)
    for (int Jb = 0; Jb < nT; Jb += PJ)
      for (int I = Ib; I < std::min(Ib + PI, nT); ++I)
        for (int J = Jb; J < std::min(Jb + PJ, nT); ++J)
          if ((I >> 1) <= (J >> 1)) tiles.push_back({I, J});
  const int ksplit = choose_ksplit((int)tiles.size(), (int)nk, 8, 64);
  std::vector<KinJob> jobs;
  for (int s = 0; s < ksplit; ++s) {
    const int k0 = (int)((int64_t)nk * s / ksplit), k1 = (int)((int64_t)nk * (s + 1) / ksplit);
    for (auto& t : tiles) jobs.push_back(KinJob{t.first, t.second, k0, k1, s, 0, 0, 0});
  }
  while (jobs.size() % 256) jobs.push_back(KinJob{0, 0, 0, 0, 0, 0, 0, 0});
  KinJob* djobs = nullptr;
  { int rcj = job_buffer(ctx, jobs, &djobs); if (rcj) return rcj; }
  int abl = 0;                                            // MMG_GRM4_ABL=1..7: timing ablations (wrong results; tools/grm4_abl.py),
#ifdef MMG_EXPERIMENTS                                    // 8..11: stamps -- in a `make EXPERIMENTS=1` library only
  if (const char* e = std::getenv("MMG_GRM4_ABL")) abl = std::atoi(e);
#endif
#define MMG_LAUNCH_G4(A)                                                                                                \
  do {                                                                                                                 \
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_grm4_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS)); \
    EvScope ev(ctx, EV_KIN);                                                                                           \
    hipLaunchKernelGGL(kinship_grm4_kernel<A>, dim3((unsigned)jobs.size()), dim3(W4_THREADS), G4_LDS, ctx->stream, S, ld, \
                       Npad, dig, (int)dig_stride, djobs, C32);                                                        \
  } while (0)
  static const std::string layout = [] { const char* e = std::getenv("MMG_GRM4_LAYOUT"); return std::string(e ? e : ""); }();
  // default (round 5): kinship_grm4j_kernel, every slice scales its own operands; MMG_GRM4_LAYOUT=quad: the kernel of rounds
  // 3-4 (scaling written one slice ahead), =strips: round 4's row strips -- all three bit-identical (tools/grm4_layouts.py)
  const bool quad = layout != "strips";
  if (layout != "quad" && layout != "strips" && abl == 0) {
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_grm4j_kernel<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
    EvScope ev(ctx, EV_KIN);
    hipLaunchKernelGGL(kinship_grm4j_kernel<-1>, dim3((unsigned)jobs.size()), dim3(W4_THREADS), G4_LDS, ctx->stream, S, ld, Npad, dig,
                       (int)dig_stride, djobs, C32, (unsigned long long*)nullptr);
  } else
  if (!quad && abl == 0) {                                // MMG_GRM4_LAYOUT=strips: four row strips (round 4; not faster)
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_grm4r_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G4R_LDS));
    EvScope ev(ctx, EV_KIN);
    hipLaunchKernelGGL(kinship_grm4r_kernel<0>, dim3((unsigned)jobs.size()), dim3(W4_THREADS), G4R_LDS, ctx->stream, S, ld, Npad,
                       dig, (int)dig_stride, djobs, C32);
  } else
#ifdef MMG_EXPERIMENTS
  if (abl == 7) {                                         // strips without the digit reads (timing only)
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_grm4r_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, G4R_LDS));
    EvScope ev(ctx, EV_KIN);
    hipLaunchKernelGGL(kinship_grm4r_kernel<7>, dim3((unsigned)jobs.size()), dim3(W4_THREADS), G4R_LDS, ctx->stream, S, ld, Npad,
                       dig, (int)dig_stride, djobs, C32);
  } else
  if (abl == 1) MMG_LAUNCH_G4(1); else if (abl == 2) MMG_LAUNCH_G4(2); else if (abl == 3) MMG_LAUNCH_G4(3); else if (abl == 4) MMG_LAUNCH_G4(4); else if (abl == 5) MMG_LAUNCH_G4(5); else if (abl == 6) MMG_LAUNCH_G4(6);
  else if (abl >= 8 && abl <= 15) {
    // right results + s_memtime stamps: variant 8 + k stamps every K step behind position k (0: slice 2, 1: slice 3, 2: the
    // vmcnt / lgkmcnt wait + barrier, 3: slice 0 with the P pieces) and at the end of the step (slice 1 with the Q pieces and
    // the digits).  One position per variant: with all five in one kernel the allocator spills 134 VGPRs (the loop body sits at
    // 254 + 256 registers).  tools/grm4_stamps.py runs the four and differences them.
    static unsigned long long* dst = nullptr;
    static size_t cap = 0;
    const size_t n = jobs.size() * 4 * 8;
    if (cap < n) { (void)hipFree(dst); dst = nullptr; cap = 0; MMG_HIP(ctx, hipMalloc(&dst, n * sizeof(unsigned long long))); cap = n; }
    MMG_HIP(ctx, hipMemsetAsync(dst, 0, n * sizeof(unsigned long long), ctx->stream));
#define MMG_LAUNCH_G4S(A)                                                                                              \
  do {                                                                                                                 \
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_grm4_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS)); \
    EvScope ev(ctx, EV_KIN);                                                                                           \
    hipLaunchKernelGGL(kinship_grm4_kernel<A>, dim3((unsigned)jobs.size()), dim3(W4_THREADS), G4_LDS, ctx->stream, S, ld, \
                       Npad, dig, (int)dig_stride, djobs, C32, dst);                                                   \
  } while (0)
#define MMG_LAUNCH_G4JS(A)                                                                                             \
  do {                                                                                                                 \
    MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_grm4j_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS)); \
    EvScope ev(ctx, EV_KIN);                                                                                           \
    hipLaunchKernelGGL(kinship_grm4j_kernel<A>, dim3((unsigned)jobs.size()), dim3(W4_THREADS), G4_LDS, ctx->stream, S, ld, \
                       Npad, dig, (int)dig_stride, djobs, C32, dst);                                                   \
  } while (0)
    // 12..15: the same four positions of kinship_grm4j_kernel (slices 0, 1, 2 + barrier, 3 in ITS order)
    if (abl == 8) MMG_LAUNCH_G4S(8); else if (abl == 9) MMG_LAUNCH_G4S(9); else if (abl == 10) MMG_LAUNCH_G4S(10); else if (abl == 11) MMG_LAUNCH_G4S(11);
    else if (abl == 12) MMG_LAUNCH_G4JS(0); else if (abl == 13) MMG_LAUNCH_G4JS(1); else if (abl == 14) MMG_LAUNCH_G4JS(2); else MMG_LAUNCH_G4JS(3);
#undef MMG_LAUNCH_G4JS
#undef MMG_LAUNCH_G4S
    std::vector<unsigned long long> h(n);
    MMG_HIP(ctx, hipMemcpyAsync(h.data(), dst, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double head = 0, tail = 0, steps = 0, total = 0, hw[4] = {0, 0, 0, 0}, sw[4] = {0, 0, 0, 0};
    long cnt = 0;
    for (size_t w = 0; w < jobs.size() * 4; ++w) {
      if (h[w * 8 + 5] == 0) continue;
      head += (double)h[w * 8 + ((abl - 8) & 3)]; tail += (double)h[w * 8 + 4]; steps += (double)h[w * 8 + 5]; total += (double)h[w * 8 + 6];
      hw[w & 3] += (double)h[w * 8 + ((abl - 8) & 3)]; sw[w & 3] += (double)h[w * 8 + 5];
      ++cnt;
    }
    if (cnt)
      fprintf(stderr, "[grm4 stamps] position %d  waves %ld  K steps / wave %.0f  cycles per K step: start -> position %.1f  position -> end %.1f  "
                      "whole loop %.1f   (start -> position by wave: %.1f %.1f %.1f %.1f)\n", (abl - 8) & 3, cnt, steps / cnt, head / steps,
              tail / steps, total / steps, hw[0] / sw[0], hw[1] / sw[1], hw[2] / sw[2], hw[3] / sw[3]);
  } else
#endif
  MMG_LAUNCH_G4(0);
  (void)abl;
#undef MMG_LAUNCH_G4
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int run_kinship_i8(mmg_ctx* ctx, const int8_t* Xt, int32_t Npad, int64_t Mk, int* C32) {
  return run_kinship_i8_pq(ctx, Xt, Xt, Npad, Mk, C32);
}

int run_kinship_i8_pq(mmg_ctx* ctx, const int8_t* Xp, const int8_t* Xq, int32_t Npad, int64_t Mk, int* C32) {
  const int nT = Npad / TM, nk = (int)(Mk / BK);
  const int ksplit = kinship_pick_ksplit(Npad, Mk, false);
  std::vector<KinJob> jobs = build_jobs(nT, nk, ksplit);
  KinJob* djobs = nullptr;
  { int rcj = job_buffer(ctx, jobs, &djobs); if (rcj) return rcj; }
  // MMG_KIN_KERNEL=w8: the first-generation 8-wave kernel (A/B runs; same bits)
  static const bool w8 = [] { const char* e = std::getenv("MMG_KIN_KERNEL"); return e && std::string(e) == "w8"; }();
  const void* fn = w8 ? (const void*)kinship_i8_kernel : (const void*)kinship_i8_w4_kernel;
  MMG_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  {
    EvScope ev(ctx, EV_KIN);
    if (w8)
      hipLaunchKernelGGL(kinship_i8_kernel, dim3((unsigned)jobs.size()), dim3(NTHREADS), LDS_BYTES, ctx->stream, Xp, Xq,
                         Mk, Npad, djobs, C32);
    else
      hipLaunchKernelGGL(kinship_i8_w4_kernel, dim3((unsigned)jobs.size()), dim3(W4_THREADS), LDS_BYTES, ctx->stream, Xp,
                         Xq, Mk, Npad, djobs, C32);
  }
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int run_kinship_f32(mmg_ctx* ctx, const int8_t* Xt, int32_t Npad, int64_t Mk, const float* scale,
                    const float* shift, float* slabs, int ksplit) {
  const int nT = Npad / TM, nk = (int)(Mk / BK);
  std::vector<KinJob> jobs = build_jobs(nT, nk, ksplit);
  KinJob* djobs = nullptr;
  { int rcj = job_buffer(ctx, jobs, &djobs); if (rcj) return rcj; }
  MMG_HIP(ctx, hipFuncSetAttribute((const void*)kinship_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  {
    EvScope ev(ctx, EV_KIN);
    hipLaunchKernelGGL(kinship_f32_kernel, dim3((unsigned)jobs.size()), dim3(NTHREADS), LDS_BYTES, ctx->stream, Xt,
                       Mk, Npad, djobs, scale, shift, slabs);
  }
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

}  // namespace mmg
