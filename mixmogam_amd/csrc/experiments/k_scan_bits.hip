// k_scan_bits.hip -- the EMMAX quadratic-form GEMM for BINARY genotypes (0/1, the reference's
// 'binary' data format) with the SNP operand staged bit-packed.
//
// The 256x256 int8 tile loop of gemm_i8_core.h moves 64 KiB through the CU's vector-memory path per
// 32 MFMA per wave, which is what it waits on.  The genotype operand carries 1 bit
// of information per byte, so this kernel stages it as bits:
//   * Q stage  = 256 SNP rows x 16 bytes (128 individuals) = 4 KiB instead of 32 KiB;
//   * a lane rebuilds its MFMA B fragment (16 int8) from 16 bits with 4 x {bfe, mul, and}:
//     (nibble * 0x00204081) & 0x01010101 puts bit e of the nibble into byte e;
//   * stage = 36 KiB, so FOUR stages fit in LDS: three K steps of lookahead with counted vmcnt at
//     the same one-barrier-per-32-MFMA cadence.
// Arithmetic, accumulation and the integer atomics are those of scan_quad_kernel: results are
// bit-identical.
#include <cstdlib>
#include <string>
#include "gemm_i8_core.h"
#include "mmg_internal.h"

namespace mmg {

constexpr int QB_BYTES = 256 * 16;                 // packed Q stage
constexpr int BSLOT = TILE_BYTES + QB_BYTES;       // 36 KiB
constexpr int BSLOTS = 4;
constexpr int BITS_LDS = BSLOTS * BSLOT;           // 144 KiB

// ---- bit-pack the store: bits[m][i/8] bit (i%8) = (s[m][i] != 0); flags any value outside {0,1}
__global__ void pack_bits_kernel(const int8_t* __restrict__ S, int64_t Mpad, int32_t Npad, uint8_t* __restrict__ bits,
                                 int* __restrict__ nonbinary) {
  const int chunks = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= Mpad * chunks) return;
  const int64_t m = gid / chunks;
  const int c = (int)(gid % chunks);
  const uint4 v = *(const uint4*)(S + m * (int64_t)Npad + c * 16);
  const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
  uint32_t out = 0, bad = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t b = (wds[j >> 2] >> (8 * (j & 3))) & 0xff;
    out |= (b & 1u) << j;
    bad |= b & 0xfeu;
  }
  *(uint16_t*)(bits + m * (int64_t)(Npad >> 3) + c * 2) = (uint16_t)out;
  if (bad) atomicOr(nonbinary, 1);
}

int ensure_bits(mmg_ctx* ctx, mmg_geno* g) {
  if (g->bits_valid) return MMG_OK;
  if (!g->bits) MMG_HIP(ctx, hipMalloc(&g->bits, (size_t)g->Mpad * (g->Npad >> 3)));
  int* dflag = nullptr;
  MMG_HIP(ctx, hipMalloc(&dflag, sizeof(int)));
  MMG_HIP(ctx, hipMemsetAsync(dflag, 0, sizeof(int), ctx->stream));
  const int64_t total = g->Mpad * (int64_t)(g->Npad >> 4);
  hipLaunchKernelGGL(pack_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g->d,
                     g->Mpad, g->Npad, g->bits, dflag);
  MMG_HIP(ctx, hipGetLastError());
  int flag = 0;
  MMG_HIP(ctx, hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  hipFree(dflag);
  g->binary = (flag == 0);
  g->bits_valid = true;
  return MMG_OK;
}

__device__ __forceinline__ v4i expand16(uint32_t x16) {
  v4i o;
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (int)((((x16 >> (4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u);
  return o;
}

// Only waves 0-3 issue DMA (loader waves, as in gemm_i8_core.h MODE 5): 8 P pieces (own rows and the
// SIMD partner's) + 1 Q piece = 9 loads per stage; waves 4-7 have nothing in flight.
__device__ __forceinline__ void wait_stages(int keep) {
  if (keep >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if (keep == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(NTHREADS, 2) void scan_quad_bits_kernel(
    const uint8_t* __restrict__ Sb, int64_t ldSb, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
    int64_t digit_stride, const int* __restrict__ job_off, const int2* __restrict__ jobs, int AS,
    unsigned long long* __restrict__ q) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = blockIdx.x;
  const int x = b & 7, i = b >> 3;
  const int cohort = i >> 5, within = i & 31;
  const int a = within % AS, grp = within / AS;
  const int sb = (cohort * 8 + x) * AS + a;
  if (sb >= nSb) return;
  const int j0 = job_off[grp], j1 = job_off[grp + 1];
  if (j1 <= j0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
  const bool qloader = wave < 4;                     // waves 0-3 each move one 64-row piece of the Q stage
  // Q staging: piece `wave` = rows wave*64 .. +63, 16 bytes per row; LDS image [row][16 B] linear
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Sb + (int64_t)sb * TN * ldSb), 0, 0x7fffffff, 0x00020000);
  const int vq = (wave * 64 + lane) * (int)ldSb;
  unsigned long long qacc[2] = {0ull, 0ull};

  for (int jj = j0; jj < j1; ++jj) {
    const int2 jb = jobs[jj];
    const int d = jb.x, J = jb.y;
    const int nks = 2 * (J + 1);
    const StageOp sp = make_stage_op(Bq + (int64_t)d * digit_stride + (int64_t)J * TM * ldB, ldB, wave, lane);
    v16i acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][n][e] = 0;

    auto issue = [&](int ks) {
      if (!qloader) return;
      char* slot = lds + (ks & (BSLOTS - 1)) * BSLOT;
      stage_tile_pair(sp, ks * BK, slot, wave);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (MMG_AS3 void*)(slot + TILE_BYTES + wave * 1024), 16, vq,
                                               ks * 16, 0, 0);
    };
#pragma unroll 1
    for (int ks = 0; ks < 3 && ks < nks; ++ks) issue(ks);

#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
      const int keep = min(2, nks - 1 - ks);         // younger stages allowed to stay in flight
      if (qloader) wait_stages(keep);
      __builtin_amdgcn_s_barrier();                   // stage ks landed for every wave; slot (ks-1)&3 is free
      asm volatile("" ::: "memory");
      if (ks + 3 < nks) issue(ks + 3);   // (spreading the pieces between the MFMA groups measured 4 % slower)
      const char* pt = lds + (ks & (BSLOTS - 1)) * BSLOT;
      const char* qt = pt + TILE_BYTES;
      v4i rowbits[2];
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) rowbits[nn] = *(const v4i*)(qt + (wn * 64 + nn * 32 + r) * 16);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        v4i af[4], bf[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = lds_frag(pt, wm * 128 + m * 32 + r, 2 * kk + h);
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) bf[nn] = expand16(((uint32_t)rowbits[nn][kk] >> (16 * h)) & 0xFFFFu);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
            acc[m][nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[m], bf[nn], acc[m][nn], 0, 0, 0);
      }
    }
    // Epilogue operands: bits of columns 256J + wm*128 .. +127 = the Q stage of K step 2J + wm,
    // still resident (nothing was issued into the slots of the last two steps).
    {
      const char* qt = lds + ((nks - 2 + wm) & (BSLOTS - 1)) * BSLOT + TILE_BYTES;
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        const v4i rb = *(const v4i*)(qt + (wn * 64 + nn * 32 + r) * 16);
        long long part = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t nib = ((uint32_t)rb[m] >> (8 * g4 + 4 * h)) & 0xFu;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              part += ((nib >> e) & 1u) ? (long long)acc[m][nn][g4 * 4 + e] : 0ll;
          }
        qacc[nn] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
      }
    }
    __builtin_amdgcn_s_barrier();                     // the next job's prologue refills slots 0..2
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    unsigned long long v = qacc[nn];
    v += __shfl_xor(v, 32);
    if (h == 0) atomicAdd(q + (int64_t)sb * TN + wn * 64 + nn * 32 + r, v);
  }
}

void launch_scan_quad_bits(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, unsigned long long* q) {
  const int nSb = (int)(g->Mpad / TN);
  const int per = 8 * md.AS;
  const int ncoh = (nSb + per - 1) / per;
  hipFuncSetAttribute((const void*)scan_quad_bits_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BITS_LDS);
  hipLaunchKernelGGL(scan_quad_bits_kernel, dim3((unsigned)(ncoh * 256)), dim3(NTHREADS), BITS_LDS, ctx->stream,
                     g->bits, (int64_t)(g->Npad >> 3), nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,
                     md.job_off, md.jobs, md.AS, q);
}

}  // namespace mmg
