// k_pack.hip -- HBM-bound byte kernels around the genotype store: synthetic fill, dtype
// conversion on ingest, SNP-major -> individual-major transposition for the kinship GEMM, and
// per-SNP mean / std (kinship.py:66).  All are streaming kernels with 16-byte accesses.
#include <algorithm>
#include "mmg_internal.h"

namespace mmg {

__device__ __forceinline__ uint64_t hash3(uint64_t seed, uint64_t snp, uint64_t ind) {
  uint64_t x = (snp * 0x9E3779B97F4A7C15ull) ^ (ind * 0xBF58476D1CE4E5B9ull) ^ (seed * 0x94D049BB133111EBull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// grid.x = Npad/16 chunks (x 256 threads -> rows), one thread = 16 individuals of one SNP.
__global__ void fill_hash_kernel(int8_t* __restrict__ S, int64_t M, int32_t N, int32_t Npad,
                                 uint64_t seed, int64_t m0g, uint32_t thr16, uint8_t* __restrict__ X4) {
  const int chunks = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t m = gid / chunks;
  const int c = (int)(gid % chunks);
  if (m >= M) return;
  uint32_t wds[4] = {0, 0, 0, 0}, nib[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = c * 16 + j;
    uint32_t bit = 0;
    if (i < N) bit = ((uint32_t)(hash3(seed, (uint64_t)(m0g + m), (uint64_t)i) >> 48) < thr16) ? 1u : 0u;
    wds[j >> 2] |= bit << (8 * (j & 3));
    nib[j >> 3] |= (bit << 1) << (4 * (j & 7));           // E2M1 twin: 1 -> 0x2 (= 1.0), nibble j = individual 16 c + j
  }
  uint4 v = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  *(uint4*)(S + m * (int64_t)Npad + c * 16) = v;
  if (X4) *(uint2*)(X4 + m * (int64_t)(Npad >> 1) + c * 8) = make_uint2(nib[0], nib[1]);
}

// Structured synthetic genotypes (population structure, so that the REML optimum is interior and a scan has many
// correlated strong hits -- the regime real GWAS lives in): individuals fall into `npop` contiguous populations,
// pop(i) = i * npop / N; SNP m has an ancestral frequency a_m ~ U[0.1, 0.9] and per-population frequencies
// a_m + spread * z_mk with z_mk ~ approx N(0, 1/3) (sum of four uniforms), clamped to [0.02, 0.98]; the genotype is
// Bernoulli(frequency of the individual's population).  All in 16-bit fixed point from the same counter hash as
// fill_hash_kernel, so the CPU checker of the test tree regenerates any row bit for bit (hash_genotypes_structured).
__device__ __forceinline__ uint32_t struct_thr16(uint64_t seed, uint64_t snp, int k, uint32_t spread_q16) {
  const uint64_t s2 = seed ^ 0x5bf03635ca3d9a1full;
  const int64_t anc = 6554 + (int64_t)(((hash3(s2, snp, 1000003ull) >> 48) * 52428ull) >> 16);
  int64_t z = -2 * 65535ll;
#pragma unroll
  for (int j = 0; j < 4; ++j) z += (int64_t)(hash3(s2, snp, 2000003ull + (uint64_t)(4 * k + j)) >> 48);
  int64_t thr = anc + (((int64_t)spread_q16 * z) >> 16);
  thr = thr < 1311 ? 1311 : (thr > 64225 ? 64225 : thr);
  return (uint32_t)thr;
}

__global__ void fill_struct_kernel(int8_t* __restrict__ S, int64_t M, int32_t N, int32_t Npad, uint64_t seed,
                                   int64_t m0g, int npop, uint32_t spread_q16, uint8_t* __restrict__ X4) {
  const int chunks = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t m = gid / chunks;
  const int c = (int)(gid % chunks);
  if (m >= M) return;
  uint32_t wds[4] = {0, 0, 0, 0}, nib[2] = {0, 0};
  int kcur = -1;
  uint32_t thr = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = c * 16 + j;
    uint32_t bit = 0;
    if (i < N) {
      const int k = (int)(((int64_t)i * npop) / N);
      if (k != kcur) { kcur = k; thr = struct_thr16(seed, (uint64_t)(m0g + m), k, spread_q16); }
      bit = ((uint32_t)(hash3(seed, (uint64_t)(m0g + m), (uint64_t)i) >> 48) < thr) ? 1u : 0u;
    }
    wds[j >> 2] |= bit << (8 * (j & 3));
    nib[j >> 3] |= (bit << 1) << (4 * (j & 7));
  }
  *(uint4*)(S + m * (int64_t)Npad + c * 16) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  if (X4) *(uint2*)(X4 + m * (int64_t)(Npad >> 1) + c * 8) = make_uint2(nib[0], nib[1]);
}

void launch_fill_struct(mmg_ctx* ctx, mmg_geno* g, uint64_t seed, int64_t m_global0, int npop, uint32_t spread_q16) {
  const int64_t total = g->M * (int64_t)(g->Npad >> 4);
  hipLaunchKernelGGL(fill_struct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g->d, g->M,
                     g->N, g->Npad, seed, m_global0, npop, spread_q16, g->fp4);
}

void launch_fill_hash(mmg_ctx* ctx, mmg_geno* g, uint64_t seed, int64_t m_global0, uint32_t thr16) {
  const int64_t total = g->M * (int64_t)(g->Npad >> 4);
  const int bs = 256;
  const int64_t nb = (total + bs - 1) / bs;
  hipLaunchKernelGGL(fill_hash_kernel, dim3((unsigned)nb), dim3(bs), 0, ctx->stream, g->d, g->M, g->N,
                     g->Npad, seed, m_global0, thr16, g->fp4);
}

template <typename T>
__global__ void cvt_kernel(const T* __restrict__ src, int8_t* __restrict__ dst, int64_t rows, int32_t N,
                           int64_t ld, int* __restrict__ bad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = rows * (int64_t)N;
  if (gid >= total) return;
  const int64_t r = gid / N;
  const int c = (int)(gid % N);
  const double x = (double)src[gid];
  // the store holds small integers: anything else (dosages, normalised SNPs, NaN, |x| > 127) would be silently
  // rounded or wrapped -- flag it and let the entry point fail
  if (!(x == rint(x)) || !(fabs(x) <= 127.0)) { atomicOr(bad, 1); return; }
  dst[r * ld + c] = (int8_t)(int)x;
}

void launch_cvt_f32(mmg_ctx* ctx, const float* src, int8_t* dst, int64_t rows, int32_t N, int64_t ld, int* d_bad) {
  const int64_t total = rows * (int64_t)N;
  hipLaunchKernelGGL(cvt_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     src, dst, rows, N, ld, d_bad);
}
void launch_cvt_f64(mmg_ctx* ctx, const double* src, int8_t* dst, int64_t rows, int32_t N, int64_t ld, int* d_bad) {
  const int64_t total = rows * (int64_t)N;
  hipLaunchKernelGGL(cvt_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     src, dst, rows, N, ld, d_bad);
}

// Packed genotype rows -> the int8 store.  BITS = 1: genotype i of a row is bit (i & 7) of byte i >> 3; BITS = 2: bits
// 2 (i & 3) .. +1 of byte i >> 2 (least significant first: the bit order of numpy.packbits(bitorder='little') and of a
// PLINK .bed row).  lut: the int8 value of each code, one byte per code (4 codes in a uint32).  One thread writes one
// 16-byte chunk of the store row (16 individuals = 2 or 4 packed bytes); columns >= N are written as zeros, so a
// reused store needs no separate clearing of its padding columns.  HBM-bound on the store write: Npad B per SNP out,
// Npad * BITS / 8 in.
template <int BITS>
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t* __restrict__ src, int64_t row_bytes,
                                                     int8_t* __restrict__ dst, int64_t rows, int32_t N, int32_t Npad,
                                                     uint32_t lut, uint8_t* __restrict__ x4) {
  const int nchunk = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * nchunk) return;
  const int64_t r = gid / nchunk;
  const int c = (int)(gid % nchunk);
  const uint8_t* row = src + r * row_bytes;
  constexpr int NB = 2 * BITS;                           // packed bytes per 16 genotypes
  uint32_t bits = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int64_t o = (int64_t)c * NB + b;
    if (o < row_bytes) bits |= (uint32_t)row[o] << (8 * b);
  }
  uint32_t w[4] = {0, 0, 0, 0}, nib[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t code = (bits >> (BITS * j)) & ((1u << BITS) - 1u);
    const uint32_t v = (c * 16 + j < N) ? ((lut >> (8 * code)) & 0xffu) : 0u;
    w[j >> 2] |= v << (8 * (j & 3));
    nib[j >> 3] |= ((v & 1u) << 1) << (4 * (j & 7));        // the store's E2M1 twin: bit 0 of the value as 0x0 / 0x2
  }
  *(uint4*)(dst + r * (int64_t)Npad + c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  if (x4) *(uint2*)(x4 + r * (int64_t)(Npad >> 1) + c * 8) = make_uint2(nib[0], nib[1]);
}

// [rows x N] contiguous int8 -> rows of the padded store (columns >= N zero): the device half of a staged upload.
// A row starts at byte r * N of the source -- any alignment -- so a thread fetches the 5 aligned dwords that cover its
// 16 bytes and realigns them with v_alignbyte (16 single-byte loads per thread ran at a third of the copy rate of the
// link: 19 GB/s end to end).  The staging buffer carries 32 bytes of slack behind the last row.
__global__ __launch_bounds__(256) void pitch_rows_kernel(const int8_t* __restrict__ src, int8_t* __restrict__ dst,
                                                         int64_t rows, int32_t N, int32_t Npad) {
  const int nchunk = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * nchunk) return;
  const int64_t r = gid / nchunk;
  const int c = (int)(gid % nchunk);
  const int nvalid = min(16, max(0, N - c * 16));
  uint32_t o[4] = {0, 0, 0, 0};
  if (nvalid > 0) {
    const int64_t a = r * (int64_t)N + c * 16;          // first source byte
    const uint32_t* s32 = (const uint32_t*)(src + (a & ~(int64_t)3));
    const uint32_t sh = (uint32_t)(a & 3);
    uint32_t w[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) w[k] = s32[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh);
    if (nvalid < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int keep = min(4, max(0, nvalid - 4 * j));   // bytes of dword j inside the row
        o[j] = keep == 4 ? o[j] : (keep == 0 ? 0u : (o[j] & ((1u << (8 * keep)) - 1u)));
      }
    }
  }
  *(uint4*)(dst + r * (int64_t)Npad + c * 16) = make_uint4(o[0], o[1], o[2], o[3]);
}

void launch_pitch_rows(mmg_ctx* ctx, const int8_t* src, int8_t* dst, int64_t rows, int32_t N, int32_t Npad) {
  const int64_t total = rows * (Npad >> 4);
  if (total <= 0) return;
  hipLaunchKernelGGL(pitch_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, src, dst, rows,
                     N, Npad);
}

// the reverse for downloads: rows of the padded store, packed N bytes apiece (one byte per thread: downloads are test /
// diagnostic traffic; what matters is that the copy behind it is ONE contiguous transfer)
__global__ __launch_bounds__(256) void unpitch_rows_kernel(const int8_t* __restrict__ src, int8_t* __restrict__ dst,
                                                           int64_t rows, int32_t N, int32_t Npad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * N) return;
  const int64_t r = gid / N;
  dst[gid] = src[r * (int64_t)Npad + (gid - r * N)];
}

void launch_unpitch_rows(mmg_ctx* ctx, const int8_t* src, int8_t* dst, int64_t rows, int32_t N, int32_t Npad) {
  const int64_t total = rows * N;
  if (total <= 0) return;
  hipLaunchKernelGGL(unpitch_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, src, dst, rows,
                     N, Npad);
}

void launch_unpack(mmg_ctx* ctx, const uint8_t* src, int64_t row_bytes, int8_t* dst, int64_t rows, int32_t N,
                   int32_t Npad, int bits, uint32_t lut, uint8_t* x4) {
  const int64_t total = rows * (Npad >> 4);
  if (total <= 0) return;
  if (bits == 1)
    hipLaunchKernelGGL(unpack_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, src, row_bytes,
                       dst, rows, N, Npad, lut, x4);
  else
    hipLaunchKernelGGL(unpack_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, src, row_bytes,
                       dst, rows, N, Npad, lut, x4);
}

// 64 SNPs x 64 individuals per block; out[i][m] = valid ? mul * s + add : 0.
__global__ __launch_bounds__(256) void transpose_kernel(const int8_t* __restrict__ S, int64_t M, int32_t N,
                                                        int32_t Npad, int8_t* __restrict__ Xt, int64_t Mk,
                                                        int mul, int add, int64_t m_begin, int thr) {
  __shared__ int8_t tile[64][64 + 4];
  const int t = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * 64;       // column of Xt; SNP row m_begin + m0
  const int i0 = blockIdx.y * 64;
  {
    const int r = t >> 2, c = t & 3;
    const uint4 v = *(const uint4*)(S + (m_begin + m0 + r) * (int64_t)Npad + i0 + c * 16);
    uint32_t* dstw = (uint32_t*)&tile[r][c * 16];
    dstw[0] = v.x; dstw[1] = v.y; dstw[2] = v.z; dstw[3] = v.w;
  }
  __syncthreads();
  const int i = t >> 2, mc = t & 3;
  uint32_t wds[4] = {0, 0, 0, 0};
  const bool ivalid = (i0 + i) < N;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t m = m_begin + m0 + mc * 16 + j;
    int v = 0;
    if (ivalid && m < M) {
      const int sv = (int)tile[mc * 16 + j][i];
      v = thr > 0 ? (sv >= thr ? 1 : 0) : mul * sv + add;     // indicator [s >= thr] or affine
    }
    wds[j >> 2] |= ((uint32_t)(v & 0xff)) << (8 * (j & 3));
  }
  *(uint4*)(Xt + (int64_t)(i0 + i) * Mk + m0 + mc * 16) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
}

// Weighted twin for the exact GRM (api.hip:kinship_grm_i8_into): one pass over S writes the plain image Xq[i][m] = s
// and D digit images Xp[d][i][m] = dig[d][m] * s, where dig holds non-negative digits of the per-SNP weight 1/std^2 whose
// range was chosen so that every product fits int8.  Images are [Npad x Mk], digit image d at Xp + d * Npad * Mk.
template <int D>
__global__ __launch_bounds__(256) void transpose_digits_kernel(const int8_t* __restrict__ S, int64_t M, int32_t N,
                                                               int32_t Npad, int8_t* __restrict__ Xq,
                                                               int8_t* __restrict__ Xp, int64_t Mk, int64_t m_begin,
                                                               const int8_t* __restrict__ dig /*[D][Mk]*/) {
  __shared__ int8_t tile[64][64 + 4];
  const int t = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int i0 = blockIdx.y * 64;
  {
    const int r = t >> 2, c = t & 3;
    const uint4 v = *(const uint4*)(S + (m_begin + m0 + r) * (int64_t)Npad + i0 + c * 16);
    uint32_t* dstw = (uint32_t*)&tile[r][c * 16];
    dstw[0] = v.x; dstw[1] = v.y; dstw[2] = v.z; dstw[3] = v.w;
  }
  __syncthreads();
  const int i = t >> 2, mc = t & 3;
  uint32_t wq[4] = {0, 0, 0, 0};
  uint32_t wp[D][4];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) wp[d][e] = 0;
  const bool ivalid = (i0 + i) < N;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t mk = m0 + mc * 16 + j;
    int sv = 0;
    if (ivalid && m_begin + mk < M) sv = (int)tile[mc * 16 + j][i];
    wq[j >> 2] |= ((uint32_t)(sv & 0xff)) << (8 * (j & 3));
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int v = sv * (int)dig[(int64_t)d * Mk + mk];
      wp[d][j >> 2] |= ((uint32_t)(v & 0xff)) << (8 * (j & 3));
    }
  }
  const int64_t off = (int64_t)(i0 + i) * Mk + m0 + mc * 16;
  *(uint4*)(Xq + off) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
#pragma unroll
  for (int d = 0; d < D; ++d)
    *(uint4*)(Xp + (int64_t)d * Npad * Mk + off) = make_uint4(wp[d][0], wp[d][1], wp[d][2], wp[d][3]);
}

// FP4 image of a BINARY store for the FP4 kinship GEMM (gemm_i8_w4tr.h FmtF4): X4[m][i / 2] holds the genotypes of
// individuals i (low nibble: even i) as E2M1 nibbles, 1 -> 0x2 (= 1.0), 0 -> 0x0.  One thread per 16 output bytes
// (32 genotypes).  HBM-bound: Npad in + Npad / 2 out per SNP.
// thr = 1 on a binary store: the genotypes themselves; thr >= 1 in general: the indicator [s >= thr] (the two products of
// the 'diploid_int' IBS kinship, kinship.py:33-41).
template <bool BINARY>
__global__ __launch_bounds__(256) void pack_fp4_kernel(const int8_t* __restrict__ S, int64_t rows, int32_t Npad,
                                                       uint8_t* __restrict__ X4, int thr) {
  const int nchunk = Npad >> 5;                          // 16-byte output chunks per row
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * nchunk) return;
  const int64_t r = gid / nchunk;
  const int c = (int)(gid % nchunk);
  const uint4* src = (const uint4*)(S + r * (int64_t)Npad + c * 32);
  const uint4 a = src[0], b = src[1];
  uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  if (!BINARY) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t x = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) x |= ((int)(int8_t)((w[q] >> (8 * j)) & 0xff) >= thr ? 1u : 0u) << (8 * j);
      w[q] = x;
    }
  }
  uint32_t o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // dwords 2q, 2q+1 = 8 genotype bytes (0/1) -> 8 nibbles (0/2): byte j of the pair lands in nibble j
    const uint32_t lo = w[2 * q] & 0x01010101u, hi = w[2 * q + 1] & 0x01010101u;
    const uint32_t l4 = (lo | (lo >> 4)) & 0x00110011u, h4 = (hi | (hi >> 4)) & 0x00110011u;
    const uint32_t l2 = (l4 | (l4 >> 8)) & 0x0000ffffu, h2 = (h4 | (h4 >> 8)) & 0x0000ffffu;
    o[q] = ((l2 & 0x1111u) | ((h2 & 0x1111u) << 16)) << 1;
  }
  *(uint4*)(X4 + r * (int64_t)(Npad >> 1) + c * 16) = make_uint4(o[0], o[1], o[2], o[3]);
}

// Both indicator images of a 0/1/2 store from ONE read of it (round 6): X4a = [s >= 1], X4b = [s >= 2] -- the two operands of the
// 'diploid_int' IBS kinship (kinship.py:33-41), stacked by the caller into one 2 M-row image so that u'u + v'v is ONE GEMM.
__global__ __launch_bounds__(256) void pack_fp4_two_kernel(const int8_t* __restrict__ S, int64_t rows, int32_t Npad,
                                                           uint8_t* __restrict__ X4a, uint8_t* __restrict__ X4b) {
  const int nchunk = Npad >> 5;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * nchunk) return;
  const int64_t r = gid / nchunk;
  const int c = (int)(gid % nchunk);
  const uint4* src = (const uint4*)(S + r * (int64_t)Npad + c * 32);
  const uint4 a = src[0], b = src[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int img = 0; img < 2; ++img) {
    uint32_t x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) v |= ((int)(int8_t)((w[q] >> (8 * j)) & 0xff) >= 1 + img ? 1u : 0u) << (8 * j);
      x[q] = v;
    }
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t lo = x[2 * q], hi = x[2 * q + 1];
      const uint32_t l4 = (lo | (lo >> 4)) & 0x00110011u, h4 = (hi | (hi >> 4)) & 0x00110011u;
      const uint32_t l2 = (l4 | (l4 >> 8)) & 0x0000ffffu, h2 = (h4 | (h4 >> 8)) & 0x0000ffffu;
      o[q] = ((l2 & 0x1111u) | ((h2 & 0x1111u) << 16)) << 1;
    }
    *(uint4*)((img ? X4b : X4a) + r * (int64_t)(Npad >> 1) + c * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
void launch_pack_fp4_two(mmg_ctx* ctx, const int8_t* S, int64_t rows, int32_t Npad, uint8_t* X4a, uint8_t* X4b) {
  const int64_t total = rows * (Npad >> 5);
  if (total <= 0) return;
  hipLaunchKernelGGL(pack_fp4_two_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, S, rows, Npad, X4a, X4b);
}

// binary: every stored value is 0 or 1 and thr == 1 (the bytes are the indicator already)
void launch_pack_fp4_on(mmg_ctx* ctx, hipStream_t stream, const int8_t* S, int64_t rows, int32_t Npad, uint8_t* X4,
                        int thr, bool binary) {
  (void)ctx;
  const int64_t total = rows * (Npad >> 5);
  if (total <= 0) return;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (binary && thr == 1) hipLaunchKernelGGL(pack_fp4_kernel<true>, grid, dim3(256), 0, stream, S, rows, Npad, X4, thr);
  else hipLaunchKernelGGL(pack_fp4_kernel<false>, grid, dim3(256), 0, stream, S, rows, Npad, X4, thr);
}

// SNP-major twin for the transposed-read kinship GEMM (gemm_i8_w4tr.h): nothing is transposed -- image d holds row m
// of the store times digit d of that SNP's weight, Xp[d][m - m_begin][i] = dig[d][m - m_begin] * s_mi, and the plain
// operand is the store itself.  One thread owns one 16-byte column chunk over a slab of GRM_SLAB rows, which also gives
// it the weighted column sums c1[i] = sum_m coef[m] s_mi (the a b (s 1' + 1 s') term of z z') of its slab in 16 fp64
// registers: partial[slab][Npad], reduced over the slabs in fixed order by grm_colsum_reduce_kernel (deterministic).
// HBM-bound: Npad in + D * Npad out per SNP, fully coalesced (round 2: 16.3 ms of a 64 ms GRM for the transposing
// version with its byte-wise LDS reads).
constexpr int GRM_SLAB = 512;

// SHIFT (round 4): a store of 0 / 1 / 2 is read as s - 1 in {-1, 0, 1} (the GRM does not change when a constant is added
// to a SNP: only s - mean enters), whose products with a 7-bit digit fit int8 -- four planes of 7 bits instead of five of 6.
// Columns >= N stay zero.  Image D is then the plain shifted store, the GEMM's second operand (api.hip).
template <int D, bool NEG, bool SHIFT = false>
__global__ __launch_bounds__(256) void grm_scale_rows_kernel(const int8_t* __restrict__ S, int64_t rows_valid, int64_t Mk,
                                                             int32_t Npad, int8_t* __restrict__ Xp,
                                                             const int8_t* __restrict__ dig /*[D][Mk]*/,
                                                             const double* __restrict__ coef /*[Mk]*/,
                                                             double* __restrict__ partial /*[slabs][Npad]*/, int32_t N) {
  const int nchunk = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nslab = (Mk + GRM_SLAB - 1) / GRM_SLAB;
  if (gid >= nslab * nchunk) return;
  const int64_t slab = gid / nchunk;
  const int c = (int)(gid % nchunk);
  const int64_t r0 = slab * GRM_SLAB, r1 = min(Mk, r0 + GRM_SLAB);
  double acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0;
  for (int64_t r = r0; r < r1; ++r) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows_valid) v = *(const uint4*)(S + r * (int64_t)Npad + c * 16);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if (SHIFT) {
      // byte-wise s - 1 without borrows between bytes: (s | 0x80) - 1 never crosses a byte, and ^ 0x80 puts the sign back
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t one = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) one |= (c * 16 + q * 4 + b < N ? 1u : 0u) << (8 * b);
        w[q] = ((w[q] | 0x80808080u) - one) ^ 0x80808080u;
      }
    }
    const double cf = coef[r];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = fma(cf, (double)(int)(int8_t)((w[e >> 2] >> (8 * (e & 3))) & 0xff), acc[e]);
    if (SHIFT) *(uint4*)(Xp + ((int64_t)D * Mk + r) * Npad + c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int dg = (int)dig[(int64_t)d * Mk + r];       // 0 .. 127, digit * |s| <= 127
      uint32_t o[4];
      if (!NEG && !SHIFT) {
        // non-negative genotype bytes: every byte product stays below 128, so one dword multiply does four of them
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = w[q] * (uint32_t)dg;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t x = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) x |= ((uint32_t)(((int)(int8_t)((w[q] >> (8 * b)) & 0xff) * dg) & 0xff)) << (8 * b);
          o[q] = x;
        }
      }
      *(uint4*)(Xp + ((int64_t)d * Mk + r) * Npad + c * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
  double* out = partial + slab * Npad + c * 16;
#pragma unroll
  for (int e = 0; e < 16; ++e) out[e] = acc[e];
}

__global__ void grm_colsum_reduce_kernel(const double* __restrict__ partial, int64_t nslab, int32_t Npad,
                                         double* __restrict__ c1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad) return;
  double s = 0.0;
  for (int64_t k = 0; k < nslab; ++k) s += partial[k * Npad + i];
  c1[i] = s;
}

__global__ __launch_bounds__(256) void row_sums_f64_kernel(const double* __restrict__ A, int64_t N, double* __restrict__ rows,
                                                           double* __restrict__ diag) {
  const int64_t i = blockIdx.x;
  const double* a = A + i * N;
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < N; j += 256) s += a[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { rows[i] = (w[0] + w[1]) + (w[2] + w[3]); diag[i] = a[i]; }
}

__global__ void scale_f64_kernel(double* __restrict__ x, int64_t n, double f) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) { double2 v = *(double2*)(x + i); v.x *= f; v.y *= f; *(double2*)(x + i) = v; }
  else if (i < n) x[i] *= f;
}

__global__ void add_scalar_f64_kernel(double* __restrict__ x, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += v;
}
void launch_add_scalar_f64(mmg_ctx* ctx, double* x, int64_t n, double v) {
  if (n > 0) hipLaunchKernelGGL(add_scalar_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n, v);
}

void launch_row_sums_f64(mmg_ctx* ctx, const double* A, int64_t N, double* rows, double* diag) {
  hipLaunchKernelGGL(row_sums_f64_kernel, dim3((unsigned)N), dim3(256), 0, ctx->stream, A, N, rows, diag);
}

void launch_scale_f64(mmg_ctx* ctx, double* x, int64_t n, double f) {
  const int64_t pairs = (n + 1) / 2;
  hipLaunchKernelGGL(scale_f64_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, ctx->stream, x, n, f);
}

int64_t grm_partial_doubles(int64_t Mk, int32_t Npad) { return (Mk + GRM_SLAB - 1) / GRM_SLAB * (int64_t)Npad; }

// S: first row of the chunk in the store (row stride Npad); rows_valid: rows of the chunk that exist in the store
// n_shift > 0: the store holds 0 / 1 / 2 and is read as s - 1 over its n_shift individuals; image D is the shifted store
void launch_grm_scale_rows(mmg_ctx* ctx, const int8_t* S, int64_t rows_valid, int64_t Mk, int32_t Npad, bool neg,
                           int8_t* Xp, const int8_t* dig, int D, const double* coef, double* partial, double* c1,
                           int32_t n_shift) {
  const int64_t nslab = (Mk + GRM_SLAB - 1) / GRM_SLAB;
  const int64_t total = nslab * (Npad >> 4);
  const dim3 grid((unsigned)((total + 255) / 256));
#define MMG_GS(D_, NEG_)                                                                                              \
  hipLaunchKernelGGL((grm_scale_rows_kernel<D_, NEG_>), grid, dim3(256), 0, ctx->stream, S, rows_valid, Mk, Npad, Xp, dig, \
                     coef, partial, 0)
#define MMG_GSS(D_)                                                                                                   \
  hipLaunchKernelGGL((grm_scale_rows_kernel<D_, true, true>), grid, dim3(256), 0, ctx->stream, S, rows_valid, Mk, Npad, Xp, \
                     dig, coef, partial, n_shift)
  if (n_shift > 0) { if (D == 3) MMG_GSS(3); else if (D == 4) MMG_GSS(4); else if (D == 5) MMG_GSS(5); else if (D == 6) MMG_GSS(6); }
  else if (D == 0) MMG_GS(0, false);                       // only the weighted column sums (the fused 4-plane GEMM scales in registers)
  else if (neg) { if (D == 3) MMG_GS(3, true); else if (D == 4) MMG_GS(4, true); else if (D == 5) MMG_GS(5, true); else if (D == 6) MMG_GS(6, true); }
  else { if (D == 1) MMG_GS(1, false); else if (D == 3) MMG_GS(3, false); else if (D == 4) MMG_GS(4, false); else if (D == 5) MMG_GS(5, false); else if (D == 6) MMG_GS(6, false); }
  // (any other plane count launches nothing: api.hip admits 3..6 -- a count that fell through to the 6-plane kernel wrote
  // six images into a buffer sized for fewer)
#undef MMG_GS
#undef MMG_GSS
  hipLaunchKernelGGL(grm_colsum_reduce_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, ctx->stream, partial,
                     nslab, Npad, c1);
}

void launch_transpose_digits(mmg_ctx* ctx, const mmg_geno* g, int8_t* Xq, int8_t* Xp, int64_t Mk, int64_t m_begin,
                             const int8_t* dig, int D) {
  dim3 grid((unsigned)(Mk / 64), (unsigned)(g->Npad / 64));
#define MMG_TD(D_)                                                                                                  \
  hipLaunchKernelGGL(transpose_digits_kernel<D_>, grid, dim3(256), 0, ctx->stream, g->d, g->M, g->N, g->Npad, Xq, Xp, \
                     Mk, m_begin, dig)
  if (D == 3) MMG_TD(3);
  else if (D == 4) MMG_TD(4);
  else if (D == 5) MMG_TD(5);
  else if (D == 6) MMG_TD(6);
#undef MMG_TD
}

void launch_transpose(mmg_ctx* ctx, const mmg_geno* g, int8_t* Xt, int64_t Mk, int mul, int add, int64_t m_begin,
                      int thr) {
  dim3 grid((unsigned)(Mk / 64), (unsigned)(g->Npad / 64));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, ctx->stream, g->d, g->M, g->N, g->Npad, Xt, Mk,
                     mul, add, m_begin, thr);
}

// Column sums of the store, r[i] += sum over rows [0, M) of S[m][i] (exact, 64-bit atomics): the rank-1 terms of the
// IBS kinship written on the raw 0/1 genotypes (api.hip:kinship_counts_i8).  One thread per 16-byte column chunk and
// slab of rows.
__global__ __launch_bounds__(256) void colsum_kernel(const int8_t* __restrict__ S, int64_t M, int32_t Npad,
                                                     int64_t rows_per_slab, unsigned long long* __restrict__ r) {
  const int nchunk = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(gid % nchunk);
  const int64_t m0 = (gid / nchunk) * rows_per_slab, m1 = min(M, m0 + rows_per_slab);
  if (m0 >= m1) return;
  int acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0;
  for (int64_t m = m0; m < m1; ++m) {
    const uint4 v = *(const uint4*)(S + m * (int64_t)Npad + c * 16);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += (int)(int8_t)((w[e >> 2] >> (8 * (e & 3))) & 0xff);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if (acc[e]) atomicAdd(r + c * 16 + e, (unsigned long long)(long long)acc[e]);
}

void launch_colsum(mmg_ctx* ctx, const mmg_geno* g, unsigned long long* r) {
  const int64_t rows_per_slab = 1024;                    // |sum| <= 1024 * 127 per thread: int32 partials
  const int64_t slabs = (g->M + rows_per_slab - 1) / rows_per_slab;
  const int64_t total = slabs * (g->Npad >> 4);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g->d, g->M, g->Npad,
                     rows_per_slab, r);
}

// one wave per SNP: exact integer sum and sum of squares -> mean, population std
__global__ __launch_bounds__(256) void snp_stats_kernel(const int8_t* __restrict__ S, int64_t M, int32_t N,
                                                        int32_t Npad, double* __restrict__ mean,
                                                        double* __restrict__ sd) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  long long s1 = 0, s2 = 0;
  for (int c = lane; c < (Npad >> 4); c += 64) {
    const uint4 v = *(const uint4*)(S + m * (int64_t)Npad + c * 16);
    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int x = (int)(int8_t)((wds[j >> 2] >> (8 * (j & 3))) & 0xff);
      s1 += x; s2 += x * x;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (lane == 0) {
    const double mu = (double)s1 / (double)N;
    const double var = (double)s2 / (double)N - mu * mu;
    mean[m] = mu;
    sd[m] = sqrt(var > 0.0 ? var : 0.0);
  }
}

void launch_snp_stats(mmg_ctx* ctx, const mmg_geno* g, double* mean, double* sd) {
  hipLaunchKernelGGL(snp_stats_kernel, dim3((unsigned)((g->M + 3) / 4)), dim3(256), 0, ctx->stream, g->d,
                     g->M, g->N, g->Npad, mean, sd);
}

// Weights of the exact GRM from the per-SNP statistics, on the device (round 4: the host loops over M means / stds and
// their 16 B per SNP of download were 5 ms of a 43 ms call at C3).  Pass 1: per block of GRM_WB SNPs the largest and the
// smallest weight 1 / sd^2, the block's share of c0 = sum mean^2 / sd^2 (summed in a fixed tree: deterministic) and the
// number of SNPs without variation -- 4 doubles per block go to the host, which picks the digit step.
constexpr int GRM_WB = 4096;

__global__ __launch_bounds__(256) void grm_weight_stats_kernel(const double* __restrict__ mean, const double* __restrict__ sd,
                                                               int64_t M, double* __restrict__ out /*[blocks][4]*/) {
  const int64_t m0 = (int64_t)blockIdx.x * GRM_WB;
  double wmax = 0.0, wmin = 1e300, c0 = 0.0, bad = 0.0;
  for (int k = threadIdx.x; k < GRM_WB; k += 256) {
    const int64_t m = m0 + k;
    if (m >= M) break;
    const double s = sd[m];
    if (!(s > 0.0)) { bad += 1.0; continue; }
    const double w = 1.0 / (s * s), mu = mean[m];
    wmax = fmax(wmax, w);
    wmin = fmin(wmin, w);
    c0 += mu * mu * w;
  }
  __shared__ double red[4][256];
  red[0][threadIdx.x] = wmax; red[1][threadIdx.x] = wmin; red[2][threadIdx.x] = c0; red[3][threadIdx.x] = bad;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] = fmax(red[0][threadIdx.x], red[0][threadIdx.x + o]);
      red[1][threadIdx.x] = fmin(red[1][threadIdx.x], red[1][threadIdx.x + o]);
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
      red[3][threadIdx.x] += red[3][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) out[(int64_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

int64_t grm_weight_blocks(int64_t M) { return (M + GRM_WB - 1) / GRM_WB; }

void launch_grm_weight_stats(mmg_ctx* ctx, const double* mean, const double* sd, int64_t M, double* out) {
  hipLaunchKernelGGL(grm_weight_stats_kernel, dim3((unsigned)grm_weight_blocks(M)), dim3(256), 0, ctx->stream, mean, sd, M, out);
}

// Pass 2, per chunk of Mk SNPs starting at SNP mb: the D digits (bd bits each) of llrint(weight / step) and the
// coefficient -mean * weight of the rank-one terms; rows past M are zero.  The same IEEE operations the host loop did.
__global__ __launch_bounds__(256) void grm_digits_kernel(const double* __restrict__ mean, const double* __restrict__ sd,
                                                         int64_t mb, int64_t M, int64_t Mk, double step, int bd, int D,
                                                         int8_t* __restrict__ dig /*[D][Mk]*/, double* __restrict__ coef,
                                                         int64_t stream_pos) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= Mk) return;
  long long Z = 0;
  double cf = 0.0;
  if (mb + k < M) {
    const double s = sd[mb + k];
    const double w = 1.0 / (s * s);
    // dithered rounding: floor(w / step + u) with u in [0, 1) a hash of the SNP's index.  Round-to-nearest gives SNPs of
    // equal weight the SAME rounding error, and with weights 1 / (p (1 - p)), p = count / N, few distinct weights carry most
    // SNPs -- the errors added up like M instead of sqrt(M) (4.8e-5 on entries of 7e4 over 70,000 SNPs of frequency one
    // half, test_gpu_round3).  Dithered, the errors of different SNPs are independent and unbiased whatever the weights.
    // The index is the SNP's position in the accumulator's whole stream (stream_pos = SNPs of the earlier calls): hashed from
    // the in-call index alone, every call of a chunked pass drew the same u sequence and a position's bias (u - 1/2) added up
    // coherently over the ~100 calls of a streamed kinship (advisor r4) -- and a stream now rounds the same however it is cut.
    unsigned long long h = (unsigned long long)(stream_pos + mb + k) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
    const double u = (double)(h >> 11) * 0x1.0p-53;
    Z = (long long)floor(w / step + u);
    cf = -mean[mb + k] * w;
  }
  const long long mask = (1ll << bd) - 1, zmax = (1ll << (bd * D)) - 1;
  Z = Z > zmax ? zmax : Z;
  for (int d = 0; d < D; ++d) {
    dig[(int64_t)d * Mk + k] = (int8_t)(Z & mask);
    Z >>= bd;
  }
  coef[k] = cf;
}

void launch_grm_digits(mmg_ctx* ctx, const double* mean, const double* sd, int64_t mb, int64_t M, int64_t Mk, double step,
                       int bd, int D, int8_t* dig, double* coef, int64_t stream_pos) {
  hipLaunchKernelGGL(grm_digits_kernel, dim3((unsigned)((Mk + 255) / 256)), dim3(256), 0, ctx->stream, mean, sd, mb, M, Mk,
                     step, bd, D, dig, coef, stream_pos);
}

__global__ void add_into_f64_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

void launch_add_into_f64(mmg_ctx* ctx, double* dst, const double* src, int64_t n) {
  hipLaunchKernelGGL(add_into_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dst, src, n);
}

// max |s| over a 16-byte aligned range (write paths of the genotype store keep an upper bound of it)
// out[0] = max |s|, out[1] = max(-s) (0 for a store without negative values).  PACK: the same pass writes the E2M1 twin of
// the bytes it reads (bit 0 of byte i -> nibble i, 0x0 / 0x2): 8 bytes out per 16 in, no second sweep over the store.
template <bool PACK>
__global__ __launch_bounds__(256) void absmax_i8_kernel(const int8_t* __restrict__ p, int64_t n16, int* __restrict__ out,
                                                        uint8_t* __restrict__ x4) {
  int mx = 0, ng = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = *(const uint4*)(p + i * 16);
    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int x = (int)(int8_t)((wds[j >> 2] >> (8 * (j & 3))) & 0xff);
      mx = max(mx, x < 0 ? -x : x);
      ng = max(ng, -x);
    }
    if (PACK) {
      uint32_t o[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {                        // dwords 2q, 2q+1 = 8 genotype bytes -> 8 nibbles (pack_fp4_kernel)
        const uint32_t lo = wds[2 * q] & 0x01010101u, hi = wds[2 * q + 1] & 0x01010101u;
        const uint32_t l4 = (lo | (lo >> 4)) & 0x00110011u, h4 = (hi | (hi >> 4)) & 0x00110011u;
        const uint32_t l2 = (l4 | (l4 >> 8)) & 0x0000ffffu, h2 = (h4 | (h4 >> 8)) & 0x0000ffffu;
        o[q] = ((l2 & 0x1111u) | ((h2 & 0x1111u) << 16)) << 1;
      }
      *(uint2*)(x4 + i * 8) = make_uint2(o[0], o[1]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o)); ng = max(ng, __shfl_xor(ng, o)); }
  if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(out, mx);
  if ((threadIdx.x & 63) == 0 && ng > 0) atomicMax(out + 1, ng);
}

void launch_absmax_i8(mmg_ctx* ctx, const int8_t* p, int64_t bytes, int* d_out, uint8_t* x4) {
  const int64_t n16 = bytes >> 4;
  if (n16 <= 0) return;
  const int64_t nb = std::min<int64_t>((n16 + 255) / 256, 4096);
  if (x4) hipLaunchKernelGGL(absmax_i8_kernel<true>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, p, n16, d_out, x4);
  else hipLaunchKernelGGL(absmax_i8_kernel<false>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, p, n16, d_out, x4);
}

}  // namespace mmg
