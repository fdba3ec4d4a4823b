// k_scan.hip -- the EMMAX per-SNP scan (replaces the chunked GEMM + per-SNP lstsq loop of
// linear_models.py:1316-1349 with its closed form):
//     num_m = (s_m . w)^2,  den_m = s_m' A s_m,  rss_m = h0_rss - num_m/den_m,
//     F_m = (h0_rss/rss_m - 1) * df2,  p_m = f.sf(F_m, 1, df2).
//
// den is the hot part (2 N^2 flop per SNP).  Genotypes are small integers, so the product
// S . A is computed EXACTLY on the int8 matrix cores after writing the (fp64) matrix as D
// balanced base-256 digits ("Ozaki" splitting):  2*A_jk = step * sum_d 256^d z_d[j][k], k < j.
// Only the strictly lower triangle is stored (A symmetric: s'As = sum_i A_ii s_i^2 +
// sum_{k<j} 2 A_jk s_j s_k), which halves the MFMA work; the diagonal term and s.w are
// evaluated in fp64 by the HBM-bound finalize kernel (one wave per 8 SNP rows, coalesced
// 16-byte genotype reads), which also evaluates F and the p-value.
//
// All integer partial sums are exact and accumulated with 64-bit integer atomics, so results
// are bitwise reproducible from run to run and independent of the tile schedule.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstdio>
#include "gemm_i8_core.h"
#include "gemm_i8_ring.h"
#include "gemm_i8_w4.h"
#include "mmg_internal.h"

namespace mmg {

// ------------------------------------------------------------------ model quantisation
__global__ void absmax_offdiag_kernel(const double* __restrict__ A, int32_t N, unsigned long long* out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (gid < (int64_t)N * N) {
    const int i = (int)(gid / N), j = (int)(gid % N);
    if (j < i) v = fabs(A[gid]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0 && v > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(v));
}

void launch_absmax_offdiag(mmg_ctx* ctx, const double* A, int32_t N, unsigned long long* out_bits) {
  const int64_t total = (int64_t)N * N;
  hipLaunchKernelGGL(absmax_offdiag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, A,
                     N, out_bits);
}

// one thread = 16 consecutive k of row j; Bq[d][j][k] = digit d of rint(2 A[j][k] / step), k < j
__global__ void quantize_kernel(const double* __restrict__ A, int32_t N, int32_t Npad, int D, double inv_step,
                                int8_t* __restrict__ Bq, double* __restrict__ diag) {
  const int chunks = Npad >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)Npad * chunks) return;
  const int j = (int)(gid / chunks), c = (int)(gid % chunks);
  uint32_t out[6][4];
#pragma unroll
  for (int d = 0; d < 6; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) out[d][e] = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = c * 16 + e;
    long long Z = 0;
    if (j < N && k < j) Z = __double2ll_rn(2.0 * A[(int64_t)j * N + k] * inv_step);
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      if (d < D) {
        const long long z = ((Z + 128) & 255) - 128;
        Z = (Z - z) >> 8;
        out[d][e >> 2] |= ((uint32_t)(z & 0xff)) << (8 * (e & 3));
      }
    }
  }
  for (int d = 0; d < D; ++d)
    *(uint4*)(Bq + ((int64_t)d * Npad + j) * Npad + c * 16) = make_uint4(out[d][0], out[d][1], out[d][2], out[d][3]);
  if (c == 0) diag[j] = (j < N) ? A[(int64_t)j * N + j] : 0.0;
}

void launch_quantize(mmg_ctx* ctx, const double* A, int32_t N, int32_t Npad, int D, double inv_step, int8_t* Bq,
                     double* diag) {
  const int64_t total = (int64_t)Npad * (Npad >> 4);
  hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, A, N, Npad,
                     D, inv_step, Bq, diag);
}

// ------------------------------------------------------------------ quadratic-form GEMM
// Workgroup -> (SNP block of 256, job group).  Blocks b, b+8, ... share an XCD (observed
// placement, speed only): a cohort of 32 consecutive such blocks works on AS SNP blocks x G
// job groups, so the S rows are L2 hits for G workgroups and the digit tiles for AS.
template <int ABLATE>
__global__ __launch_bounds__(NTHREADS, 2) void scan_quad_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
  const int8_t* Q = S + (int64_t)sb * TN * ldS;
  unsigned long long qacc[2] = {0ull, 0ull};
  for (int jj = j0; jj < j1; ++jj) {
    const int2 jb = jobs[jj];
    const int d = jb.x, J = jb.y;
    const int8_t* P = Bq + (int64_t)d * digit_stride + (int64_t)J * TM * ldB;
    v16i acc[4][2];
    gemm_tile_i8<ABLATE>(P, ldB, Q, ldS, 0, 2 * (J + 1), lds, acc);
    // Epilogue: lane holds SNP column n = wn*64 + nn*32 + r and rows j = wm*128 + m*32 +
    // (reg&3) + 8*(reg>>2) + 4*h of T = Z_d(J-tile rows) . S^T; multiply by s[snp][256J + j].
    // Those genotype bytes are the Q tiles of the job's last two K steps (k = 256J .. 256J+255),
    // which are still in LDS: K step 2J sits in buffer 0, 2J+1 in buffer 1 (stage parity), and
    // no stage was issued during the last step.  Wave row half wm reads buffer wm: 8 conflict-free
    // ds_read_b128 per SNP instead of 16 scattered global dword loads.
    const char* qbuf = lds + wm * BUF_BYTES + TILE_BYTES;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int qrow = wn * 64 + nn * 32 + r;
      long long part = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const v4i ch = lds_frag(qbuf, qrow, 2 * m + cc);       // bytes m*32 + cc*16 .. +16 of the row
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {                        // g4 = 2*cc + gp; dword (gp*2 + h)
            const int wd = h ? ch[gp * 2 + 1] : ch[gp * 2];
            const int g4 = 2 * cc + gp;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              part += (long long)acc[m][nn][g4 * 4 + e] * (long long)(int)(int8_t)((wd >> (8 * e)) & 0xff);
          }
        }
      qacc[nn] += ((unsigned long long)part) << (8 * d);
    }
    __syncthreads();   // the next job's prologue refills buffer 0
  }
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    unsigned long long v = qacc[nn];
    v += __shfl_xor(v, 32);
    if (h == 0) atomicAdd(q + (int64_t)sb * TN + wn * 64 + nn * 32 + r, v);
  }
}

// Diagnostic build of scan_quad_kernel with in-kernel stamps (tools/prof_scan.py, MMG_SCAN_KERNEL=timed).
// Stamps go to a buffer of their own (dbg); the q outputs are still produced but the run time of this
// build is not quoted anywhere.
__global__ __launch_bounds__(NTHREADS, 2) void scan_quad_timed_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
    int64_t digit_stride, const int* __restrict__ job_off, const int2* __restrict__ jobs, int AS,
    unsigned long long* __restrict__ q, unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = blockIdx.x;
  const int x = b & 7, i = b >> 3;
  const int cohort = i >> 5, within = i & 31;
  const int a = within % AS, grp = within / AS;
  const int sb = (cohort * 8 + x) * AS + a;
  if (sb >= nSb) return;
  const int j0 = job_off[grp], j1 = job_off[grp + 1];
  if (j1 <= j0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int8_t* Q = S + (int64_t)sb * TN * ldS;
  unsigned long long seg[5] = {0, 0, 0, 0, 0};
  unsigned long long tepi = 0, ksteps = 0;
  const unsigned long long tstart = stamp();
  for (int jj = j0; jj < j1; ++jj) {
    const int2 jb = jobs[jj];
    const int8_t* P = Bq + (int64_t)jb.x * digit_stride + (int64_t)jb.y * TM * ldB;
    v16i acc[4][2];
    gemm_tile_i8_timed(P, ldB, Q, ldS, 0, 2 * (jb.y + 1), lds, acc, seg);
    ksteps += 2 * (jb.y + 1);
    const unsigned long long te = stamp();
    long long part = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) part += acc[m][n][e];
    if (part == 0x7fffffffffffll) q[0] = 1;      // keep the accumulators alive
    __syncthreads();
    tepi += stamp() - te;
  }
  const unsigned long long ttot = stamp() - tstart;
  if (lane == 0 && b < 2048) {
    unsigned long long* o = dbg + ((size_t)b * 8 + wave) * 8;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3]; o[4] = seg[4]; o[5] = tepi; o[6] = ttot; o[7] = ksteps;
  }
}

// Flattened-pipeline flavour: the workgroup's jobs form one K-step stream (gemm_i8_ring.h
// run_tiles_flat2); the epilogue operands (genotype bytes of the job's diagonal block) are captured
// from the Q tile in LDS while the matching K step is resident, so the next job's first stage can be
// in flight during the epilogue.
__global__ __launch_bounds__(NTHREADS, 2) void scan_quad_flat_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
  const int8_t* Q = S + (int64_t)sb * TN * ldS;
  unsigned long long qacc[2] = {0ull, 0ull};
  int2* jl = (int2*)(lds + LDS_BYTES);
  for (int t = threadIdx.x; t < j1 - j0; t += NTHREADS) jl[t] = jobs[j0 + t];
  __syncthreads();
  int sv[2][16];                       // this lane's 64 genotype bytes per SNP column for the epilogue
  auto tile = [&](int t) {
    const int2 jb = jl[t];
    TileDesc d;
    d.P = Bq + (int64_t)jb.x * digit_stride + (int64_t)jb.y * TM * ldB;
    d.Q = Q;
    d.nks = 2 * (jb.y + 1);
    return d;
  };
  auto hook = [&](int t, int ks, int nks, const char* stage) {
    if (ks != nks - 2 + wm) return;    // K step 2J + wm holds columns 256J + wm*128 .. +127
    const char* qt = stage + TILE_BYTES;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int qrow = wn * 64 + nn * 32 + r;
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        const v4i ch = lds_frag(qt, qrow, c8);
        sv[nn][c8 * 2 + 0] = h ? ch[1] : ch[0];
        sv[nn][c8 * 2 + 1] = h ? ch[3] : ch[2];
      }
    }
  };
  auto epi = [&](int t, v16i (&acc)[4][2]) {
    const int d = jl[t].x;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      long long part = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int wd = sv[nn][(2 * m + (g4 >> 1)) * 2 + (g4 & 1)];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            part += (long long)acc[m][nn][g4 * 4 + e] * (long long)(int)(int8_t)((wd >> (8 * e)) & 0xff);
        }
      qacc[nn] += ((unsigned long long)part) << (8 * d);
    }
  };
  run_tiles_flat2(j1 - j0, ldB, ldS, lds, tile, hook, epi);
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    unsigned long long v = qacc[nn];
    v += __shfl_xor(v, 32);
    if (h == 0) atomicAdd(q + (int64_t)sb * TN + wn * 64 + nn * 32 + r, v);
  }
}

// 16x16x64-MFMA flavour of scan_quad_kernel (same tiles, same exact integers).
__global__ __launch_bounds__(NTHREADS, 2) void scan_quad16_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, g = lane >> 4, r = lane & 15;
  const int8_t* Q = S + (int64_t)sb * TN * ldS;
  unsigned long long qacc[4] = {0ull, 0ull, 0ull, 0ull};
  for (int jj = j0; jj < j1; ++jj) {
    const int2 jb = jobs[jj];
    const int d = jb.x, J = jb.y;
    const int8_t* P = Bq + (int64_t)d * digit_stride + (int64_t)J * TM * ldB;
    v4i acc[8][4];
    gemm_tile_i8_16(P, ldB, Q, ldS, 0, 2 * (J + 1), lds, acc);
    // lane holds SNP column wn*64 + nn*16 + r and rows j = wm*128 + m*16 + 4*g + reg
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int8_t* srow = Q + (int64_t)(wn * 64 + nn * 16 + r) * ldS + J * TM + wm * 128 + 4 * g;
      long long part = 0;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int wd = *(const int*)(srow + m * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          part += (long long)acc[m][nn][e] * (long long)(int)(int8_t)((wd >> (8 * e)) & 0xff);
      }
      qacc[nn] += ((unsigned long long)part) << (8 * d);
    }
  }
#pragma unroll
  for (int nn = 0; nn < 4; ++nn) {
    unsigned long long v = qacc[nn];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (g == 0) atomicAdd(q + (int64_t)sb * TN + wn * 64 + nn * 16 + r, v);
  }
}

// Second-generation mainloop: the workgroup's jobs form one flattened K-step pipeline over a
// 4-slot LDS ring (gemm_i8_ring.h).  Same arithmetic, same exact integer results.
template <int PINGPONG>
__global__ __launch_bounds__(NTHREADS, 2) void scan_quad_ring_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int nSb, const int8_t* __restrict__ Bq, int64_t ldB,
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, r = lane & 31;
  const int8_t* Q = S + (int64_t)sb * TN * ldS;
  unsigned long long qacc[2] = {0ull, 0ull};
  // job descriptors -> LDS (behind the ring) so that the pipeline never waits on a VMEM load for them
  int2* jl = (int2*)(lds + LDS_BYTES);
  for (int t = threadIdx.x; t < j1 - j0; t += NTHREADS) jl[t] = jobs[j0 + t];
  __syncthreads();
  auto tile = [&](int t) {
    const int2 jb = jl[t];
    TileDesc d;
    d.P = Bq + (int64_t)jb.x * digit_stride + (int64_t)jb.y * TM * ldB;
    d.Q = Q;
    d.nks = 4 * (jb.y + 1);
    return d;
  };
  auto epi = [&](int t, v16i (&acc)[4][2]) {
    const int2 jb = jl[t];
    const int d = jb.x, J = jb.y;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int8_t* srow = Q + (int64_t)(wn * 64 + nn * 32 + r) * ldS + J * TM + wm * 128 + 4 * h;
      long long part = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int wd = *(const int*)(srow + m * 32 + 8 * g4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            part += (long long)acc[m][nn][g4 * 4 + e] * (long long)(int)(int8_t)((wd >> (8 * e)) & 0xff);
        }
      qacc[nn] += ((unsigned long long)part) << (8 * d);
    }
  };
  if (PINGPONG == 1) run_tiles_pingpong(j1 - j0, ldB, ldS, lds, tile, epi);
  else run_tiles_ring(j1 - j0, ldB, ldS, lds, tile, epi);
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    unsigned long long v = qacc[nn];
    v += __shfl_xor(v, 32);
    if (h == 0) atomicAdd(q + (int64_t)sb * TN + wn * 64 + nn * 32 + r, v);
  }
}


void launch_scan_quad(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, unsigned long long* q) {
  const int nSb = (int)(g->Mpad / TN);
  const int per = 8 * md.AS;
  const int ncoh = (nSb + per - 1) / per;
  int ablate = 5;   // 5 = production (loader waves); 0 = symmetric staging; 1-3 timing ablations
  if (const char* e = std::getenv("MMG_ABLATE")) ablate = std::atoi(e);
#define MMG_LAUNCH_QUAD(AB)                                                                                       \
  do {                                                                                                            \
    hipFuncSetAttribute((const void*)scan_quad_kernel<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL(scan_quad_kernel<AB>, dim3((unsigned)(ncoh * 256)), dim3(NTHREADS), LDS_BYTES, ctx->stream, \
                       g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,         \
                       md.job_off, md.jobs, md.AS, q);                                                            \
  } while (0)
  const char* kv = std::getenv("MMG_SCAN_KERNEL");
  if (ablate == 5 && kv && std::string(kv) == "timed") {
    static unsigned long long* dbg = nullptr;
    if (!dbg) hipMalloc(&dbg, (size_t)2048 * 8 * 8 * sizeof(unsigned long long));
    hipMemsetAsync(dbg, 0, (size_t)2048 * 8 * 8 * sizeof(unsigned long long), ctx->stream);
    hipFuncSetAttribute((const void*)scan_quad_timed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL(scan_quad_timed_kernel, dim3((unsigned)(ncoh * 256)), dim3(NTHREADS), LDS_BYTES, ctx->stream,
                       g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,
                       md.job_off, md.jobs, md.AS, q, dbg);
    std::vector<unsigned long long> h((size_t)2048 * 64);
    hipMemcpyAsync(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    for (int role = 0; role < 2; ++role) {           // loader waves (0-3) and their SIMD partners (4-7)
      double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      long cnt = 0;
      for (size_t w = 0; w < (size_t)2048 * 8; ++w) {
        if (h[w * 8 + 7] == 0 || (int)((w & 7) >> 2) != role) continue;
        for (int k = 0; k < 8; ++k) s[k] += (double)h[w * 8 + k];
        ++cnt;
      }
      if (cnt) {
        const double ks = s[7] / cnt;
        fprintf(stderr, "[timed] %s waves %ld  K-steps/wave %.0f  per K-step cycles: issueDMA %.0f  lds+mfma %.0f  vmcnt %.0f  barrier %.0f | "
                        "per wave: prologues %.0f  epilogues %.0f  total %.0f cycles\n", role ? "partner" : "loader ",
                cnt, ks, s[0] / s[7], s[1] / s[7], s[2] / s[7], s[3] / s[7], s[4] / cnt, s[5] / cnt, s[6] / cnt);
      }
    }
    return;
  }
  if (ablate == 5 && kv && std::string(kv) == "m16") {
    hipFuncSetAttribute((const void*)scan_quad16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL(scan_quad16_kernel, dim3((unsigned)(ncoh * 256)), dim3(NTHREADS), LDS_BYTES, ctx->stream,
                       g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,
                       md.job_off, md.jobs, md.AS, q);
    return;
  }
  if (ablate == 5 && kv && std::string(kv) == "flat") {
    const int lds_bytes = LDS_BYTES + 8 * std::max(1, md.njobs);
    hipFuncSetAttribute((const void*)scan_quad_flat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(scan_quad_flat_kernel, dim3((unsigned)(ncoh * 256)), dim3(NTHREADS), lds_bytes, ctx->stream,
                       g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,
                       md.job_off, md.jobs, md.AS, q);
    return;
  }
  if (ablate == 5 && kv && (std::string(kv) == "ring" || std::string(kv) == "pp")) {
    const int lds_bytes = LDS_BYTES + 8 * std::max(1, md.njobs);
    if (std::string(kv) == "ring") {
      hipFuncSetAttribute((const void*)scan_quad_ring_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      hipLaunchKernelGGL(scan_quad_ring_kernel<0>, dim3((unsigned)(ncoh * 256)), dim3(NTHREADS), lds_bytes, ctx->stream,
                         g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,
                         md.job_off, md.jobs, md.AS, q);
    } else {
      hipFuncSetAttribute((const void*)scan_quad_ring_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      hipLaunchKernelGGL(scan_quad_ring_kernel<1>, dim3((unsigned)(ncoh * 256)), dim3(NTHREADS), lds_bytes, ctx->stream,
                         g->d, (int64_t)g->Npad, nSb, md.Bq, (int64_t)md.Npad, (int64_t)md.Npad * md.Npad,
                         md.job_off, md.jobs, md.AS, q);
    }
    return;
  }
  switch (ablate) {
    case 1: MMG_LAUNCH_QUAD(1); break;
    case 2: MMG_LAUNCH_QUAD(2); break;
    case 3: MMG_LAUNCH_QUAD(3); break;
    case 0: MMG_LAUNCH_QUAD(0); break;
    default: MMG_LAUNCH_QUAD(5); break;
  }
#undef MMG_LAUNCH_QUAD
}

// ------------------------------------------------------------------ p-value
// Upper tail of F(1, nu) = I_x(nu/2, 1/2), x = nu/(nu+F)  (scipy.stats.f.sf, :1349).
// Continued fraction (modified Lentz); the tail 1-x = F/(nu+F) is formed directly.
__device__ double betacf(double a, double b, double x) {
  const double EPS = 1e-16, FPMIN = 1e-300;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < FPMIN) d = FPMIN;
  d = 1.0 / d;
  double hh = d;
  for (int m = 1; m <= 2000; ++m) {
    const double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa / c; if (fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d; hh *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa / c; if (fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d;
    const double del = d * c;
    hh *= del;
    if (fabs(del - 1.0) < EPS) break;
  }
  return hh;
}

__device__ double f_sf_1(double F, double nu, double lnbeta) {
  if (!(F > 0.0)) return (F != F) ? F : 1.0;
  if (isinf(F)) return 0.0;
  const double a = 0.5 * nu, b = 0.5;
  const double y = F / (nu + F), x = nu / (nu + F);
  const double bt = exp(a * log1p(-y) + b * log(y) - lnbeta);
  if (x < (a + 1.0) / (a + b + 2.0)) return bt * betacf(a, b, x) / a;
  return 1.0 - bt * betacf(b, a, y) / b;
}

__global__ void f_sf_kernel(const double* __restrict__ F, int64_t n, double nu, double lnbeta, double* __restrict__ p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n) p[gid] = f_sf_1(F[gid], nu, lnbeta);
}

void launch_f_sf(mmg_ctx* ctx, const double* F, int64_t n, int32_t df2, double lnbeta, double* p) {
  hipLaunchKernelGGL(f_sf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, F, n, (double)df2,
                     lnbeta, p);
}

// ------------------------------------------------------------------ finalize
// HBM-bound.  A block of 4 waves handles 32 SNP rows (8 per wave).  Per 1024-column chunk the block
// stages w and diag(A) once into LDS (lane-interleaved 16-byte units: conflict-free ds_read_b128)
// and every lane streams its 16 genotype bytes of each of its wave's 8 rows with one 16-byte load.
constexpr int FR = 8;                      // SNP rows per wave
constexpr int FIN_ROWS = 4 * FR;           // per block
__global__ __launch_bounds__(256, 2) void scan_finalize_kernel(
    const int8_t* __restrict__ S, int64_t ldS, int64_t M, int32_t Npad, const double* __restrict__ w,
    const double* __restrict__ diag, const unsigned long long* __restrict__ q, double step, double h0_rss, double nu,
    double lnbeta, double* __restrict__ rss, double* __restrict__ Fst, double* __restrict__ pv,
    double* __restrict__ dotv, double* __restrict__ denv, double* __restrict__ sumv) {
  __shared__ double2 lw[8 * 64], ldg[8 * 64];          // [unit e>>1][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t m0 = (int64_t)blockIdx.x * FIN_ROWS + (tid >> 6) * FR;
  double dw[FR], dd[FR];
  int sm[FR];
#pragma unroll
  for (int rr = 0; rr < FR; ++rr) { dw[rr] = 0.0; dd[rr] = 0.0; sm[rr] = 0; }
  const int nchunks = Npad >> 4;
  for (int c0 = 0; c0 < nchunks; c0 += 64) {
    __syncthreads();
    {
      const int k = c0 * 16 + tid * 4;                  // 4 consecutive columns per thread
      double2 w0 = make_double2(0, 0), w1 = w0, d0 = w0, d1 = w0;
      if (k < Npad) {
        w0 = *(const double2*)(w + k); w1 = *(const double2*)(w + k + 2);
        d0 = *(const double2*)(diag + k); d1 = *(const double2*)(diag + k + 2);
      }
      const int l = tid >> 2, u = (tid & 3) * 2;        // owning lane, first 16-byte unit
      lw[u * 64 + l] = w0; lw[(u + 1) * 64 + l] = w1;
      ldg[u * 64 + l] = d0; ldg[(u + 1) * 64 + l] = d1;
    }
    __syncthreads();
    const int c = c0 + lane;
    if (c < nchunks) {
      double wv[16], dv[16];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double2 t = lw[u * 64 + lane], x = ldg[u * 64 + lane];
        wv[2 * u] = t.x; wv[2 * u + 1] = t.y; dv[2 * u] = x.x; dv[2 * u + 1] = x.y;
      }
#pragma unroll
      for (int rr = 0; rr < FR; ++rr) {
        // rows beyond M are inside the padded store (Mpad is a multiple of 256) and hold zeros
        const uint4 v = *(const uint4*)(S + (m0 + rr) * ldS + c * 16);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int xi = (int)(int8_t)((wds[e >> 2] >> (8 * (e & 3))) & 0xff);
          const double xd = (double)xi;
          dw[rr] = fma(xd, wv[e], dw[rr]);
          dd[rr] = fma(xd * xd, dv[e], dd[rr]);
          sm[rr] += xi;
        }
      }
    }
  }
  double my_dw = 0.0, my_dd = 0.0;
  int my_sm = 0;
#pragma unroll
  for (int rr = 0; rr < FR; ++rr) {
    double a = dw[rr], b = dd[rr];
    int s = sm[rr];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
      s += __shfl_xor(s, o);
    }
    if (lane == rr) { my_dw = a; my_dd = b; my_sm = s; }
  }
  const int64_t m = m0 + lane;
  if (lane < FR && m < M) {
    const double qd = (double)(long long)q[m];
    const double den = fma(step, qd, my_dd);
    const double num = my_dw * my_dw;
    double r = h0_rss;
    // den ~ 0: monomorphic after projection; the reference's lstsq returns no residual and rss
    // stays h0_rss (linear_models.py:1308,1329)
    if (den > 1e-7 * my_dd && den > 0.0) r = h0_rss - num / den;
    const double ratio = h0_rss / r;
    const double F = (ratio - 1.0) * nu;
    if (rss) rss[m] = r;
    if (Fst) Fst[m] = F;
    if (dotv) dotv[m] = my_dw;
    if (denv) denv[m] = den;
    if (sumv) sumv[m] = (double)my_sm;
  }
}

// out[m] = s_m . v for an arbitrary fp64 vector v (zero padded to Npad); same streaming shape.
__global__ __launch_bounds__(256, 2) void snp_dot_kernel(const int8_t* __restrict__ S, int64_t ldS, int64_t M,
                                                         int32_t Npad, const double* __restrict__ v,
                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  constexpr int DR = 4;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * DR;
  if (m0 >= M) return;
  double dw[DR];
#pragma unroll
  for (int rr = 0; rr < DR; ++rr) dw[rr] = 0.0;
  for (int c = lane; c < (Npad >> 4); c += 64) {
    double wv[16];
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const double2 t = *(const double2*)(v + c * 16 + e);
      wv[e] = t.x; wv[e + 1] = t.y;
    }
#pragma unroll
    for (int rr = 0; rr < DR; ++rr) {
      const uint4 u = *(const uint4*)(S + (m0 + rr) * ldS + c * 16);
      const uint32_t wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 16; ++e)
        dw[rr] = fma((double)(int)(int8_t)((wds[e >> 2] >> (8 * (e & 3))) & 0xff), wv[e], dw[rr]);
    }
  }
  double mine = 0.0;
#pragma unroll
  for (int rr = 0; rr < DR; ++rr) {
    double a = dw[rr];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == rr) mine = a;
  }
  if (lane < DR && m0 + lane < M) out[m0 + lane] = mine;
}

void launch_snp_dot(mmg_ctx* ctx, const mmg_geno* g, const double* v, double* out) {
  const int64_t nwaves = (g->M + 3) / 4;
  hipLaunchKernelGGL(snp_dot_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, ctx->stream, g->d,
                     (int64_t)g->Npad, g->M, g->Npad, v, out);
}

void launch_scan_finalize(mmg_ctx* ctx, const mmg_geno* g, const mmg_scan_model& md, mmg_scan_result& res,
                          double h0_rss, int32_t df2, double lnbeta) {
  hipLaunchKernelGGL(scan_finalize_kernel, dim3((unsigned)(g->Mpad / FIN_ROWS)), dim3(256), 0, ctx->stream, g->d,
                     (int64_t)g->Npad, g->M, g->Npad, md.w, md.diag, res.q, md.step, h0_rss, (double)df2, lnbeta,
                     res.rss, res.F, res.p, res.dot, res.den, res.sum);
  // p-values in their own launch: one lane per SNP (in the finalize kernel only 8 of 64 lanes hold a
  // finished SNP, and the continued fraction is ~100 dependent fp64 divisions long)
  if (res.p && g->M > 0) launch_f_sf(ctx, res.F, g->M, df2, lnbeta, res.p);
}

}  // namespace mmg
