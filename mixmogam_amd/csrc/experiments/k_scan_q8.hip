// k_scan_q8.hip -- the 8-wave generations of the EMMAX quadratic-form GEMM (gemm_i8_core.h mainloop): the
// loader-wave kernel that was the production kernel before k_scan_w4s.hip (MMG_SCAN_KERNEL=q8; MMG_ABLATE
// instruments it), its in-kernel-stamp twin (timed) and the structures that were tried on the way (flat, m16, ring,
// pp).  All of them accumulate the same exact integers as the production kernel
// (tests/test_gpu_parity.py::test_scan_kernel_generations_agree_bit_for_bit); they stay selectable for A/B runs.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "gemm_i8_core.h"
#include "gemm_i8_ring.h"
#include "mmg_internal.h"

namespace mmg {

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
      qacc[nn] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
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
      qacc[nn] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
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
      qacc[nn] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
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
      qacc[nn] += ((unsigned long long)part) << (SCAN_DIGIT_BITS * d);
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

}  // namespace mmg
