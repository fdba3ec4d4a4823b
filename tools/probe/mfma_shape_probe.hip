// MFMA-shape probe (GPU box): bare int8 MFMA loops on random register operands, one wave per SIMD, the same
// 128x128 output tile per wave -- v_mfma_i32_32x32x32_i8 (16 tiles) vs v_mfma_i32_16x16x64_i8 (64 tiles).
// Prints wall time, TOP/s and the clock held (s_memtime / s_memrealtime), MI355X_MICROARCH.md 'DVFS give-back' (7).
// build: hipcc -O3 --offload-arch=gfx950 mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ inline uint32_t hsh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(int iters, int zero, int* out, unsigned long long* clk) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  v4i a[8], b[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = zero ? 0 : (int)hsh(t * 64 + i * 4 + j);
      b[i][j] = zero ? 0 : (int)(hsh(t * 64 + 32 + i * 4 + j) & 0x01010101u);     // genotype-like 0/1 bytes
    }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  if (SHAPE == 32) {
    v16i acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) acc[m][n][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m + 4 * kk], b[n + 4 * kk], acc[m][n], 0, 0, 0);
    }
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) s += acc[m][n][e];
  } else {
    v4i acc[8][8];
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 8; ++n) for (int e = 0; e < 4; ++e) acc[m][n][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 8; ++n) for (int e = 0; e < 4; ++e) s += acc[m][n][e];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[t] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const char* name, int zero) {
  const int nb = 256, iters = 40000;                 // per iteration and wave: 2 x 128x128x32 MACs = 1048576 MACs
  int* out; unsigned long long* clk;
  hipMalloc(&out, nb * 256 * sizeof(int)); hipMalloc(&clk, nb * 2 * sizeof(unsigned long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 40; ++rep) {               // ~2 s of back-to-back launches; keep the last timing
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<SHAPE>, dim3(nb), dim3(256), 0, 0, iters, zero, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(nb * 2);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double ghz = 0; for (int i = 0; i < nb; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; ghz /= nb;
  const double ops = 2.0 * 1048576.0 * iters * 4.0 * nb;   // 2 ops per MAC, 4 waves per block
  printf("%-22s %s  %.2f ms  %.0f TOP/s  clock %.3f GHz  cycles/iter %.0f\n", name, zero ? "zeros " : "random", ms,
         ops / (ms * 1e-3) / 1e12, ghz, (double)h[0] / iters);
  hipFree(out); hipFree(clk);
}

int main() {
  run<32>("i32_32x32x32_i8", 0);
  run<16>("i32_16x16x64_i8", 0);
  run<32>("i32_32x32x32_i8", 1);
  run<16>("i32_16x16x64_i8", 1);
  run<32>("i32_32x32x32_i8", 0);
  run<16>("i32_16x16x64_i8", 0);
  return 0;
}
