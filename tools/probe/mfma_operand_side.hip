// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_operand_side.hip -o tools/probe/mfma_operand_side.bin
// Does it matter for power (= clock, on this power-capped part) WHICH operand of v_mfma_i32_32x32x32_i8 carries the
// sparse 0/1 genotype bytes and which the dense digit bytes?  Sustained rate of a bare MFMA loop, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int mode>
__global__ __launch_bounds__(256) void k(int iters, int* out) {
  v4i dense[4], sparse[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const uint32_t h = hash(threadIdx.x * 64 + i * 4 + j + 1);
      dense[i][j] = (int)h;
      sparse[i][j] = (int)(hash(h) & 0x01010101u);
    }
  v16i acc[16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (mode == 0) acc[m * 4 + n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(dense[m], sparse[n], acc[m * 4 + n], 0, 0, 0);
        else if (mode == 1) acc[m * 4 + n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(sparse[m], dense[n], acc[m * 4 + n], 0, 0, 0);
        else if (mode == 2) acc[m * 4 + n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(dense[m], dense[n], acc[m * 4 + n], 0, 0, 0);
        else acc[m * 4 + n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(sparse[m], sparse[n], acc[m * 4 + n], 0, 0, 0);
      }
  }
  int s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  int* d; hipMalloc(&d, 1024 * 256 * 4);
  const char* names[4] = {"A dense digits, B 0/1 (the scan kernel)", "A 0/1, B dense digits", "both dense", "both 0/1"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      const int iters = 1500000, blocks = 256;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto go = [&](int n) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, n, d);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, n, d);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, n, d);
        else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, n, d);
      };
      go(2000);
      hipEventRecord(e0);
      go(iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double ops = (double)blocks * 4 * iters * 16 * 65536.0;
      printf("%-42s %.1f ms  %.2f POP/s  (=> %.2f GHz if the pipe never stalls)\n", names[mode], ms, ops / ms / 1e12, ops / ms / 1e12 / 5.0 * 2.4);
    }
  return 0;
}
