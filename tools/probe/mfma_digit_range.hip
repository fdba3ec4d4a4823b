// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_digit_range.hip -o tools/probe/mfma_digit_range.bin
// sustained int8 MFMA rate under the power cap by the VALUE RANGE of the dense operand (digit width) and the density of the 0/1 operand
// Does it matter for power (= clock, on this power-capped part) WHICH operand of v_mfma_i32_32x32x32_i8 carries the
// sparse 0/1 genotype bytes and which the dense digit bytes?  Sustained rate of a bare MFMA loop, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int mode>
__global__ __launch_bounds__(256) void k(int iters, int* out, int lo, int span, int mul, int dens) {
  v4i dense[4], sparse[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const uint32_t h = hash(threadIdx.x * 64 + i * 4 + j + 1);
      uint32_t dv = 0, sv = 0;
      for (int b = 0; b < 4; ++b) {
        const uint32_t hb = hash(h + b * 977u);
        const int v = (lo + (int)(hb % (uint32_t)span)) * mul;
        dv |= ((uint32_t)(v & 0xff)) << (8 * b);
        sv |= ((hash(hb) % 100u) < (uint32_t)dens ? 1u : 0u) << (8 * b);
      }
      dense[i][j] = (int)dv;
      sparse[i][j] = (int)sv;
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
  struct Case { const char* name; int lo, span, mul, dens; } cases[] = {
      {"A uniform [-128,127], B 0/1 p=.5", -128, 256, 1, 50}, {"A uniform [-64,63]", -64, 128, 1, 50},
      {"A uniform [-32,31]", -32, 64, 1, 50},                 {"A uniform [-16,15]", -16, 32, 1, 50},
      {"A uniform [0,127]", 0, 128, 1, 50},                   {"A uniform [0,63]", 0, 64, 1, 50},
      {"A multiples of 16 in [-128,112]", -8, 16, 16, 50},     {"A uniform [-128,127], B p=.25", -128, 256, 1, 25},
      {"A uniform [-128,127], B p=.1", -128, 256, 1, 10},      {"A uniform [-64,63], B p=.25", -64, 128, 1, 25}};
  for (auto& c : cases) {
    const int iters = 1200000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, 2000, d, c.lo, c.span, c.mul, c.dens);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, iters, d, c.lo, c.span, c.mul, c.dens);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 16 * 65536.0;
    printf("%-36s %.1f ms  %.2f POP/s\n", c.name, ms, ops / ms / 1e12);
  }
  return 0;
}
