// build: hipcc -O3 --offload-arch=gfx950 tools/probe/chain_latency.hip -o tools/probe/chain_latency.bin
// What a single-workgroup latency chain costs on this part: dependent v_fma_f64, fp64 division, sqrt, an LDS write ->
// barrier -> read round trip, and the shader clock such a launch runs at (s_memtime against s_memrealtime, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma(double* out, int n, double a, double b) {
  double x = threadIdx.x;
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) x = fma(x, a, b);
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[512] = (double)(t1 - t0); out[513] = (double)(w1 - w0); }
}
__global__ void k_div(double* out, int n, double a) {
  double x = 1.0 + threadIdx.x;
  for (int i = 0; i < n; ++i) x = a / x + 1.0;
  out[threadIdx.x] = x;
}
__global__ void k_sqrt(double* out, int n, double a) {
  double x = 1.0 + threadIdx.x;
  for (int i = 0; i < n; ++i) x = sqrt(x) + a;
  out[threadIdx.x] = x;
}
__global__ void k_lds(double* out, int n) {
  __shared__ double buf[2][64];
  double x = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    if ((threadIdx.x >> 6) == (i & 3)) buf[i & 1][threadIdx.x & 63] = x;
    __syncthreads();
    x = fma(buf[i & 1][(threadIdx.x + 1) & 63], 0.5, 1.0);
  }
  out[threadIdx.x] = x;
}
__global__ void k_lds1(double* out, int n) {       // one wave, no barrier needed across waves
  __shared__ double buf[2][64];
  double x = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    buf[i & 1][threadIdx.x] = x;
    __syncthreads();
    x = fma(buf[i & 1][(threadIdx.x + 1) & 63], 0.5, 1.0);
  }
  out[threadIdx.x] = x;
}
__global__ void k_readlane(double* out, int n) {
  double x = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), i & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), i & 63);
    x = fma(__hiloint2double(hi, lo), 0.5, x);
  }
  out[threadIdx.x] = x;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; hipMalloc(&d, 1 << 20);
  const int n = 200000;
  for (int blocks : {1, 256, 1024}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(64), 0, 0, d, n, 0.999, 0.5); });
    double h[2]; hipMemcpy(h, d + 512, 16, hipMemcpyDeviceToHost);
    printf("dependent v_fma_f64, %4d blocks of 1 wave: %.2f ns per fma; shader clock %.0f MHz (s_memtime / wall 100 MHz)\n", blocks, ms * 1e6 / n,
           h[0] / h[1] * 100.0);
  }
  printf("fp64 division chain (+1 add): %.1f ns per step\n", timeit([&] { hipLaunchKernelGGL(k_div, dim3(1), dim3(64), 0, 0, d, n, 3.0); }) * 1e6 / n);
  printf("fp64 sqrt chain (+1 add):     %.1f ns per step\n", timeit([&] { hipLaunchKernelGGL(k_sqrt, dim3(1), dim3(64), 0, 0, d, n, 3.0); }) * 1e6 / n);
  printf("LDS write -> barrier -> read, 4 waves: %.1f ns per step\n", timeit([&] { hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 0, 0, d, n); }) * 1e6 / n);
  printf("LDS write -> barrier -> read, 1 wave:  %.1f ns per step\n", timeit([&] { hipLaunchKernelGGL(k_lds1, dim3(1), dim3(64), 0, 0, d, n); }) * 1e6 / n);
  printf("v_readlane pair + fma chain:           %.1f ns per step\n", timeit([&] { hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, d, n); }) * 1e6 / n);
  // many short launches back to back: what a dependent launch costs
  for (int len : {10, 1000}) {
    const int reps = 200;
    float ms = timeit([&] { for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, d, len, 0.999, 0.5); });
    printf("%d back-to-back launches of a %d-fma single-wave kernel: %.2f us per launch\n", reps, len, ms * 1e3 / reps);
  }
  return 0;
}
