// build: hipcc -O3 --offload-arch=gfx950 tools/probe/fp64_sgpr_fma_rate.hip -o tools/probe/fp64_sgpr_fma_rate.bin -- v_fma_f64 fed by scalar loads
// measured: 24 coefficients per step, 2 elements per lane: 59.9 TF; 48 per step: 49-52 TF (the SGPR file cannot double-buffer 96 registers)
#include <hip/hip_runtime.h>
#include <cstdio>
// scalar-fed fp64 FMA stream, NH halves of 24 wave-uniform coefficients per step, R elements per lane
template <int NH, int R, int BAR>
__global__ __launch_bounds__(256) void sfma(const double* __restrict__ coef, int steps, double* out) {
  double acc[R][NH * 24];
  double tau[R];
  for (int r = 0; r < R; ++r) { tau[r] = 1e-3 * (threadIdx.x + r); for (int c = 0; c < NH * 24; ++c) acc[r][c] = 0; }
  for (int i = 0; i < steps; ++i) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const double* c = coef + ((size_t)i * NH + h) * 24;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 24; ++k) acc[r][h * 24 + k] = fma(tau[r], c[k], acc[r][h * 24 + k]);
      if (BAR) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) tau[r] += 1e-9;
  }
  double s = 0;
  for (int r = 0; r < R; ++r) for (int c = 0; c < NH * 24; ++c) s += acc[r][c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NH, int R, int BAR> void run() {
  const int steps = 5000;
  double *coef, *out; hipMalloc(&coef, steps * NH * 24 * 8); hipMemset(coef, 0, steps * NH * 24 * 8); hipMalloc(&out, 256 * 4096 * 8);
  const int blocks = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((sfma<NH, R, BAR>), dim3(blocks), dim3(256), 0, 0, coef, 100, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((sfma<NH, R, BAR>), dim3(blocks), dim3(256), 0, 0, coef, steps, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)blocks * 256 * steps * NH * 24 * R * 2.0;
  printf("halves %d R %d barrier %d: %.2f ms  %.1f TF\n", NH, R, BAR, ms, fl / ms / 1e9);
  hipFree(coef); hipFree(out);
}
int main() {
  run<1, 2, 0>(); run<2, 2, 0>(); run<2, 2, 1>(); run<2, 1, 1>(); run<2, 1, 0>();
  return 0;
}
