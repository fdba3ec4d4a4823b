// build: hipcc -O3 --offload-arch=gfx950 tools/probe/fp64_rate.hip -o tools/probe/fp64_rate.bin -- sustained fp64 rate of the matrix pipe, the VALU, and both in one wave
// measured (MI355X, round 2): MFMA 35.5 / 47.8 TF at 1 / 2 waves per SIMD, VALU 61.1 / 69.2 TF, both 49.4 / 59.2 TF (shared DP units)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double seed) {
  v4d acc[12];
  double f[24];
  for (int i = 0; i < 12; ++i) acc[i] = v4d{0, 0, 0, 0};
  for (int i = 0; i < 24; ++i) f[i] = seed * i;
  double a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 24; ++i) f[i] = fma(a, f[i], b);
    }
  }
  double s = 0;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 24; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> double run(int waves_per_simd, const char* name) {
  double* d; hipMalloc(&d, 256 * 1024 * 8 * sizeof(double));
  const int iters = 20000, blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1e-9);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  const double mf = (MODE != 1) ? waves * iters * 12 * 2048.0 : 0.0;     // 16*16*4*2 flops
  const double vf = (MODE != 0) ? waves * iters * 8 * 24 * 64 * 2.0 : 0.0;
  printf("%-28s waves/SIMD %d: %.2f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", name, waves_per_simd, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  hipFree(d);
  return ms;
}
int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>(w, "mfma_f64_16x16x4 only");
    run<1>(w, "v_fma_f64 only");
    run<2>(w, "both in one wave");
  }
  return 0;
}
