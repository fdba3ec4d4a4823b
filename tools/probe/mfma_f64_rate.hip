// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: one wave per SIMD and two, 4 independent accumulators (the shape of
// sym_skinny_kernel's inner loop) and 8.  hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC, int LB = 512>
__global__ __launch_bounds__(LB) void k(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int LB = 512>
void run(int wgs, const char* what, int threads = 256) {
  double* out;
  hipMalloc(&out, (size_t)wgs * threads * 8);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LB>), dim3(wgs), dim3(threads), 0, 0, out, 100, 1.0, 2.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LB>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)iters * NACC;
  const double flops = mfma_per_wave * 2048.0 * (threads / 64) * wgs;
  printf("%-34s %d accumulators: %.3f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", what, NACC, ms, ms * 1e6 / mfma_per_wave, flops / ms / 1e9);
  hipFree(out);
}
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k8(int* out, int iters) {
  v16i acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const v4i a = v4i{(int)threadIdx.x, 1, 2, 3}, b = v4i{4, 5, 6, (int)threadIdx.x};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
void run8(int wgs) {
  int* out;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k8, dim3(wgs), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k8, dim3(wgs), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 4;
  printf("int8 32x32x32 for comparison, %d WGs: %.3f ms, %.1f ns per MFMA per wave, %.2f POP/s\n", wgs, ms, ms * 1e6 / n,
         n * 65536.0 * 4 * wgs / ms / 1e12);
  hipFree(out);
}

int main() {
  run<4>(256, "256 WGs (1 wave / SIMD)");
  run<8>(256, "256 WGs (1 wave / SIMD)");
  run<4>(512, "512 WGs (2 waves / SIMD)");
  run<8>(512, "512 WGs (2 waves / SIMD)");
  run<4>(1024, "1024 WGs (4 waves / SIMD)");
  run<8>(512, "512 WGs again (clock held?)");
  // is it the clock (power) or the instruction's cadence?  Two waves per SIMD on a few CUs only:
  run<4, 256>(512, "launch_bounds(256), 512 WGs");
  run<8, 256>(512, "launch_bounds(256), 512 WGs");
  run<8>(8, "8 WGs x 512 threads (8 CUs busy)", 512);
  run<8>(64, "64 WGs x 512 threads (64 CUs busy)", 512);
  run<8>(256, "256 WGs x 512 threads (all CUs)", 512);
  run8(256);
  run8(512);
  return 0;
}
