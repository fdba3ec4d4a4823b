// build: hipcc -O3 --offload-arch=gfx950 tools/probe/fp64_valu_gemm.hip -o tools/probe/fp64_valu_gemm.bin
// The inner loop of a register-tiled fp64 GEMM on the VALU (v_fma_f64, operands from LDS) against the 33-35 TFLOP/s the
// v_mfma_f64_16x16x4 kernels of dense64.hip reach: 256 threads = a 128 x 128 tile, 8 x 8 accumulators per thread, operand
// chunks [KC][128] resident in LDS (no global traffic: the ceiling of the loop itself).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int KC = 16;
template <int TR, int TC>
__global__ __launch_bounds__(256) void loop(double* out, int iters) {
  __shared__ double A[KC][128], B[KC][128];
  for (int i = threadIdx.x; i < KC * 128; i += 256) { (&A[0][0])[i] = 1e-3 * (i % 17); (&B[0][0])[i] = 1e-3 * (i % 13); }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[TR][TC];
  for (int r = 0; r < TR; ++r) for (int c = 0; c < TC; ++c) acc[r][c] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      double a[TR], b[TC];
#pragma unroll
      for (int j = 0; j < TR / 2; ++j) { const v2d v = *(const v2d*)&A[k][(j * 16 + ty) * 2]; a[2 * j] = v.x; a[2 * j + 1] = v.y; }
#pragma unroll
      for (int j = 0; j < TC / 2; ++j) { const v2d v = *(const v2d*)&B[k][(j * 16 + tx) * 2]; b[2 * j] = v.x; b[2 * j + 1] = v.y; }
#pragma unroll
      for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
    }
  }
  double s = 0;
  for (int r = 0; r < TR; ++r) for (int c = 0; c < TC; ++c) s += acc[r][c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int TR, int TC> void run(int wgs_per_cu) {
  double* d; hipMalloc(&d, 256 * 1024 * sizeof(double));
  const int iters = 4000, blocks = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop<TR, TC>), dim3(blocks), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((loop<TR, TC>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)blocks * 256 * iters * KC * TR * TC * 2.0;
  printf("thread tile %d x %d, %d workgroup(s) of 256 per CU: %.2f ms  %.1f TFLOP/s\n", TR, TC, wgs_per_cu, ms, fl / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 2; ++w) { run<8, 8>(w); run<8, 4>(w); run<4, 8>(w); run<8, 12>(w); }
  run<8, 8>(3);
  return 0;
}
