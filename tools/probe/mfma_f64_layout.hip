// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_f64_layout.hip -o tools/probe/mfma_f64_layout.bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D, unsigned* S) {
  const int l = threadIdx.x;
  // A[16 x 4] row-major, B[4 x 16] row-major
  const double a = A[(l % 16) * 4 + l / 16];
  const double b = B[(l / 16) * 16 + l % 16];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
  unsigned x = 100 + l, y = 200 + l;
  auto p = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  S[l] = p[0]; S[64 + l] = p[1];
  auto q = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  S[128 + l] = q[0]; S[192 + l] = q[1];
}
int main() {
  double hA[64], hB[64], hD[256]; unsigned hS[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = (i + 1) * (k == 0 ? 1 : k == 1 ? 100 : k == 2 ? 10000 : 1000000);
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (k == 0) ? (j + 1) : 0;  // D[i][j] = (i+1)*(j+1)
  double *dA, *dB, *dD; unsigned* dS;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dS, sizeof hS);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 5) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) printf(" D=%g (i=%d,j=%d)", hD[l*4+r], (int)(hD[l*4+r]) / ((l%16)+1) - 1, l % 16); printf("\n"); }
  printf("permlane32_swap x':"); for (int l = 0; l < 64; l += 8) printf(" %u", hS[l]); printf("\n                 y':"); for (int l = 0; l < 64; l += 8) printf(" %u", hS[64 + l]);
  printf("\npermlane16_swap x':"); for (int l = 0; l < 64; l += 8) printf(" %u", hS[128 + l]); printf("\n                 y':"); for (int l = 0; l < 64; l += 8) printf(" %u", hS[192 + l]); printf("\n");
  return 0;
}
