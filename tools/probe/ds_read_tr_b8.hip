#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// Probe: semantics of ds_read_b64_tr_b8 on gfx950.  LDS holds byte value = its own offset (mod 256) in a 256-byte
// region; lane l supplies address A(l); we dump the 8 bytes each lane receives.
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(const int* __restrict__ addr, uint32_t* __restrict__ out) {
  __shared__ uint8_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint8_t)(i & 255);
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
  v2i r;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 2] = r.x; out[threadIdx.x * 2 + 1] = r.y;
}
int main() {
  int h[64]; uint32_t o[128]; int* d; uint32_t* dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  for (int exp = 0; exp < 3; ++exp) {
    for (int l = 0; l < 64; ++l) {
      int grp = l / 16, k = l % 16;
      if (exp == 0) h[l] = grp * 256 + k * 8;                      // 16 lanes x 8 B contiguous = 8 rows of 16 B
      if (exp == 1) h[l] = grp * 1024 + (k >> 1) * 64 + (k & 1) * 8; // rows 64 B apart, two 8-B halves per row
      if (exp == 2) h[l] = grp * 1024 + (k & 7) * 64 + (k >> 3) * 8; // alternative lane->(row, half) map
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("exp %d\n", exp);
    for (int l = 0; l < 64; ++l) {
      printf(" lane %2d addr %4d :", l, h[l]);
      for (int b = 0; b < 8; ++b) printf(" %3u", (o[l * 2 + b / 4] >> (8 * (b % 4))) & 255);
      printf("\n");
    }
  }
  return 0;
}
