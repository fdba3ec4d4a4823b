// build: hipcc -O3 --offload-arch=gfx950 tools/probe/ds_read_tr_b4.hip -o tools/probe/ds_read_tr_b4.bin
// Probe: semantics of ds_read_b64_tr_b4 on gfx950 (4-bit transposed LDS read; this image's guides document the 16-bit
// form only).  LDS is filled so that every NIBBLE is identifiable: byte at offset o holds (lo = o & 15 ... ) -- two
// experiments: (a) nibble value = row index (which rows land in which output nibble), (b) nibble value = column index
// within the 16-nibble row segment (which column a lane receives).  Rows are 8 bytes (16 nibbles) apart by `stride`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(const uint8_t* __restrict__ img, const int* __restrict__ addr, uint32_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = img[i];
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
  v2i r;
  asm volatile("ds_read_b64_tr_b4 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 2] = r.x; out[threadIdx.x * 2 + 1] = r.y;
}
int main() {
  static uint8_t img[8192]; int h[64]; uint32_t o[128];
  uint8_t* dimg; int* d; uint32_t* dout;
  (void)hipMalloc(&dimg, sizeof(img)); (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
  const int stride = 128;                      // bytes between consecutive k rows
  for (int exp = 0; exp < 2; ++exp) {
    // group g of 16 lanes reads the 16 rows x 16 nibble-columns block at rows 16g.., byte columns 0..7
    for (int r = 0; r < 64; ++r)
      for (int b = 0; b < 128; ++b) {
        const int c0 = 2 * b, c1 = 2 * b + 1;   // nibble columns of this byte: low nibble = column c0 (hypothesis)
        const int v0 = exp == 0 ? (r & 15) : (c0 & 15), v1 = exp == 0 ? (r & 15) : (c1 & 15);
        img[r * stride + b] = (uint8_t)(v0 | (v1 << 4));
      }
    for (int l = 0; l < 64; ++l) h[l] = (16 * (l / 16) + (l % 16)) * stride;   // lane q of a group -> row q, bytes 0..7
    (void)hipMemcpy(dimg, img, sizeof(img), hipMemcpyHostToDevice);
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dimg, d, dout);
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("exp %d (%s): per lane the 16 nibbles received, lowest first\n", exp, exp == 0 ? "nibble = source row & 15" : "nibble = source nibble-column & 15");
    for (int l = 0; l < 64; ++l) {
      printf(" lane %2d:", l);
      for (int n = 0; n < 16; ++n) printf(" %2u", (o[l * 2 + n / 8] >> (4 * (n % 8))) & 15);
      printf("\n");
    }
  }
  return 0;
}
