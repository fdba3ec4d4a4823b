// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMMG_D64_STAMPS -Imixmogam_amd/csrc -Iinclude tools/probe/head_probe.hip -o tools/probe/head_probe.bin
// Phase times inside the single-workgroup head kernels of dense64.hip and launch-to-launch times of the row kernels, on a
// random tall panel (n x 64).
#include "../../mixmogam_amd/csrc/dense64.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
using namespace mmg;
// worst relative error of one / two Newton steps on v_rcp_f64 against IEEE division
__global__ void rcp_err_kernel(double* out) {
  double e1 = 0, e2 = 0;
  unsigned long long x = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1);
  for (int it = 0; it < 20000; ++it) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const double p = ldexp(1.0 + (double)(x >> 12) * 0x1p-52, (int)(x & 63) - 32);
    const double q = 1.0 / p;
    e1 = fmax(e1, fabs(rcp1_f64(p) - q) / q);
    e2 = fmax(e2, fabs(rcp_f64(p) - q) / q);
  }
  out[2 * threadIdx.x] = e1; out[2 * threadIdx.x + 1] = e2;
}
int main(int argc, char** argv) {
  {
    double* d; hipMalloc(&d, 128 * 8);
    hipLaunchKernelGGL(rcp_err_kernel, dim3(1), dim3(64), 0, 0, d);
    double h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0; for (int i = 0; i < 64; ++i) { e1 = std::max(e1, h[2 * i]); e2 = std::max(e2, h[2 * i + 1]); }
    printf("v_rcp_f64 + 1 Newton step: max rel err %.3e (%.2f ulp);  + 2 steps: %.3e\n", e1, e1 / 1.11e-16, e2);
  }
  const int64_t n = argc > 1 ? atoll(argv[1]) : 5000;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> hP((size_t)n * 64);
  for (auto& v : hP) v = nd(rng);
  double *P, *V, *small, *part;
  PanelFlags* flags;
  hipMalloc(&P, n * 64 * 8); hipMalloc(&V, n * 64 * 8); hipMalloc(&small, 6 * 4096 * 8); hipMalloc(&part, 2 * D64_MAX_SLICES * 4096 * 8);
  hipMalloc(&flags, sizeof(PanelFlags)); hipMemset(flags, 0, sizeof(PanelFlags));
  hipMemcpy(P, hP.data(), n * 64 * 8, hipMemcpyHostToDevice);
  dense64_init();
  double *R1 = small, *R1inv = small + 4096, *Mk = small + 2 * 4096, *Cb = small + 3 * 4096, *Rt = small + 4 * 4096, *G1 = small + 5 * 4096;
  double* part2; hipMalloc(&part2, 128 * 4096 * 8);
  hipStream_t st = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t ev[10]; for (auto& e : ev) hipEventCreate(&e);
    hipEventRecord(ev[0], st);
    int G = launch_gram_slices(st, P, n, P, n, 64, n, part2, 128);
    hipEventRecord(ev[1], st);
    launch_gram_reduce(st, part2, G, G1, 64);
    hipEventRecord(ev[2], st);
    launch_cholqr_head1(st, G1, 1, R1, R1inv, flags);
    hipEventRecord(ev[3], st);
    launch_rows_gemm(st, P, n, V, n, n, R1inv);
    hipEventRecord(ev[4], st);
    G = launch_gram_slices(st, V, n, V, n, 64, n, part2, 128);
    launch_gram_reduce(st, part2, G, G1, 64);
    hipEventRecord(ev[5], st);
    launch_cholqr_head2(st, G1, 1, R1, V, n, Mk, Cb, Rt, 64, flags);
    hipEventRecord(ev[6], st);
    launch_rows_gemm(st, V + 64, n, V + 64, n, n - 64, Cb);
    hipEventRecord(ev[7], st);
    hipStreamSynchronize(st);
    const char* nm[7] = {"gram", "reduce", "head1", "rows_gemm", "gram+reduce", "head2", "rows_gemm"};
    printf("n=%lld G=%d:", (long long)n, G);
    for (int i = 0; i < 7; ++i) { float ms; hipEventElapsedTime(&ms, ev[i], ev[i + 1]); printf(" %s %.1f us", nm[i], ms * 1e3); }
    unsigned long long hs[64];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(d64_stamps), sizeof(hs));
    printf("\n  head1 phases (us): load %.1f  chol+inv %.1f  write %.1f\n", (hs[1] - hs[0]) / 100.0, (hs[2] - hs[1]) / 100.0, (hs[3] - hs[2]) / 100.0);
    printf("  chol64_lds inside head1 (us): panels + updates %.1f  inverse %.1f\n", (hs[20] - hs[1]) / 100.0, (hs[21] - hs[20]) / 100.0);
    printf("  head2 phases (us): load %.1f  chol+inv %.1f  products %.1f  gauss-jordan %.1f  write %.1f\n", (hs[8] - hs[7]) / 100.0,
           (hs[9] - hs[8]) / 100.0, (hs[10] - hs[9]) / 100.0, (hs[11] - hs[10]) / 100.0, (hs[12] - hs[11]) / 100.0);
  }
  // orthogonality of H = I - V M V' on the panel: H'P must be [R; 0]; check |V'V - (M^-1 + M^-T)| via M (V'V) M' - (M + M')
  {
    std::vector<double> hV((size_t)n * 64), hM(4096), hR(4096);
    hipMemcpy(hV.data(), V, n * 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hM.data(), Mk, 4096 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hR.data(), Rt, 4096 * 8, hipMemcpyDeviceToHost);
    std::vector<double> vtv(4096, 0.0), t1(4096, 0.0);
    for (int a = 0; a < 64; ++a) for (int b2 = 0; b2 < 64; ++b2) { double s = 0; for (int64_t r = 0; r < n; ++r) s += hV[r + a * n] * hV[r + b2 * n]; vtv[a + 64 * b2] = s; }
    // orthogonality: M + M' = M' (V'V) M
    double err = 0;
    for (int a = 0; a < 64; ++a) for (int b2 = 0; b2 < 64; ++b2) { double s = 0; for (int k = 0; k < 64; ++k) s += vtv[a + 64 * k] * hM[k + 64 * b2]; t1[a + 64 * b2] = s; }
    for (int a = 0; a < 64; ++a) for (int b2 = 0; b2 < 64; ++b2) { double s = 0; for (int k = 0; k < 64; ++k) s += hM[k + 64 * a] * t1[k + 64 * b2]; err = std::max(err, std::abs(s - hM[a + 64 * b2] - hM[b2 + 64 * a])); }
    // H'P = P - V M'(V'P): rows >= 64 must vanish, the top block = R (upper)
    std::vector<double> vtp(4096, 0.0), c2(4096, 0.0);
    for (int a = 0; a < 64; ++a) for (int b2 = 0; b2 < 64; ++b2) { double s = 0; for (int64_t r = 0; r < n; ++r) s += hV[r + a * n] * hP[r + b2 * n]; vtp[a + 64 * b2] = s; }
    for (int a = 0; a < 64; ++a) for (int b2 = 0; b2 < 64; ++b2) { double s = 0; for (int k = 0; k < 64; ++k) s += hM[k + 64 * a] * vtp[k + 64 * b2]; c2[a + 64 * b2] = s; }
    double below = 0, top = 0;
    for (int64_t r = 0; r < n; ++r) for (int b2 = 0; b2 < 64; ++b2) {
      double s = hP[r + b2 * n];
      for (int k = 0; k < 64; ++k) s -= hV[r + k * n] * c2[k + 64 * b2];
      if (r >= 64 || r > b2) below = std::max(below, std::abs(s));
      else top = std::max(top, std::abs(s - hR[r + 64 * b2]));
    }
    printf("checks: |M + M' - M'(V'V)M| %.2e   |H'P below the triangle| %.2e   |top - R| %.2e\n", err, below, top);
  }
  {
    // head1 on its own: R1'R1 = P'P and R1inv R1 = I
    int G = launch_gram_slices(st, P, n, P, n, 64, n, part2, 128);
    launch_gram_reduce(st, part2, G, G1, 64);
    launch_cholqr_head1(st, G1, 1, R1, R1inv, flags);
    hipStreamSynchronize(st);
    std::vector<double> g(4096), r1(4096), ri(4096);
    hipMemcpy(g.data(), G1, 4096 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), R1, 4096 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ri.data(), R1inv, 4096 * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, gmax = 0, low = 0;
    for (int a = 0; a < 64; ++a) for (int b2 = 0; b2 < 64; ++b2) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k < 64; ++k) { s1 += r1[k + 64 * a] * r1[k + 64 * b2]; s2 += ri[a + 64 * k] * r1[k + 64 * b2]; }
      e1 = std::max(e1, std::abs(s1 - g[a + 64 * b2])); gmax = std::max(gmax, std::abs(g[a + 64 * b2]));
      e2 = std::max(e2, std::abs(s2 - (a == b2 ? 1.0 : 0.0)));
      if (a > b2) low = std::max(low, std::max(std::abs(r1[a + 64 * b2]), std::abs(ri[a + 64 * b2])));
    }
    printf("head1: |R1'R1 - G| / |G| %.2e   |R1inv R1 - I| %.2e   below the diagonal %.2e\n", e1 / gmax, e2, low);
  }
  PanelFlags hf; hipMemcpy(&hf, flags, sizeof(hf), hipMemcpyDeviceToHost);
  printf("flags: bad %d panels %d series %d tiny %d\n", hf.bad, hf.panels, hf.series, hf.tiny);
  return 0;
}
