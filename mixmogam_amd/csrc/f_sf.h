// f_sf.h -- upper tail of the F(1, nu) distribution on the device (replaces scipy.stats.f.sf at
// linear_models.py:1349,1172); shared by the scan finalize kernels (k_scan.hip, k_rot.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mmg {

// ------------------------------------------------------------------ p-value
// Upper tail of F(1, nu) = I_x(nu/2, 1/2), x = nu/(nu+F)  (scipy.stats.f.sf, :1349).
// Continued fraction (modified Lentz); the tail 1-x = F/(nu+F) is formed directly.
// 1/y by v_rcp_f64 + two Newton steps (full double accuracy for the normal-range values of the continued
// fraction; ~5 instructions instead of the ~12 of the IEEE division sequence -- the p-value kernel is latency bound
// on its six divisions per iteration)
static __device__ __forceinline__ double frcp(double y) {
  double r = __builtin_amdgcn_rcp(y);
  r = fma(r, fma(-y, r, 1.0), r);
  r = fma(r, fma(-y, r, 1.0), r);
  return r;
}

static __device__ double betacf(double a, double b, double x) {
  const double EPS = 1e-16, FPMIN = 1e-300;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x * frcp(qap);
  if (fabs(d) < FPMIN) d = FPMIN;
  d = frcp(d);
  double hh = d;
  for (int m = 1; m <= 2000; ++m) {
    const double m2 = 2.0 * m;
    double aa = m * (b - m) * x * frcp((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa * frcp(c); if (fabs(c) < FPMIN) c = FPMIN;
    d = frcp(d); hh *= d * c;
    aa = -(a + m) * (qab + m) * x * frcp((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa * frcp(c); if (fabs(c) < FPMIN) c = FPMIN;
    d = frcp(d);
    const double del = d * c;
    hh *= del;
    if (fabs(del - 1.0) < EPS) break;
  }
  return hh;
}

static __device__ double f_sf_1(double F, double nu, double lnbeta) {
  if (!(F > 0.0)) return (F != F) ? F : 1.0;
  if (isinf(F)) return 0.0;
  const double a = 0.5 * nu, b = 0.5;
  const double y = F / (nu + F), x = nu / (nu + F);
  const double bt = exp(a * log1p(-y) + b * log(y) - lnbeta);
  if (x < (a + 1.0) / (a + b + 2.0)) return bt * betacf(a, b, x) / a;
  return 1.0 - bt * betacf(b, a, y) / b;
}

}  // namespace mmg
