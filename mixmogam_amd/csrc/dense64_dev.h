// dense64_dev.h -- device-side building blocks shared by dense64.hip (panel heads of the band reduction, blocked Cholesky) and
// reml_band.hip (banded Cholesky of B + delta I in 64-column blocks): the blocked 64 x 64 Cholesky with its inverse, 16 x 16 block
// products on v_mfma_f64_16x16x4 over row-major LDS images of stride LD.  Include inside no namespace; D64_STAMP may be predefined.
#pragma once
#include <hip/hip_runtime.h>
#ifndef D64_STAMP
#define D64_STAMP(slot) do { } while (0)
#endif

namespace mmg {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LD = 65;                     // row stride of the 64 x 64 LDS images (conflict-free column access)

__device__ __forceinline__ double rcp_f64(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = fma(-p, r, 1.0);
  r = fma(r, e, r);
  e = fma(-p, r, 1.0);
  return fma(r, e, r);
}

// ---- 64 x 64 Cholesky + inverse, blocked: panels inside ONE wave, the rest on the matrix pipe ------------------------------------
// The chains above pay an LDS write -> barrier -> read round trip (195 cycles) and a Newton reciprocal (dependent v_fma_f64 of 40
// cycles each) per column with four waves in lock step: 650 ns per column, 35 us per factorisation.  Here a 64 x 16 block column
// is factored by wave 0 alone with the matrix row-per-lane in registers: what a column step needs from other rows comes over
// v_readlane (no LDS, no barrier), and the NEXT pivot and its reciprocal are computed one step ahead from three readlanes
// (d - (u rp) u, the very operations the owning lane performs), so the dependent chain of a step is multiply, multiply-add,
// reciprocal -- the 15 row updates run in its shadow.  The trailing 16 x 16 blocks are updated by all four waves with
// v_mfma_f64_16x16x4 between two barriers per block column (8 barriers in all instead of 64).  Nothing is scaled inside the chain:
// the factor is kept as G = Lu D Lu' (Lu = I + F unit lower, D the pivots), L = Lu D^1/2 is formed by the caller's write-out.
// The inverse of Lu: its four diagonal blocks by forward substitution (wave w: block w, one lane per column, no division: unit
// diagonal), then with N[bi][bc] = Dinv[bi] F[bi][bc] (block-strictly lower, nilpotent) T = (I + N)^-1 by block rows of distance
// 1, 2, 3 (T[bi][bc] = -sum_k N[bi][k] T[k][bc]) and Lu^-1 = T blockdiag(Dinv): 6 + 2 + 1 + 6 block products of 16^3 on the matrix
// pipe in four barrier-separated stages.
// As: the matrix (row-major, stride LD; lower triangle + diagonal read) -> unscaled columns u_ik (i > k), the pivots p_k on the
//     diagonal, zeros above;   Fs -> F = u_ik / p_k (strictly lower; everything else zero);   T1 -> Lu^-1 (unit lower, row-major);
// Xs: scratch;  rs[64] -> 1 / sqrt(p_k);  sh: >= 2 ints.   L_ik = u_ik rs_k,  L_kk = p_k rs_k,  (L^-1)_ik = rs_i Lu^-1_ik.
// Returns 0, or 1 + the index of the first non-positive / non-finite pivot (replaced by 1: everything stays finite), uniform.
__device__ __forceinline__ double lane_val64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// one Newton step on v_rcp_f64: up to 20 ulp off (2.2e-15, measured by tools/probe/head_probe.hip) -- NOT used by the panels: with
// the two-step rcp_f64 they take the same 13 us (a panel step is bound by its ~45 v_readlane / v_fma issues, not by the chain)
__device__ __forceinline__ double rcp1_f64(double p) {
  const double r = __builtin_amdgcn_rcp(p);
  return fma(r, fma(-p, r, 1.0), r);
}
__device__ __forceinline__ v4d blk_load(const double* __restrict__ Cm, int r0, int c0, int lr, int lk) {
  v4d acc;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) acc[rr] = Cm[(r0 + 4 * rr + lk) * LD + c0 + lr];
  return acc;
}
__device__ __forceinline__ void blk_store(double* __restrict__ Cm, int r0, int c0, int lr, int lk, v4d acc) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Cm[(r0 + 4 * rr + lk) * LD + c0 + lr] = acc[rr];
}
// acc += sgn * A[ar.., ac..] B[br.., bc..] (16 x 16 blocks of row-major LDS images)
__device__ __forceinline__ v4d blk_mma(v4d acc, const double* __restrict__ Am, int ar, int ac, const double* __restrict__ Bm, int br,
                                       int bc, int lr, int lk, double sgn) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * Am[(ar + lr) * LD + ac + 4 * ks + lk], Bm[(br + 4 * ks + lk) * LD + bc + lr], acc, 0, 0, 0);
  return acc;
}
// ... with the B operand transposed: acc += sgn * A[ar.., ac..] (B[br.., bc..])'
__device__ __forceinline__ v4d blk_mma_nt(v4d acc, const double* __restrict__ Am, int ar, int ac, const double* __restrict__ Bm, int br,
                                          int bc, int lr, int lk, double sgn) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * Am[(ar + lr) * LD + ac + 4 * ks + lk], Bm[(br + lr) * LD + bc + 4 * ks + lk], acc, 0, 0, 0);
  return acc;
}
// ... with the A operand transposed: acc += sgn * (A[ak.., ai..])' B[bk.., bj..]  (the contraction runs over the ROWS of both)
__device__ __forceinline__ v4d blk_mma_tn(v4d acc, const double* __restrict__ Am, int ak, int ai, const double* __restrict__ Bm, int bk,
                                          int bj, int lr, int lk, double sgn) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * Am[(ak + 4 * ks + lk) * LD + ai + lr], Bm[(bk + 4 * ks + lk) * LD + bj + lr], acc, 0, 0, 0);
  return acc;
}
// t -> (r, c), r > c: the six strictly-lower blocks of a 4 x 4 block matrix
__device__ __forceinline__ void lower_block(int t, int& r, int& c) {
  r = t >= 3 ? 3 : (t >= 1 ? 2 : 1);
  c = t - r * (r - 1) / 2;
}
// Lu^-1 -> T1 for Lu = I + Fs (unit lower); Xs scratch (N is kept in T1's strictly-lower blocks until the last stage overwrites them).
// Ends with a barrier.
__device__ __forceinline__ void unit_lower_inverse64(const double* __restrict__ Fs, double* __restrict__ Xs, double* __restrict__ T1,
                                                     int tid) {
  double* const T2 = T1;
  const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  if (l < 16) {                                                // Dinv[w] = (I + F_ww)^-1: lane = column
    const int b0 = 16 * w;
    double t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = k == l ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
#pragma unroll
      for (int i = k + 1; i < 16; ++i) t[i] = fma(-Fs[(b0 + i) * LD + b0 + k], t[k], t[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { Xs[(b0 + i) * LD + b0 + l] = t[i]; T1[(b0 + i) * LD + b0 + l] = t[i]; }
  }
  __syncthreads();
  for (int t = w; t < 6; t += 4) {                             // N[r][c] = Dinv[r] F[r][c] -> T2 (N lives in T2's strictly-lower blocks)
    int r, c;
    lower_block(t, r, c);
    blk_store(T2, 16 * r, 16 * c, lr, lk, blk_mma(zero, Xs, 16 * r, 16 * r, Fs, 16 * r, 16 * c, lr, lk, 1.0));
  }
  __syncthreads();
  // T = (I + N)^-1 into the strictly-lower blocks of Xs (its diagonal blocks keep Dinv): distance 1 and 2 now, (3, 0) after them
  for (int t = w; t < 5; t += 4) {
    const int r = t < 3 ? t + 1 : t - 1, c = t < 3 ? t : t - 3;   // (1,0) (2,1) (3,2) | (2,0) (3,1)
    v4d acc = blk_load(T2, 16 * r, 16 * c, lr, lk);
    acc = -acc;
    if (r - c == 2) acc = blk_mma(acc, T2, 16 * r, 16 * (c + 1), T2, 16 * (c + 1), 16 * c, lr, lk, 1.0);   // + N[r][c+1] N[c+1][c]
    blk_store(Xs, 16 * r, 16 * c, lr, lk, acc);
  }
  __syncthreads();
  if (w == 0) {                                                // T30 = -N30 - N31 T10 - N32 T20
    v4d acc = blk_load(T2, 48, 0, lr, lk);
    acc = -acc;
    acc = blk_mma(acc, T2, 48, 16, Xs, 16, 0, lr, lk, -1.0);
    acc = blk_mma(acc, T2, 48, 32, Xs, 32, 0, lr, lk, -1.0);
    blk_store(Xs, 48, 0, lr, lk, acc);
  }
  __syncthreads();
  for (int t = w; t < 6; t += 4) {                             // Lu^-1[r][c] = T[r][c] Dinv[c]
    int r, c;
    lower_block(t, r, c);
    blk_store(T1, 16 * r, 16 * c, lr, lk, blk_mma(zero, Xs, 16 * r, 16 * c, Xs, 16 * c, 16 * c, lr, lk, 1.0));
  }
  for (int t = w; t < 6; t += 4) {                             // zeros above the block diagonal
    int r, c;
    lower_block(t, r, c);
    blk_store(T1, 16 * c, 16 * r, lr, lk, zero);
  }
  __syncthreads();
}
__device__ __forceinline__ int chol64_lds(double* __restrict__ As, double* __restrict__ Fs, double* __restrict__ Xs,
                                          double* __restrict__ T1, double* __restrict__ rs, int* __restrict__ sh, int tid) {
  const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  if (tid == 0) sh[0] = 0;
  for (int e = tid; e < 64 * LD; e += 256) Fs[e] = 0.0;
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    const int c0 = 16 * kb;
    __syncthreads();
    if (w == 0) {
      double a[16], pv[16], rpv[16];
      int bad = 0;
#pragma unroll
      for (int m = 0; m < 16; ++m) a[m] = As[l * LD + c0 + m];
      double p = lane_val64(a[0], c0);
      if (!(p > 0.0 && p < 1e300)) { bad = c0 + 1; p = 1.0; }
      double rp = rcp_f64(p);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jg = c0 + j;
        pv[j] = p;
        rpv[j] = rp;
        double pn = 1.0, rpn = 1.0;
        if (j < 15) {                                          // the next pivot, one step ahead: exactly lane jg + 1's own arithmetic
          const double u = lane_val64(a[j], jg + 1);
          pn = fma(-(u * rp), u, lane_val64(a[j + 1], jg + 1));
          if (!(pn > 0.0 && pn < 1e300)) { if (!bad) bad = jg + 2; pn = 1.0; }
          rpn = rcp_f64(pn);
        }
        const double f = l > jg ? a[j] * rp : 0.0;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) a[k] = fma(-f, lane_val64(a[j], c0 + k), a[k]);   // u of row c0 + k, column j
        p = pn;
        rp = rpn;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        As[l * LD + c0 + j] = l > c0 + j ? a[j] : (l == c0 + j ? pv[j] : 0.0);
        if (l > c0 + j) Fs[l * LD + c0 + j] = a[j] * rpv[j];
      }
      if (l == 0 && bad && sh[0] == 0) sh[0] = bad;
    }
    __syncthreads();
    const int nb = 3 - kb;
    for (int t = w; t < nb * (nb + 1) / 2; t += 4) {           // trailing blocks (bi >= bc > kb): C -= F U'
      const int r = t >= 3 ? 2 : (t >= 1 ? 1 : 0), c = t - r * (r + 1) / 2;
      const int i0 = 16 * (kb + 1 + r), j0 = 16 * (kb + 1 + c);
      blk_store(As, i0, j0, lr, lk, blk_mma_nt(blk_load(As, i0, j0, lr, lk), Fs, i0, c0, As, j0, c0, lr, lk, -1.0));
    }
  }
  __syncthreads();
  D64_STAMP(20);
  if (tid < 64) rs[tid] = 1.0 / sqrt(As[tid * LD + tid]);
  unit_lower_inverse64(Fs, Xs, T1, tid);                        // (ends with a barrier: rs is visible too)
  D64_STAMP(21);
  return sh[0];
}

}  // namespace mmg
