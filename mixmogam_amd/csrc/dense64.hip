// dense64.hip -- see dense64.h.  All kernels are fp64; the only matrix-pipe kernel is nt_update_lower_kernel
// (v_mfma_f64_16x16x4_f64, operands straight from global memory: at 64 cycles per instruction the matrix pipe leaves the
// vector memory path idle enough that an LDS stage buys nothing for a contraction of 64-128).  The 64 x 64 "head" kernels
// are single-workgroup latency chains (Cholesky / LU with one barrier per column); what they buy is launches: a panel of the
// band reduction was 67 launches of panel_qr_step_kernel and is 6 launches here.
#include "dense64.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace mmg {

// -DMMG_D64_STAMPS: the head kernels leave wall-clock stamps (100 MHz) of their phases in d64_stamps (tools/probe/head_probe.hip)
#ifdef MMG_D64_STAMPS
__device__ unsigned long long d64_stamps[64];
#define D64_STAMP(slot) do { if (threadIdx.x == 0) d64_stamps[slot] = wall_clock64(); } while (0)
#else
#define D64_STAMP(slot) do { } while (0)
#endif
}  // namespace mmg
#include "dense64_dev.h"                    // v4d, LD, rcp_f64, the blocked 64 x 64 Cholesky + inverse (shared with reml_band.hip)
namespace mmg {

// ---- tall-skinny Gram slices ------------------------------------------------------------------------------------------
// part[slice] = A[rows]' B[rows] on the matrix pipe, operands straight from global memory: lane l supplies row k0 + l / 16 of
// column l % 16 (32-byte runs of 16 columns: poor coalescing, but a 64-column panel is L2 resident and a 16x16x4 fp64 MFMA
// takes 64 cycles -- five loads per four of them).  Wave w owns rows 16 w .. 16 w + 15 of the result.  (The first version
// staged 32-row chunks through LDS for VALU multiply-adds: 53 us per 5000-row launch, LDS-read bound; this one: see DESIGN.)
__global__ __launch_bounds__(256) void gram_slices_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B0,
                                                          int64_t ldb0, int kb0, double* __restrict__ part0,
                                                          const double* __restrict__ B1, int64_t ldb1, int kb1,
                                                          double* __restrict__ part1, int64_t n, int rows_per) {
  // blockIdx.y picks the right-hand operand (two products with the same A in one launch)
  const double* __restrict__ B = blockIdx.y ? B1 : B0;
  const int64_t ldb = blockIdx.y ? ldb1 : ldb0;
  const int kb = blockIdx.y ? kb1 : kb0;
  double* __restrict__ part = blockIdx.y ? part1 : part0;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per, r_end = r_begin + rows_per < n ? r_begin + rows_per : n;
  v4d acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = v4d{0.0, 0.0, 0.0, 0.0};
  const double* __restrict__ ap = A + (int64_t)(16 * w + lr) * lda;
  const double* bp[4];
  bool bin[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int col = 16 * ct + lr;
    bin[ct] = col < kb;
    bp[ct] = B + (int64_t)(bin[ct] ? col : 0) * ldb;
  }
#pragma unroll 8
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 4) {
    const int64_t r = r0 + lk;
    const bool in = r < r_end;
    const double a = in ? ap[r] : 0.0;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const double bv = (in && bin[ct]) ? bp[ct][r] : 0.0;
      acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[ct], 0, 0, 0);
    }
  }
  double* o = part + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) o[(16 * w + 4 * rr + lk) + 64 * (16 * ct + lr)] = acc[ct][rr];
}

__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ part, int G, double* __restrict__ out, int ldo) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  double s = 0.0;
#pragma unroll 8
  for (int g = 0; g < G; ++g) s += part[(size_t)g * 4096 + e];   // (unrolled: a rolled loop waits for one load per slice)
  out[(e & 63) + ldo * (e >> 6)] = s;
}

int launch_gram_slices2(hipStream_t st, const double* A, int64_t lda, int64_t n, const double* B0, int64_t ldb0, int kb0,
                        double* part0, const double* B1, int64_t ldb1, int kb1, double* part1, int max_slices) {
  int64_t rows_per = std::max<int64_t>(64, (n + max_slices - 1) / max_slices);
  rows_per = (rows_per + 3) / 4 * 4;
  const int G = (int)((n + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(gram_slices_kernel, dim3(G, B1 ? 2 : 1), dim3(256), 0, st, A, lda, B0, ldb0, kb0, part0, B1, ldb1, kb1, part1, n,
                     (int)rows_per);
  return G;
}

int launch_gram_slices(hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb, int kb, int64_t n,
                       double* part, int max_slices) {
  return launch_gram_slices2(st, A, lda, n, B, ldb, kb, part, nullptr, 0, 0, nullptr, max_slices);
}

void launch_gram_reduce(hipStream_t st, const double* part, int G, double* out, int ldo) {
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(16), dim3(256), 0, st, part, G, out, ldo);
}

// ---- 64 x 64 chains ------------------------------------------------------------------------------------------------------
// The matrix lives in REGISTERS: thread (i = tid & 63, kq = tid >> 6) holds a[m] = M[i][16 kq + m].  A step publishes what
// the others need through LDS (column j + 1 of the matrix, row j + 1 of the inverse being accumulated) into one of two
// buffers, so a step is ONE barrier, a few LDS writes, broadcast reads and 32 predicated multiply-adds.  The loop over the
// columns is ROLLED: every register index is static because a thread publishes "the entry whose column is j + 1" by
// comparing its 16 column numbers with j + 1, never by indexing with j.  (History, tools/probe/head_probe.hip: the matrix
// in LDS with read-modify-write per element, 85 us per factorisation; registers with the 64 steps fully unrolled, 58 us --
// 66 KB of straight-line code run once is an instruction-fetch stream, not a kernel; a dependent v_fma_f64 costs 40 cycles
// on this part and an LDS write -> barrier -> read round trip 195, tools/probe/chain_latency.hip.)
//
// What keeps a step short is the INSTRUCTION COUNT (a dependent v_fma_f64 costs 40 cycles on this part, an LDS write ->
// barrier -> read round trip 195: tools/probe/chain_latency.hip): no predicate per element -- the published vectors carry
// zeros where an update must not happen -- one Newton reciprocal instead of an IEEE division.
// Cholesky G = L L' with the inverse for free: the row operations that eliminate column j are applied to an identity as
// well, which leaves Bm = Lu^-1 for the unit-lower Lu of G = Lu D Lu'.  On exit (k = 16 kq + m):
//   a[m] (k < i) = the unscaled entry of column k: L_ik = a / sqrt(piv[k]);   b[m] (k <= i) = Lu^-1[i][k]: L^-1[i][k] = b / sqrt(piv[i])
// (a[m] for k > i is scratch).  The published column j holds zeros in rows <= j, so rows at or above the pivot get a zero
// multiplier and the column's own entries a_ij stay what they are.  A non-positive / non-finite pivot is replaced by 1
// (everything stays finite) and reported: returns 0, or 1 + the index of the first one (uniform).
// LDS: cb, rb: [2][64] doubles each; pb: [2]; piv: [64].
__device__ __forceinline__ int chol64_inv_reg(double (&a)[16], double (&b)[16], double* __restrict__ cb, double* __restrict__ rb,
                                              double* __restrict__ pb, double* __restrict__ piv, int tid) {
  const int i = tid & 63, kq = tid >> 6;
  // the thread that holds the next pivot checks it and publishes its reciprocal: the Newton steps then run beside that
  // wave's own multiply-adds instead of in front of everybody's
  auto publish_pivot = [&](int jn, double v) {
    const bool ok = v > 0.0 && v < 1e300;
    const double p = ok ? v : 1.0;
    piv[jn] = p;
    pb[2 * (jn & 1)] = rcp_f64(p);
    if (!ok && pb[4] == 0.0) pb[4] = (double)(jn + 1);
  };
#pragma unroll
  for (int m = 0; m < 16; ++m) b[m] = 16 * kq + m == i ? 1.0 : 0.0;
  if (tid == 0) pb[4] = 0.0;
  __syncthreads();
  if (kq == 0) {
    cb[i] = i > 0 ? a[0] : 0.0;
    if (i == 0) publish_pivot(0, a[0]);
  }
  if (i == 0) {
#pragma unroll
    for (int m = 0; m < 16; ++m) rb[16 * kq + m] = b[m];
  }
  // 4 x 16 steps: the inner 16 are unrolled so that "the entry of column j + 1" is a[(jm + 1) % 16], a static register (a
  // 16-way select on j made the compiler keep a[] in scratch memory: a store and a load of global-memory latency per step,
  // 1.3 us); 16 steps of code (14 KB) stay in the instruction cache, 64 (the first version) do not
#pragma unroll 1
  for (int jq = 0; jq < 4; ++jq)
#pragma unroll
  for (int jm = 0; jm < 16; ++jm) {
    const int j = 16 * jq + jm;
    const double* __restrict__ c = cb + (j & 1) * 64;
    const double* __restrict__ rw = rb + (j & 1) * 64;
    double* __restrict__ cn = cb + ((j + 1) & 1) * 64;
    double* __restrict__ rn = rb + ((j + 1) & 1) * 64;
    __syncthreads();
    const double f = c[i] * pb[2 * (j & 1)];                   // 0 for the rows i <= j
    const int jn = j + 1;
    if (kq == jq + (jm == 15 ? 1 : 0)) {                       // wave-uniform: the wave that holds column j + 1
      const int mn = (jm + 1) & 15;
      a[mn] = fma(-f, c[16 * kq + mn], a[mn]);
      cn[i] = i > jn ? a[mn] : 0.0;
      if (i == jn) publish_pivot(jn, a[mn]);
#pragma unroll
      for (int m = 0; m < 16; ++m)
        if (m != mn) a[m] = fma(-f, c[16 * kq + m], a[m]);     // c[k] = 0 for k <= j
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) a[m] = fma(-f, c[16 * kq + m], a[m]);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) b[m] = fma(-f, rw[16 * kq + m], b[m]);   // row j of the inverse: zero beyond column j
    if (i == jn) {
#pragma unroll
      for (int m = 0; m < 16; ++m) rn[16 * kq + m] = b[m];
    }
  }
  __syncthreads();
  return (int)pb[4];
}

// Gauss-Jordan inverse of Q - S where the sign matrix S is chosen on the fly, S_jj = -sgn(current diagonal entry), so that
// every pivot has magnitude >= 1 (the sign rule of Ballard et al. 2014, Alg. 5, applied to a full elimination instead of an
// LU).  a: Q (row i, 16 columns per thread); on exit b[m] = the unnormalised inverse: (Q - S)^-1[i][k] = b[m] / pv[i]; sg[j]
// = S_jj.  LDS: cb: [2][64], rab: [2][128] (row j of both halves, its diagonal entry already the pivot), pb: [2][2] (1 / pivot).
__device__ __forceinline__ void gj64_signed_reg(double (&a)[16], double (&b)[16], double* __restrict__ cb, double* __restrict__ rab,
                                                double* __restrict__ pb, double* __restrict__ pv, double* __restrict__ sg, int tid) {
  const int i = tid & 63, kq = tid >> 6;
  auto publish_pivot = [&](int jn, double d, double* __restrict__ row) {
    const double s = d >= 0.0 ? -1.0 : 1.0, p = d - s;
    row[jn] = p;
    pv[jn] = p;
    sg[jn] = s;
    pb[2 * (jn & 1)] = rcp_f64(p);
  };
#pragma unroll
  for (int m = 0; m < 16; ++m) b[m] = 16 * kq + m == i ? 1.0 : 0.0;
  if (kq == 0) cb[i] = a[0];
  if (i == 0) {
#pragma unroll
    for (int m = 0; m < 16; ++m) { rab[16 * kq + m] = a[m]; rab[64 + 16 * kq + m] = b[m]; }
    if (kq == 0) publish_pivot(0, a[0], rab);
  }
#pragma unroll 1
  for (int jq = 0; jq < 4; ++jq)
#pragma unroll
  for (int jm = 0; jm < 16; ++jm) {
    const int j = 16 * jq + jm;
    const double* __restrict__ c = cb + (j & 1) * 64;
    const double* __restrict__ rw = rab + (j & 1) * 128;
    double* __restrict__ cn = cb + ((j + 1) & 1) * 64;
    double* __restrict__ rn = rab + ((j + 1) & 1) * 128;
    __syncthreads();
    const double f = i == j ? 0.0 : c[i] * pb[2 * (j & 1)];
    const int jn = j + 1, mn = (jm + 1) & 15;
    const bool holder = kq == jq + (jm == 15 ? 1 : 0);         // wave-uniform: the wave that holds column j + 1
    a[mn] = fma(-f, rw[16 * kq + mn], a[mn]);
    if (holder) cn[i] = a[mn];
#pragma unroll
    for (int m = 0; m < 16; ++m)
      if (m != mn) a[m] = fma(-f, rw[16 * kq + m], a[m]);
#pragma unroll
    for (int m = 0; m < 16; ++m) b[m] = fma(-f, rw[64 + 16 * kq + m], b[m]);
    if (i == jn) {
#pragma unroll
      for (int m = 0; m < 16; ++m) { rn[16 * kq + m] = a[m]; rn[64 + 16 * kq + m] = b[m]; }
      if (holder) publish_pivot(jn, a[mn], rn);
    }
  }
  __syncthreads();
}

// C = A B for 64 x 64 operands given as element functors (LDS reads), on the matrix pipe: wave w computes rows 16 w .. 16 w + 15;
// acc[jt][rr] = C[16 w + 4 rr + l / 16][16 jt + l % 16]
template <class FA, class FB>
__device__ __forceinline__ void mm64_mfma(int tid, FA fa, FB fb, v4d (&acc)[4]) {
  const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) acc[jt] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int ks = 0; ks < 16; ++ks) {
    const double av = fa(16 * w + lr, 4 * ks + lk);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, fb(4 * ks + lk, 16 * jt + lr), acc[jt], 0, 0, 0);
  }
}
template <class FC>
__device__ __forceinline__ void mm64_store(int tid, const v4d (&acc)[4], FC fc) {
  const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) fc(16 * w + 4 * rr + lk, 16 * jt + lr, acc[jt][rr]);
}

// LU factorisation without pivoting of Q - S, S_jj = -sgn(current diagonal entry) chosen at step j so that |pivot| >= 1 (the rule
// of gj64_signed_reg), blocked like chol64_lds: wave 0 eliminates the 64 x 16 block column (lane = row: the multipliers), wave 1
// -- at the same time, on its own SIMD -- the 16-row block row (lane = column: the rows of U to the right of the diagonal block);
// both walk the same pivots with the same arithmetic, so they need nothing from each other; then C -= F U on the matrix pipe.
// Bq: Q (row-major, stride LD) -> U on and above the diagonal (the modified pivots on it; what is below is scratch);
// Fq -> the multipliers (strictly lower, zeros elsewhere);  pv[64], sg[64] -> pivots and signs.  Ends with a barrier.
__device__ __forceinline__ void lu64_signed_lds(double* __restrict__ Bq, double* __restrict__ Fq, double* __restrict__ pv,
                                                double* __restrict__ sg, int tid) {
  const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  for (int e = tid; e < 64 * LD; e += 256) Fq[e] = 0.0;
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    const int c0 = 16 * kb;
    __syncthreads();
    if (w == 0) {
      double a[16], pvv[16], rpv[16], sgv[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) a[m] = Bq[l * LD + c0 + m];
      double d = lane_val64(a[0], c0);
      double sn = d >= 0.0 ? -1.0 : 1.0, p = d - sn, rp = rcp_f64(p);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jg = c0 + j;
        pvv[j] = p; rpv[j] = rp; sgv[j] = sn;
        double pn = 1.0, rpn = 1.0, snn = 1.0;
        if (j < 15) {                                          // the next pivot one step ahead, with row jg + 1's own arithmetic
          const double dn = fma(-(lane_val64(a[j], jg + 1) * rp), lane_val64(a[j + 1], jg), lane_val64(a[j + 1], jg + 1));
          snn = dn >= 0.0 ? -1.0 : 1.0;
          pn = dn - snn;
          rpn = rcp_f64(pn);
        }
        const double f = l > jg ? a[j] * rp : 0.0;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) a[k] = fma(-f, lane_val64(a[k], jg), a[k]);   // the pivot row's entry of column c0 + k
        p = pn; rp = rpn; sn = snn;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (l > c0 + j) Fq[l * LD + c0 + j] = a[j] * rpv[j];
        else if (l >= c0) Bq[l * LD + c0 + j] = l == c0 + j ? pvv[j] : a[j];
        if (l == 0) { pv[c0 + j] = pvv[j]; sg[c0 + j] = sgv[j]; }
      }
    } else if (w == 1 && kb < 3) {
      double t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = Bq[(c0 + r) * LD + l];
      double d = lane_val64(t[0], c0);
      double sn = d >= 0.0 ? -1.0 : 1.0, p = d - sn, rp = rcp_f64(p);
#pragma unroll
      for (int j = 0; j < 15; ++j) {
        const int jg = c0 + j;
        const double dn = fma(-(lane_val64(t[j + 1], jg) * rp), lane_val64(t[j], jg + 1), lane_val64(t[j + 1], jg + 1));
        const double snn = dn >= 0.0 ? -1.0 : 1.0, pn = dn - snn, rpn = rcp_f64(pn);
#pragma unroll
        for (int r = j + 1; r < 16; ++r) t[r] = fma(-(lane_val64(t[r], jg) * rp), t[j], t[r]);   // multiplier of row c0 + r, uniform
        p = pn; rp = rpn; sn = snn;
      }
      if (l >= c0 + 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Bq[(c0 + r) * LD + l] = t[r];
      }
    }
    __syncthreads();
    const int nb = 3 - kb;
    for (int t = w; t < nb * nb; t += 4) {                     // C -= F[.., kb] U[kb, ..] on the trailing blocks
      const int i0 = 16 * (kb + 1 + t / nb), j0 = 16 * (kb + 1 + t % nb);
      blk_store(Bq, i0, j0, lr, lk, blk_mma(blk_load(Bq, i0, j0, lr, lk), Fq, i0, c0, Bq, c0, j0, lr, lk, -1.0));
    }
  }
  __syncthreads();
}

// ---- Y = X0 C0 (+ X1 C1) for tall X and 64 x 64 coefficients, on the matrix pipe ----------------------------------------------
// One workgroup per 64 rows (wave w: rows 16 w .. 16 w + 15, all 64 columns; the transposed tile is computed so that a
// register holds 16 consecutive rows of one column of Y: 128-byte stores).  The coefficient matrices sit in LDS transposed
// with a row stride of 68 doubles (a 16 x 4 fragment read touches every bank pair twice: the minimum for 512 bytes).  A
// wave reads its 16 rows of X completely before it stores: Y may be X0.  WITHZ: the same workgroup rotates its rows of
// [X y]:  Z[r][c] -= sum_k X0[r][k] Cz[k][c].
constexpr int YLD = 68;
// gram != nullptr (round 6): the workgroup also leaves the Gram matrix of ITS 64 rows of Y, Y[rows]'Y[rows], as slice blockIdx.x of
// `gram` (the layout of gram_slices_kernel): the second pass of Cholesky-QR needs Q1'Q1 of the Q1 this kernel has just produced.
template <bool TWO>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const double* __restrict__ X0, int64_t ld0, const double* __restrict__ C0,
                                                        const double* __restrict__ X1, int64_t ld1, const double* __restrict__ C1,
                                                        double* __restrict__ Y, int64_t ldy, int64_t n, double* __restrict__ Z,
                                                        int64_t ldz, const double* __restrict__ Cz, int q1, int c0_transposed,
                                                        double* __restrict__ gram) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Cs = sm;                         // [j][k] = C0[k][j]
  double* Ts = sm + 64 * YLD;              // [j][k] = C1[k][j]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  for (int e = tid; e < 4096; e += 256) {
    const int k = e & 63, j = e >> 6;
    Cs[j * YLD + k] = c0_transposed ? C0[j + 64 * k] : C0[e];  // (transposed: the coefficient is C0')
    if (TWO) Ts[j * YLD + k] = C1[e];
  }
  const int64_t R0 = (int64_t)blockIdx.x * 64;
  const int64_t r = R0 + 16 * w + lr;
  const bool rin = r < n;
  const int64_t rc = rin ? r : n - 1;
  double fv[16], fw[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    fv[ks] = X0[rc + (int64_t)(4 * ks + lk) * ld0];
    if (TWO) fw[ks] = X1[rc + (int64_t)(4 * ks + lk) * ld1];
  }
  __syncthreads();
  v4d acc[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) acc[jt] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cs[(16 * jt + lr) * YLD + 4 * ks + lk], fv[ks], acc[jt], 0, 0, 0);
      if (TWO) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ts[(16 * jt + lr) * YLD + 4 * ks + lk], fw[ks], acc[jt], 0, 0, 0);
    }
  }
  if (rin) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Y[r + (int64_t)(16 * jt + 4 * rr + lk) * ldy] = acc[jt][rr];
  }
  if (!TWO && gram) {
    __syncthreads();                                           // the coefficient image has been read by every wave
    double* Ys = sm;                                           // [row][col] of this workgroup's 64 x 64 tile of Y, zero rows beyond n
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Ys[(16 * w + lr) * YLD + 16 * jt + 4 * rr + lk] = rin ? acc[jt][rr] : 0.0;
    __syncthreads();
    v4d g[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) g[ct] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const double av = Ys[(4 * ks + lk) * YLD + 16 * w + lr];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) g[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Ys[(4 * ks + lk) * YLD + 16 * ct + lr], g[ct], 0, 0, 0);
    }
    double* o = gram + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) o[(16 * w + 4 * rr + lk) + 64 * (16 * ct + lr)] = g[ct][rr];
  }
  if (Z) {
    // rows of [X y]: thread (row l, columns w, w + 4, ..); X0 is not the output here
    const int64_t rz = R0 + l;
    if (rz < n) {
      for (int c = w; c < q1; c += 4) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int k = 0; k < 64; k += 4) {
          s0 = fma(X0[rz + (int64_t)k * ld0], Cz[k + 64 * c], s0);
          s1 = fma(X0[rz + (int64_t)(k + 1) * ld0], Cz[k + 1 + 64 * c], s1);
          s2 = fma(X0[rz + (int64_t)(k + 2) * ld0], Cz[k + 2 + 64 * c], s2);
          s3 = fma(X0[rz + (int64_t)(k + 3) * ld0], Cz[k + 3 + 64 * c], s3);
        }
        Z[rz + (int64_t)c * ldz] -= (s0 + s1) + (s2 + s3);
      }
    }
  }
}
constexpr int ROWS_LDS1 = 64 * YLD * (int)sizeof(double), ROWS_LDS2 = 2 * ROWS_LDS1;

void launch_rows_gemm(hipStream_t st, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t n, const double* Cf,
                      bool cf_transposed) {
  if (n <= 0) return;
  hipLaunchKernelGGL(rows_gemm_kernel<false>, dim3((unsigned)((n + 63) / 64)), dim3(256), ROWS_LDS1, st, X, ldx, Cf,
                     (const double*)nullptr, (int64_t)0, (const double*)nullptr, Y, ldy, n, (double*)nullptr, (int64_t)0,
                     (const double*)nullptr, 0, cf_transposed ? 1 : 0, (double*)nullptr);
}

int launch_rows_gemm_gram(hipStream_t st, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t n, const double* Cf,
                          double* gram) {
  const int G = (int)((n + 63) / 64);
  hipLaunchKernelGGL(rows_gemm_kernel<false>, dim3((unsigned)G), dim3(256), ROWS_LDS1, st, X, ldx, Cf, (const double*)nullptr,
                     (int64_t)0, (const double*)nullptr, Y, ldy, n, (double*)nullptr, (int64_t)0, (const double*)nullptr, 0, 0, gram);
  return G;
}

// out0 = sum of the G slices of part0, out1 likewise of part1 (two reductions of launch_gram_slices2's pair in one launch)
__global__ __launch_bounds__(256) void gram_reduce2_kernel(const double* __restrict__ part0, const double* __restrict__ part1, int G,
                                                           double* __restrict__ out0, double* __restrict__ out1) {
  const double* __restrict__ part = blockIdx.y ? part1 : part0;
  double* __restrict__ out = blockIdx.y ? out1 : out0;
  const int e = blockIdx.x * 256 + threadIdx.x;
  double s = 0.0;
#pragma unroll 8
  for (int g = 0; g < G; ++g) s += part[(size_t)g * 4096 + e];
  out[e] = s;
}
void launch_gram_reduce2(hipStream_t st, const double* part0, const double* part1, int G, double* out0, double* out1) {
  hipLaunchKernelGGL(gram_reduce2_kernel, dim3(16, 2), dim3(256), 0, st, part0, part1, G, out0, out1);
}


void launch_band_y(hipStream_t st, const double* V, const double* W, int64_t n, const double* T, const double* C, double* Y,
                   double* Z, int64_t ldz, const double* Cz, int q1) {
  hipLaunchKernelGGL(rows_gemm_kernel<true>, dim3((unsigned)((n + 63) / 64)), dim3(256), ROWS_LDS2, st, V, n, C, W, n, T, Y, n, n, Z,
                     ldz, Cz, q1, 0, (double*)nullptr);
}

// ---- C[I][J] -= A0[I] B0[J]' + A1[I] B1[J]' on the lower 64 x 64 tiles ---------------------------------------------------
// One workgroup per tile; wave w owns rows 16 w .. 16 w + 15 of it and all 64 columns.  The MFMA computes the TRANSPOSED
// tile (a operand = 16 columns j of B, b operand = the wave's 16 rows i of -A) so that an accumulator register holds 16
// consecutive rows of one column of C: 128-byte segments on the only traffic that matters here (C is read and written once
// per call; the operands are 64 KB per tile side and come out of L2).
template <int NP>
// gram != nullptr (round 6): the tiles (I >= 1, J = 0) -- the block column the NEXT panel of a blocked factorisation starts from --
// also leave the Gram matrix of their updated 64 x 64 tile as slice I - 1 of `gram` (gram_slices_kernel's layout): the next
// panel's P'P without a pass of its own over P.
__global__ __launch_bounds__(256, 2) void nt_update_lower_kernel(double* __restrict__ C, int64_t ldc, int64_t n,
                                                                 const double* __restrict__ A0, const double* __restrict__ B0,
                                                                 const double* __restrict__ A1, const double* __restrict__ B1,
                                                                 int64_t lda, int64_t ldb, int col0, double* __restrict__ gram) {
  const int64_t t = blockIdx.x;
  int64_t I = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  int64_t J = t - I * (I + 1) / 2;
  if (col0) { I = t; J = 0; }                                // only the first block column (launch_nt_update_col0)
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  const int64_t I0 = I * 64, J0 = J * 64;
  const int64_t ci = I0 + 16 * w + lr;                       // this lane's row of C
  const bool iin = ci < n;
  const int64_t cic = iin ? ci : n - 1;                      // loads are unconditional (clamped), masked afterwards
  int64_t jrow[4];
  bool jin[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const int64_t j = J0 + 16 * jt + lr;
    jin[jt] = j < n;
    jrow[jt] = jin[jt] ? j : n - 1;
  }
  // The contraction runs in chunks of 32 columns (8 k-steps = 40 loads per lane); chunk c + 1 is in flight while chunk c is
  // multiplied.  (First version: loads and MFMAs of one k-step next to each other, 133 registers, ~25 us per tile of which
  // 3.4 are MFMA issue -- every k-step waited for its own loads.)
  constexpr int NCH = 2 * NP;
  double fa[2][8], fb[2][8][4];
  auto load_chunk = [&](int ch, int buf) {
    const double* __restrict__ A = (ch >> 1) ? A1 : A0;
    const double* __restrict__ B = (ch >> 1) ? B1 : B0;
    const int kb = (ch & 1) * 32;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k = kb + 4 * ks + lk;
      fa[buf][ks] = A[cic + (int64_t)k * lda];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) fb[buf][ks][jt] = B[jrow[jt] + (int64_t)k * ldb];
    }
  };
  load_chunk(0, 0);
  v4d acc[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t cj = J0 + 16 * jt + 4 * r + lk;
      acc[jt][r] = C[cic + (cj < n ? cj : n - 1) * ldc];
    }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) load_chunk(ch + 1, (ch + 1) & 1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const double bop = iin ? -fa[ch & 1][ks] : 0.0;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const double aop = jin[jt] ? fb[ch & 1][ks][jt] : 0.0;
        acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[jt], 0, 0, 0);
      }
    }
  }
  if (iin) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t cj = J0 + 16 * jt + 4 * r + lk;
        if (cj < n) C[ci + cj * ldc] = acc[jt][r];
      }
  }
  if (gram && J == 0 && I >= 1) {                               // (workgroup-uniform)
    __shared__ double Ts[64 * YLD];                            // [row][col] of the updated tile, zero rows beyond n
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ts[(16 * w + lr) * YLD + 16 * jt + 4 * r + lk] = iin ? acc[jt][r] : 0.0;
    __syncthreads();
    v4d g[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) g[ct] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const double av = Ts[(4 * ks + lk) * YLD + 16 * w + lr];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) g[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Ts[(4 * ks + lk) * YLD + 16 * ct + lr], g[ct], 0, 0, 0);
    }
    double* o = gram + (size_t)(I - 1) * 4096;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(16 * w + 4 * r + lk) + 64 * (16 * ct + lr)] = g[ct][r];
  }
}

// The same update on 128 x 128 tiles, 8 waves per workgroup (2 along i x 4 along j: wave tile 64 x 32, two waves per SIMD --
// the fp64 matrix pipe sustains 48 TFLOP/s with two waves per SIMD against 35 with one, tools/probe/fp64_rate.hip): a tile
// side's operands are read from L2 once per 128 columns of output instead of once per 64, 6 loads per 8 MFMAs instead of 5
// per 4.  Taken for large trailing matrices (launch_nt_update_lower), where the 64 x 64 form ran at 30 TFLOP/s.
template <int NP>
__global__ __launch_bounds__(512, 1) void nt_update_lower128_kernel(double* __restrict__ C, int64_t ldc, int64_t n,
                                                                   const double* __restrict__ A0, const double* __restrict__ B0,
                                                                   const double* __restrict__ A1, const double* __restrict__ B1,
                                                                   int64_t lda, int64_t ldb) {
  const int64_t t = blockIdx.x;
  int64_t I = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int64_t J = t - I * (I + 1) / 2;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
  const int wi = w & 1, wj = w >> 1;
  const int64_t I0 = I * 128 + 64 * wi, J0 = J * 128 + 32 * wj;
  int64_t irow[4], jrow[2];
  bool iin[4], jin[2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t i = I0 + 16 * it + lr;
    iin[it] = i < n;
    irow[it] = iin[it] ? i : n - 1;
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int64_t j = J0 + 16 * jt + lr;
    jin[jt] = j < n;
    jrow[jt] = jin[jt] ? j : n - 1;
  }
  constexpr int NCH = 4 * NP;                                // chunks of 16 contraction columns (4 k-steps = 24 loads per lane)
  double fa[2][4][4], fb[2][4][2];
  auto load_chunk = [&](int ch, int buf) {
    const double* __restrict__ A = (ch >> 2) ? A1 : A0;
    const double* __restrict__ B = (ch >> 2) ? B1 : B0;
    const int kb = (ch & 3) * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k = kb + 4 * ks + lk;
#pragma unroll
      for (int it = 0; it < 4; ++it) fa[buf][ks][it] = A[irow[it] + (int64_t)k * lda];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) fb[buf][ks][jt] = B[jrow[jt] + (int64_t)k * ldb];
    }
  };
  load_chunk(0, 0);
  v4d acc[2][4];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t cj = J0 + 16 * jt + 4 * r + lk;
        acc[jt][it][r] = C[irow[it] + (cj < n ? cj : n - 1) * ldc];
      }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) load_chunk(ch + 1, (ch + 1) & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const double bop = iin[it] ? -fa[ch & 1][ks][it] : 0.0;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const double aop = jin[jt] ? fb[ch & 1][ks][jt] : 0.0;
          acc[jt][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[jt][it], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (iin[it]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t cj = J0 + 16 * jt + 4 * r + lk;
          if (cj < n) C[irow[it] + cj * ldc] = acc[jt][it][r];
        }
      }
}

int launch_nt_update_lower(hipStream_t st, double* C, int64_t ldc, int64_t n, const double* A0, const double* B0,
                           const double* A1, const double* B1, int64_t lda, int64_t ldb, double* gram_col0) {
  if (n <= 0) return 0;
  // large trailing matrices: 128 x 128 tiles (MMG_NT_TILE=64 | 128 overrides; default 128 from n = 4096, where a launch
  // still has >= 528 workgroups)
  static const int forced = [] { const char* e = std::getenv("MMG_NT_TILE"); return e ? std::atoi(e) : 0; }();
  if (forced == 128 || (forced != 64 && n >= 4096)) {
    const int64_t nt = (n + 127) / 128;
    const dim3 grid((unsigned)(nt * (nt + 1) / 2));
    if (A1) hipLaunchKernelGGL(nt_update_lower128_kernel<2>, grid, dim3(512), 0, st, C, ldc, n, A0, B0, A1, B1, lda, ldb);
    else hipLaunchKernelGGL(nt_update_lower128_kernel<1>, grid, dim3(512), 0, st, C, ldc, n, A0, B0, A0, B0, lda, ldb);
    return 0;
  }
  const int64_t nt = (n + 63) / 64;
  const dim3 grid((unsigned)(nt * (nt + 1) / 2));
  if (A1) hipLaunchKernelGGL(nt_update_lower_kernel<2>, grid, dim3(256), 0, st, C, ldc, n, A0, B0, A1, B1, lda, ldb, 0, gram_col0);
  else hipLaunchKernelGGL(nt_update_lower_kernel<1>, grid, dim3(256), 0, st, C, ldc, n, A0, B0, A0, B0, lda, ldb, 0, gram_col0);
  return gram_col0 ? (int)(nt - 1) : 0;                       // slices written: the 64-row tiles below the first
}

// the same update on the FIRST 64-column block column alone (tiles (I, 0), I = 0 ..): what the next panel of a blocked
// factorisation needs before the rest of the trailing matrix is done (reml_band.hip: look-ahead)
void launch_nt_update_col0(hipStream_t st, double* C, int64_t ldc, int64_t n, const double* A0, const double* B0,
                           const double* A1, const double* B1, int64_t lda, int64_t ldb) {
  if (n <= 0) return;
  const dim3 grid((unsigned)((n + 63) / 64));
  if (A1) hipLaunchKernelGGL(nt_update_lower_kernel<2>, grid, dim3(256), 0, st, C, ldc, n, A0, B0, A1, B1, lda, ldb, 1, (double*)nullptr);
  else hipLaunchKernelGGL(nt_update_lower_kernel<1>, grid, dim3(256), 0, st, C, ldc, n, A0, B0, A0, B0, lda, ldb, 1, (double*)nullptr);
}

// ---- Cholesky-QR heads --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cholqr_head1_kernel(const double* __restrict__ part, int G, double* __restrict__ R1,
                                                           double* __restrict__ R1inv, PanelFlags* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) double cb[2 * 64], rb[2 * 64], pb[6];
  __shared__ double piv[64];
  const int tid = threadIdx.x, i = tid & 63, kq = tid >> 6;
  D64_STAMP(0);
  double a[16], b[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) a[m] = 0.0;
#pragma unroll 4
  for (int g = 0; g < G; ++g) {
    const double* __restrict__ pg = part + (size_t)g * 4096 + i + 64 * 16 * kq;
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] += pg[64 * m];
  }
  D64_STAMP(1);
  const int bad = chol64_inv_reg(a, b, cb, rb, pb, piv, tid);
  D64_STAMP(2);
  // R1 = L' and its inverse, both column-major upper: R1[k + 64 i] = L_ik, R1inv[k + 64 i] = (L^-1)_ik
  const double rsi = 1.0 / sqrt(piv[i]);
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = 16 * kq + m;
    R1[k + 64 * i] = k < i ? a[m] / sqrt(piv[k]) : (k == i ? sqrt(piv[i]) : 0.0);
    R1inv[k + 64 * i] = k <= i ? b[m] * rsi : 0.0;
  }
  if (tid == 0 && bad) flags->bad = 1;
  D64_STAMP(3);
}

// the same on the blocked factorisation (chol64_lds): 4 LDS images
constexpr int CHOL_LDS = (4 * 64 * LD + 64 + 8) * (int)sizeof(double);
__global__ __launch_bounds__(256) void cholqr_head1_blk_kernel(const double* __restrict__ part, int G, double* __restrict__ R1,
                                                               double* __restrict__ R1inv, PanelFlags* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *As = sm, *Fs = sm + 64 * LD, *Xs = sm + 2 * 64 * LD, *T1 = sm + 3 * 64 * LD, *rs = sm + 4 * 64 * LD;
  int* sh = (int*)(rs + 64);
  const int tid = threadIdx.x, i = tid & 63, kq = tid >> 6;
  D64_STAMP(0);
  {
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = 0.0;
#pragma unroll 4
    for (int g = 0; g < G; ++g) {
      const double* __restrict__ pg = part + (size_t)g * 4096 + i + 64 * 16 * kq;
#pragma unroll
      for (int m = 0; m < 16; ++m) a[m] += pg[64 * m];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) As[i * LD + 16 * kq + m] = a[m];
  }
  __syncthreads();
  D64_STAMP(1);
  const int bad = chol64_lds(As, Fs, Xs, T1, rs, sh, tid);
  D64_STAMP(2);
  // R1 = L' and its inverse, both column-major upper: R1[k + 64 i] = L_ik = u_ik rs_k, R1inv[k + 64 i] = (L^-1)_ik = rs_i Lu^-1_ik
  const double rsi = rs[i];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = 16 * kq + m;
    R1[k + 64 * i] = As[i * LD + k] * rs[k];
    R1inv[k + 64 * i] = T1[i * LD + k] * rsi;
  }
  if (tid == 0 && bad) flags->bad = 1;
  D64_STAMP(3);
}

static bool head_blocked() {
  static const bool on = [] { const char* e = std::getenv("MMG_HEAD_CHAIN"); return !(e && e[0] == '1'); }();   // MMG_HEAD_CHAIN=1: round 5's chains (A/B)
  return on;
}

void launch_cholqr_head1(hipStream_t st, const double* part, int G, double* R1, double* R1inv, PanelFlags* flags) {
  if (head_blocked()) hipLaunchKernelGGL(cholqr_head1_blk_kernel, dim3(1), dim3(256), CHOL_LDS, st, part, G, R1, R1inv, flags);
  else hipLaunchKernelGGL(cholqr_head1_kernel, dim3(1), dim3(256), 0, st, part, G, R1, R1inv, flags);
}

// Second pass + the Householder-like representation of Q = Q1 R2^-1 in Yamamoto's basis-kernel form with the adaptive signs
// of Ballard et al. (2014):  H = I - V M V' with  V = [I; 0] - Q S  and  M = (I - Qtop S)^-T  is orthogonal and H [I; 0] = Q S,
// hence  H' P = [S R2 R1; 0].  (M is a general 64 x 64 matrix, not the triangular T of the compact WY form; nothing
// downstream needs the triangle: the two-sided update is  A - X V' - V X',  X = W M - 1/2 V M'(V'W) M.)  I - Qtop S =
// -(Qtop - S) S is inverted by Gauss-Jordan elimination of Qtop - S with S_jj chosen at step j so that |pivot| >= 1.
// The rows below the top block need no solve at all:  V[64:] = Q1[64:] Cb,  Cb = -R2^-1 S.
constexpr int HEAD2_LDS = (4 * 64 * LD + 2 * 64 + 2 * 128 + 3 * 64 + 8) * (int)sizeof(double);
template <bool BLK>
__global__ __launch_bounds__(256) void cholqr_head2_kernel(const double* __restrict__ part, int G, const double* __restrict__ R1,
                                                           double* __restrict__ V, int64_t ldv, double* __restrict__ M,
                                                           double* __restrict__ Cb, double* __restrict__ Rtop, int64_t ldr,
                                                           PanelFlags* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* B0 = sm;                         // R2 -> (free)
  double* B1 = sm + 64 * LD;               // R2^-1
  double* B2 = sm + 2 * 64 * LD;           // R1 -> R2 R1
  double* B3 = sm + 3 * 64 * LD;           // Q1[0:64] -> Qtop
  double* cb = sm + 4 * 64 * LD;
  double* rab = cb + 128;
  double* piv = rab + 256;
  double* pv = piv + 64;
  double* sg = pv + 64;
  double* pb = sg + 64;
  const int tid = threadIdx.x, i = tid & 63, kq = tid >> 6;
  D64_STAMP(7);
  double a[16], b[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) a[m] = 0.0;
#pragma unroll 4
  for (int g = 0; g < G; ++g) {
    const double* __restrict__ pg = part + (size_t)g * 4096 + i + 64 * 16 * kq;
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] += pg[64 * m];
  }
  int far = 0;
#pragma unroll
  for (int m = 0; m < 16; ++m)
    if (!(fabs(a[m] - (16 * kq + m == i ? 1.0 : 0.0)) <= 0.5)) far = 1;   // the first pass left Q1 far from orthonormal (or NaN)
  for (int e = tid; e < 4096; e += 256) {
    B2[(e & 63) * LD + (e >> 6)] = R1[e];
    B3[(e & 63) * LD + (e >> 6)] = V[(e & 63) + (int64_t)(e >> 6) * ldv];
  }
  D64_STAMP(8);
  // The second Gram matrix is I + E with |E| ~ eps cond(P)^2: when every entry of E is below 2^-24 the factor and its inverse
  // come from the series around the identity -- four 64^3 products on the matrix pipe instead of a 64-step dependent chain
  // (40 of this kernel's 100 us).  R2 = I + U, U upper triangular:  U + U' = E - U'U,  solved by  U <- Phi(E - U'U)  from
  // U = Phi(E)  (Phi: strict upper triangle + half the diagonal); two sweeps leave |E|^3 <= (64 * 2^-24)^3 = 6e-17.
  // R2^-1 = (I - U)(I + U^2) = I - U + U^2 - U^3  (the next term is |U|^4 ~ 2e-22).  Anything larger takes the chain below.
  int big = 0;
#pragma unroll
  for (int m = 0; m < 16; ++m)
    if (!(fabs(a[m] - (16 * kq + m == i ? 1.0 : 0.0)) <= 0x1p-24)) big = 1;
#ifdef MMG_HEAD2_NO_SERIES
  big = 1;                                                     // (A/B builds of tools/probe/head_probe.hip)
#endif
  big = __syncthreads_or(big);                                 // (also orders the B2 / B3 fills above before their readers)
  int bad = 0;
  static_assert(sizeof(v4d) == 32, "v4d");
  int nottiny = 0;                                             // |E| <= 2^-33: |E|_2^2 <= (64 * 2^-33)^2 = 6e-17 -- the first-order factor is exact
#pragma unroll
  for (int m = 0; m < 16; ++m)
    if (!(fabs(a[m] - (16 * kq + m == i ? 1.0 : 0.0)) <= 0x1p-33)) nottiny = 1;
#ifdef MMG_HEAD2_NO_SERIES
  nottiny = 1;
#endif
  nottiny = __syncthreads_or(nottiny);
  if (!nottiny) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = 16 * kq + m;
      const double e = a[m] - (k == i ? 1.0 : 0.0);
      const double u = k > i ? e : (k == i ? 0.5 * e : 0.0);
      B0[i * LD + k] = (k == i ? 1.0 : 0.0) + u;               // R2 = I + Phi(E)
      B1[i * LD + k] = (k == i ? 1.0 : 0.0) - u;               // R2^-1 = I - Phi(E)  (+ O(|E|^2))
    }
    if (tid == 0) { flags->series += 1; flags->tiny += 1; }
  } else if (!big) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = 16 * kq + m;
      const double e = a[m] - (k == i ? 1.0 : 0.0);
      B1[i * LD + k] = e;
      B0[i * LD + k] = k > i ? e : (k == i ? 0.5 * e : 0.0);
    }
    __syncthreads();
    v4d t[4];
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      mm64_mfma(tid, [&](int r, int k) { return B0[k * LD + r]; }, [&](int k, int c) { return B0[k * LD + c]; }, t);   // U'U
      __syncthreads();
      mm64_store(tid, t, [&](int r, int c, double v) {
        const double x = B1[r * LD + c] - v;
        B0[r * LD + c] = c > r ? x : (c == r ? 0.5 * x : 0.0);
      });
      __syncthreads();
    }
    mm64_mfma(tid, [&](int r, int k) { return B0[r * LD + k]; }, [&](int k, int c) { return B0[k * LD + c]; }, t);     // U^2
    mm64_store(tid, t, [&](int r, int c, double v) { B1[r * LD + c] = v; });                                           // (E is dead)
    __syncthreads();
    mm64_mfma(tid, [&](int r, int k) { return (r == k ? 1.0 : 0.0) - B0[r * LD + k]; },
              [&](int k, int c) { return (k == c ? 1.0 : 0.0) + B1[k * LD + c]; }, t);
    __syncthreads();
    mm64_store(tid, t, [&](int r, int c, double v) { B1[r * LD + c] = c >= r ? v : 0.0; });                           // R2^-1 (upper)
    if ((i >> 4) == kq) B0[i * LD + i] += 1.0;                                                                         // R2 = I + U
    if (tid == 0) flags->series += 1;
  } else {
    bad = chol64_inv_reg(a, b, cb, rab, pb, piv, tid);
    const double rsi = 1.0 / sqrt(piv[i]);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = 16 * kq + m;                               // L2[i][k] = R2[k][i];  L2^-1[i][k] = R2^-1[k][i]
      B0[k * LD + i] = k < i ? a[m] / sqrt(piv[k]) : (k == i ? sqrt(piv[i]) : 0.0);
      B1[k * LD + i] = k <= i ? b[m] * rsi : 0.0;
    }
  }
  D64_STAMP(9);
  far = __syncthreads_or(far);
  v4d rr[4], qt[4];
  mm64_mfma(tid, [&](int r, int k) { return B0[r * LD + k]; }, [&](int k, int c) { return B2[k * LD + c]; }, rr);   // R2 R1
  mm64_mfma(tid, [&](int r, int k) { return B3[r * LD + k]; }, [&](int k, int c) { return B1[k * LD + c]; }, qt);   // Qtop
  __syncthreads();
  mm64_store(tid, rr, [&](int r, int c, double v) { B2[r * LD + c] = v; });
  mm64_store(tid, qt, [&](int r, int c, double v) { B3[r * LD + c] = v; });
  __syncthreads();
  D64_STAMP(10);
  if constexpr (BLK) {
    // (Qtop - S)^-1 through the blocked LU: the three matrices the outputs need wait in registers while the LDS images are reused
    double q[16], r2i[16], rr1[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = 16 * kq + m;
      q[m] = B3[i * LD + k]; r2i[m] = B1[i * LD + k]; rr1[m] = B2[i * LD + k];
    }
    __syncthreads();
    double *Bq = B3, *Fq = B0, *Xs = B1, *Li = B2;
    lu64_signed_lds(Bq, Fq, pv, sg, tid);                       // Qtop - S = Lq Uq
    unit_lower_inverse64(Fq, Xs, Li, tid);                      // Li = Lq^-1
    // Uq = Dp (I + G): Gt = G' (strictly lower) -> Fq, W = (I + Gt)^-1 -> Bq;  Uq^-1[r][c] = W[c][r] / p_c
    if (tid < 64) cb[tid] = 1.0 / pv[tid];                      // (cb: 128 doubles of chain scratch, free here)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = 16 * kq + m;
      Fq[i * LD + k] = i > k ? Bq[k * LD + i] * cb[k] : 0.0;
    }
    __syncthreads();
    unit_lower_inverse64(Fq, Xs, Bq, tid);
    D64_STAMP(11);
    // M[k][i] = -N[i][k] s_i,  N = Uq^-1 Lq^-1:  M[k][i] = -s_i sum_m Li[m][k] W[m][i] / p_m  (m >= max(k, i))
    {
      const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
      for (int t = w; t < 16; t += 4) {
        const int kb_ = t >> 2, ib = t & 3;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int mb = kb_ > ib ? kb_ : ib; mb < 4; ++mb)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int mm = 16 * mb + 4 * ks + lk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[mm * LD + 16 * kb_ + lr], Bq[mm * LD + 16 * ib + lr] * cb[mm], acc, 0, 0, 0);
          }
        const double si = -sg[16 * ib + lr];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) M[(16 * kb_ + 4 * rr + lk) + 64 * (16 * ib + lr)] = acc[rr] * si;
      }
    }
    if (!(fabs(pv[i]) >= 1.0)) far = 1;                          // (NaN-aware: a panel that lost its numbers is flagged below)
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = 16 * kq + m;
      V[i + (int64_t)k * ldv] = (k == i ? 1.0 : 0.0) - q[m] * sg[k];             // V[0:64] = I - Qtop S
      Cb[i + 64 * k] = -r2i[m] * sg[k];                                          // Cb = -R2^-1 S
      if (i <= k) Rtop[i + (int64_t)k * ldr] = sg[i] * rr1[m];                  // upper triangle of S R2 R1
    }
  } else {
#pragma unroll
  for (int m = 0; m < 16; ++m) a[m] = B3[i * LD + 16 * kq + m];
  gj64_signed_reg(a, b, cb, rab, pb, pv, sg, tid);
  D64_STAMP(11);
  {
    // N = (Qtop - S)^-1 = b / pv (row i);  M = (I - Qtop S)^-T = -N' S:  M[k][i] = -N[i][k] s_i
    const double f = -sg[i] / pv[i];
    const bool sane = fabs(pv[i]) >= 1.0;                      // (NaN-aware: a panel that lost its numbers is flagged below)
    if (!sane) far = 1;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = 16 * kq + m;
      M[k + 64 * i] = b[m] * f;
      V[i + (int64_t)k * ldv] = (k == i ? 1.0 : 0.0) - B3[i * LD + k] * sg[k];   // V[0:64] = I - Qtop S
      Cb[i + 64 * k] = -B1[i * LD + k] * sg[k];                                  // Cb = -R2^-1 S
      if (i <= k) Rtop[i + (int64_t)k * ldr] = sg[i] * B2[i * LD + k];          // upper triangle of S R2 R1
    }
  }
  }
  far = __syncthreads_or(far);
  if (tid == 0) {
    if (bad || far) flags->bad = 1;
    flags->panels += 1;
  }
  D64_STAMP(12);
}

void launch_cholqr_head2(hipStream_t st, const double* part, int G, const double* R1, double* V, int64_t ldv, double* M, double* Cb,
                         double* Rtop, int64_t ldr, PanelFlags* flags) {
  if (head_blocked()) hipLaunchKernelGGL(cholqr_head2_kernel<true>, dim3(1), dim3(256), HEAD2_LDS, st, part, G, R1, V, ldv, M, Cb, Rtop, ldr, flags);
  else hipLaunchKernelGGL(cholqr_head2_kernel<false>, dim3(1), dim3(256), HEAD2_LDS, st, part, G, R1, V, ldv, M, Cb, Rtop, ldr, flags);
}

// ---- coefficients of the two-sided update ------------------------------------------------------------------------------------
constexpr int COEF_LDS = 3 * 64 * LD * (int)sizeof(double);
__global__ __launch_bounds__(256) void band_coef_kernel(const double* __restrict__ part, int G, const double* __restrict__ partz,
                                                        int Gz, int q1, const double* __restrict__ M, double* __restrict__ C,
                                                        double* __restrict__ Cz) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Ws = sm;                         // V'W, then V'Z
  double* Ms = sm + 64 * LD;               // M
  double* Xs = sm + 2 * 64 * LD;           // (V'W) M
  const int tid = threadIdx.x;
  {
    double acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = 0.0;
#pragma unroll 4
    for (int g = 0; g < G; ++g) {
      const double* __restrict__ pg = part + (size_t)g * 4096 + tid;
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m] += pg[256 * m];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int e = tid + 256 * m;
      Ws[(e & 63) * LD + (e >> 6)] = acc[m];
      Ms[(e & 63) * LD + (e >> 6)] = M[e];
    }
  }
  __syncthreads();
  v4d acc[4];
  mm64_mfma(tid, [&](int r, int k) { return Ws[r * LD + k]; }, [&](int k, int c) { return Ms[k * LD + c]; }, acc);
  mm64_store(tid, acc, [&](int r, int c, double v) { Xs[r * LD + c] = v; });
  __syncthreads();
  mm64_mfma(tid, [&](int r, int k) { return Ms[k * LD + r]; }, [&](int k, int c) { return Xs[k * LD + c]; }, acc);   // M'(V'W)M
  mm64_store(tid, acc, [&](int r, int c, double v) { C[r + 64 * c] = -0.5 * v; });
  for (int e = tid; e < 64 * q1; e += 256) {
    double s = 0.0;
    for (int g = 0; g < Gz; ++g) s += partz[(size_t)g * 4096 + e];
    Ws[(e & 63) * LD + (e >> 6)] = s;      // (nobody reads Ws any more: the second product uses Ms and Xs)
  }
  __syncthreads();
  for (int e = tid; e < 64 * q1; e += 256) {
    const int i = e & 63, c = e >> 6;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < 64; k += 2) {
      s0 = fma(Ms[k * LD + i], Ws[k * LD + c], s0);
      s1 = fma(Ms[(k + 1) * LD + i], Ws[(k + 1) * LD + c], s1);
    }
    Cz[i + 64 * c] = s0 + s1;              // M'(V'Z)
  }
}

void launch_band_coef(hipStream_t st, const double* part, int G, const double* partz, int Gz, int q1, const double* M, double* C,
                      double* Cz) {
  hipLaunchKernelGGL(band_coef_kernel, dim3(1), dim3(256), COEF_LDS, st, part, G, partz, Gz, q1, M, C, Cz);
}

// ---- diagonal block of the blocked Cholesky ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void potrf_head_kernel(double* __restrict__ A, int64_t lda, int kb, double* __restrict__ LinvT,
                                                         long long* __restrict__ info, long long base) {
  __shared__ __attribute__((aligned(16))) double cb[2 * 64], rb[2 * 64], pb[6];
  __shared__ double piv[64];
  const int tid = threadIdx.x, i = tid & 63, kq = tid >> 6;
  double a[16], b[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = 16 * kq + m;
    a[m] = (i < kb && k < kb) ? (i >= k ? A[i + (int64_t)k * lda] : 0.0) : (i == k ? 1.0 : 0.0);
  }
  const int bad = chol64_inv_reg(a, b, cb, rb, pb, piv, tid);
  const double rsi = 1.0 / sqrt(piv[i]);
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = 16 * kq + m;                                 // L[i][k], i >= k
    const double v = k < i ? a[m] / sqrt(piv[k]) : (k == i ? sqrt(piv[i]) : 0.0);
    if (i >= k && i < kb) A[i + (int64_t)k * lda] = v;
    LinvT[k + 64 * i] = k <= i ? b[m] * rsi : 0.0;             // (L^-T)[k][i] = (L^-1)[i][k]
  }
  if (tid == 0 && bad && *info == 0) *info = base + bad;
}

__global__ __launch_bounds__(256) void potrf_head_blk_kernel(double* __restrict__ A, int64_t lda, int kb, double* __restrict__ LinvT,
                                                             long long* __restrict__ info, long long base) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *As = sm, *Fs = sm + 64 * LD, *Xs = sm + 2 * 64 * LD, *T1 = sm + 3 * 64 * LD, *rs = sm + 4 * 64 * LD;
  int* sh = (int*)(rs + 64);
  const int tid = threadIdx.x, i = tid & 63, kq = tid >> 6;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = 16 * kq + m;
    As[i * LD + k] = (i < kb && k < kb) ? (i >= k ? A[i + (int64_t)k * lda] : 0.0) : (i == k ? 1.0 : 0.0);
  }
  __syncthreads();
  const int bad = chol64_lds(As, Fs, Xs, T1, rs, sh, tid);
  const double rsi = rs[i];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = 16 * kq + m;                                 // L[i][k], i >= k
    if (i >= k && i < kb) A[i + (int64_t)k * lda] = As[i * LD + k] * rs[k];
    LinvT[k + 64 * i] = T1[i * LD + k] * rsi;                  // (L^-T)[k][i] = (L^-1)[i][k]
  }
  if (tid == 0 && bad && *info == 0) *info = base + bad;
}

void launch_potrf_head(hipStream_t st, double* A, int64_t lda, int kb, double* LinvT, long long* info, long long base) {
  if (head_blocked()) hipLaunchKernelGGL(potrf_head_blk_kernel, dim3(1), dim3(256), CHOL_LDS, st, A, lda, kb, LinvT, info, base);
  else hipLaunchKernelGGL(potrf_head_kernel, dim3(1), dim3(256), 0, st, A, lda, kb, LinvT, info, base);
}

int dense64_init() {
  // once per DEVICE (the attribute belongs to the device the calling context is bound to; a process may hold contexts on several)
  static int done[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (done[dev]) return done[dev] - 1;
  const int rc = [] {
    hipError_t e = hipFuncSetAttribute((const void*)cholqr_head2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HEAD2_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)cholqr_head2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, HEAD2_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)cholqr_head1_blk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CHOL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)potrf_head_blk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CHOL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)band_coef_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, COEF_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rows_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS2);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rows_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS1);
    return e == hipSuccess ? 0 : 1;
  }();
  done[dev] = rc + 1;
  return rc;
}

}  // namespace mmg
