// dense64.h -- fp64 building blocks on 64-column panels, shared by the band reduction of K (reml_band.hip) and the blocked
// Cholesky factorisation behind the scan model (reml_chol.hip).  Everything here replaces a rocSOLVER / rocBLAS call that was
// spending its time on launch latency, not flops, at N = 5000 (linear_models.py:589-615, 771-927 are what these stages
// compute for): a tall-skinny QR as two Cholesky-QR passes + a basis-kernel representation of the orthogonal factor (8
// launches per panel instead of 67), tall-times-64x64 products and the symmetric rank-k update of the lower triangle on
// v_mfma_f64_16x16x4_f64.
// Definitions: dense64.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmg {

constexpr int D64 = 64;
constexpr int D64_TS_ROWS = 32;            // row chunk of the tall-skinny Gram kernels
constexpr int D64_MAX_SLICES = 40;         // partial Gram matrices a head kernel sums itself

// Device scalars of a panel factorisation: flags[0] != 0 when a Cholesky-QR pass met a non-positive pivot or a first-pass Q
// that is far from orthonormal (the panel is numerically rank deficient: the caller redoes the reduction with
// Householder panels); flags[1] counts the panels done.
// flags[2] counts the panels whose second Cholesky factor came from the series around the identity (cholqr_head2_kernel).
// flags[3]: ... of those, the panels whose |E| was so small (<= 2^-33) that R2 = I + Phi(E), R2^-1 = I - Phi(E) needed no product.
struct PanelFlags { int bad; int panels; int series; int tiny; };

// part[g] (64 x 64 column-major each) = A[rows of slice g]' B[rows of slice g], g < *G_out slices of `rows_per` rows;
// A [n x 64] (ld lda), B [n x kb] (ld ldb), kb <= 64.  Returns the slice count the launch used.
int launch_gram_slices(hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb, int kb, int64_t n,
                       double* part, int max_slices);
// the same for two right-hand operands in ONE launch (blockIdx.y): part0 = A'B0, part1 = A'B1
int launch_gram_slices2(hipStream_t st, const double* A, int64_t lda, int64_t n, const double* B0, int64_t ldb0, int kb0,
                        double* part0, const double* B1, int64_t ldb1, int kb1, double* part1, int max_slices);
// out (64 x 64, ld ldo) = sum_g part[g], fixed order
void launch_gram_reduce(hipStream_t st, const double* part, int G, double* out, int ldo);

// Y = X Cf for the n rows of X ([n x 64] column-major, ld ldx) and a 64 x 64 column-major coefficient matrix (device, ld 64) on
// the matrix pipe; Y (ld ldy) may be X
void launch_rows_gemm(hipStream_t st, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t n, const double* Cf,
                      bool cf_transposed = false);    // cf_transposed: Y = X Cf' 

// ... and, in the same launch, the Gram matrix of every workgroup's 64 rows of Y as slice g of `gram` (ceil(n / 64) slices of 4096
// doubles, the layout launch_gram_slices writes; their count is returned): Y'Y without a second pass over Y
int launch_rows_gemm_gram(hipStream_t st, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t n, const double* Cf,
                          double* gram);
// two reductions (ld 64 outputs) in one launch
void launch_gram_reduce2(hipStream_t st, const double* part0, const double* part1, int G, double* out0, double* out1);

// Lower 64 x 64 tiles (I >= J) of the n x n matrix C (column-major, ld ldc):
//   C[I][J] -= A0[I] B0[J]' (+ A1[I] B1[J]' when A1 != nullptr),   A*, B*: [n x 64] column-major (ld lda / ldb)
// gram_col0 != nullptr: the 64-row tiles (I >= 1, 0) also leave the Gram matrix of their updated tile as slice I - 1 there (the
// next panel's P'P for a blocked factorisation); returns the slice count written -- 0 when the launch took the 128 x 128 tiles
// (n >= 4096), which do not do this.
int launch_nt_update_lower(hipStream_t st, double* C, int64_t ldc, int64_t n, const double* A0, const double* B0,
                           const double* A1, const double* B1, int64_t lda, int64_t ldb, double* gram_col0 = nullptr);

// ... the tiles (I, 0) of the first 64-column block column only
void launch_nt_update_col0(hipStream_t st, double* C, int64_t ldc, int64_t n, const double* A0, const double* B0,
                           const double* A1, const double* B1, int64_t lda, int64_t ldb);

// First pass of Cholesky-QR: G = sum of the Gram slices -> R1 = chol(G)' and R1^-1 (64 x 64 column-major upper, ld 64)
void launch_cholqr_head1(hipStream_t st, const double* part, int G, double* R1, double* R1inv, PanelFlags* flags);
// Second pass + an orthogonal H with H'P = [R; 0] in basis-kernel form (Yamamoto 2012; signs as in Ballard et al., "Reconstructing
// Householder vectors from TSQR", 2014):  G2 = sum of the slices of Q1'Q1, R2 = chol(G2)', Qtop = Q1[0:64] R2^-1,
//   H = I - V M V',   V = [I; 0] - Q S,   M = (I - Qtop S)^-T,   S_jj = +-1 chosen during the elimination (|pivot| >= 1).
// Writes V[0:64] = I - Qtop S (over Q1[0:64]), M, Cb = -R2^-1 S (so that V[64:] = Q1[64:] Cb) and the upper triangle of
// R = S R2 R1 into Rtop (ld ldr).
void launch_cholqr_head2(hipStream_t st, const double* part, int G, const double* R1, double* V, int64_t ldv, double* M, double* Cb,
                         double* Rtop, int64_t ldr, PanelFlags* flags);

// C (64 x 64, ld 64) = -1/2 M'(V'W) M with V'W = sum of `G` Gram slices in part; Cz (64 x q1, ld 64) = M'(V'Z), V'Z = the sum
// of `Gz` slices in partz
void launch_band_coef(hipStream_t st, const double* part, int G, const double* partz, int Gz, int q1, const double* M, double* C,
                      double* Cz);
// Y = W M + V C  ([n x 64], ld n each);  Z[r][c] -= sum_k V[r][k] Cz[k][c]  (Z: [n x q1] ld ldz)
void launch_band_y(hipStream_t st, const double* V, const double* W, int64_t n, const double* M, const double* C, double* Y,
                   double* Z, int64_t ldz, const double* Cz, int q1);

// In-place Cholesky factor of the kb x kb (kb <= 64) diagonal block at A (column-major lower, ld lda); LinvT (64 x 64, ld 64) =
// the inverse of the factor, transposed (upper), padded with the identity -- the coefficient of launch_rows_gemm for the
// rows below (X L' = A  <=>  X = A L^-T).
// *info (device) = base + 1 + index of the first non-positive pivot if there is one and *info was still 0.
void launch_potrf_head(hipStream_t st, double* A, int64_t lda, int kb, double* LinvT, long long* info, long long base);

int dense64_init();   // kernel attributes (large dynamic LDS); 0 on success

}  // namespace mmg
