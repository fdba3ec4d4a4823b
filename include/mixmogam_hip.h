/* mixmogam_hip.h -- C ABI of libmixmogam_hip.so, the MI355X (gfx950) implementation of the
 * mixmogam EMMAX hot path.
 *
 * The reference (bvilhjal/mixmogam) is pure Python and has NO FFI layer; its boundary for this
 * path is its Python call surface.  Each entry point below names the reference interface whose
 * arithmetic it replaces (file:line into /root/reference).  The reference-side binding a
 * maintainer would add (a ctypes stub inside kinship.py / linear_models.py) is shown in
 * INTEGRATION.md; the build's own host mirror of that call surface is mixmogam_amd/.
 *
 * Conventions
 *   - every call returns int: 0 = ok, <0 = error (MMG_E_*); text via mmg_last_error(ctx).
 *   - the caller owns every host buffer; the library never frees or retains one past return.
 *   - the library owns device memory inside the opaque ctx / geno / model handles.
 *   - calls are blocking unless named *_async; a ctx is single-threaded (one ctx per device
 *     per host thread); no callbacks into the host language.
 *   - genotypes are SNP-major: int8 [M x N] C-contiguous, M SNPs, N individuals
 *     (simulations.py:21-23, snpsdata.py:2579-2597).
 *   - no torch / numpy types in any signature: plain pointers and sizes.
 */
#ifndef MIXMOGAM_HIP_H
#define MIXMOGAM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMG_OK 0
#define MMG_E_ARG (-1)      /* bad argument / shape */
#define MMG_E_HIP (-2)      /* HIP runtime error */
#define MMG_E_STATE (-3)    /* call order (e.g. scan before model) */
#define MMG_E_LIB (-4)      /* rocSOLVER / rocBLAS / RCCL error */
#define MMG_E_NOMEM (-5)

typedef struct mmg_ctx mmg_ctx;
typedef struct mmg_geno mmg_geno;    /* device-resident padded genotype store */
typedef struct mmg_comm mmg_comm;    /* RCCL communicator, one rank per process */
typedef struct mmg_kin_acc mmg_kin_acc;  /* device-resident N x N kinship accumulator */
typedef struct mmg_rot mmg_rot;      /* eigen-rotated genotype store (multi-phenotype scans) */
typedef struct mmg_perm_plan mmg_perm_plan;  /* SNP-independent half of the permutation test, built once per (H, Ys) */
typedef struct mmg_reml mmg_reml;    /* eigendecomposition-free REML workspace (K, X, y resident) */

/* ---- library / context -------------------------------------------------------------- */
int mmg_version(void);
/* 1 if the library was built with `make EXPERIMENTS=1` (superseded scan-GEMM generations selectable through
 * MMG_SCAN_KERNEL for A/B runs); the shipped library returns 0 and ignores that variable. */
int mmg_has_experiments(void);
int mmg_device_count(int* n);
int mmg_ctx_create(int device, mmg_ctx** ctx);
int mmg_ctx_destroy(mmg_ctx* ctx);
/* Free the buffers the context keeps BETWEEN calls for speed (the band-route factor stores of the REML search, up to 2 GB each;
 * ~650 MB each after an emmax() call at N = 5000).  They are re-allocated on demand; factors kept by mmg_reml_band_factor are
 * forgotten (the next sums() call of their workspace factors again). */
int mmg_ctx_trim(mmg_ctx* ctx);
const char* mmg_last_error(mmg_ctx* ctx);    /* ctx may be NULL: last global error */
int mmg_device_info(mmg_ctx* ctx, char* name, int name_len, int* n_cu, int64_t* hbm_bytes);
/* PCI bus id of the context's device ("0000:c1:00.0"): what a multi-process launcher compares across ranks to be sure that no
 * two of them sit on the same GPU. */
int mmg_device_pci_bus_id(mmg_ctx* ctx, char* buf, int buf_len);
/* milliseconds the dominant kernel of the last call took, from hipEvents recorded on the
 * ctx stream around it.  which: 0 = kinship GEMM, 1 = scan quadratic-form GEMM,
 * 2 = scan finalize (per-SNP dot + F + p), 3 = permutation GEMM, 4 = eigh, 5 = transpose/pack,
 * 7 = eigen-rotation GEMM (mmg_rot_load), 8 = multi-phenotype passes (mmg_emmax_scan_multi, summed over batches),
 * 9 = digit-plane GEMMs of the last mmg_kin_acc_add_grm (summed) */
int mmg_last_kernel_ms(mmg_ctx* ctx, int which, double* ms);

/* Page-lock (unlock) a caller-owned host buffer so that result fetches into it run at full PCIe
 * rate; optional -- every entry point also accepts pageable memory. */
int mmg_host_pin(mmg_ctx* ctx, void* p, int64_t bytes);
int mmg_host_unpin(mmg_ctx* ctx, void* p);   /* MUST precede freeing a pinned buffer */
/* Page-locked host memory owned by the library until mmg_host_free. */
int mmg_host_alloc(mmg_ctx* ctx, int64_t bytes, void** p);
int mmg_host_free(mmg_ctx* ctx, void* p);

/* ---- genotype store ----------------------------------------------------------------- */
/* Allocates an [Mpad x Npad] zero-filled int8 store (Npad = N rounded up to 256, Mpad = M
 * rounded up to 256) in HBM. */
int mmg_geno_create(mmg_ctx* ctx, int64_t M, int32_t N, mmg_geno** g);
int mmg_geno_destroy(mmg_ctx* ctx, mmg_geno* g);
/* Reuse a store for another block of M <= (M at creation) SNPs without freeing / allocating HBM (both synchronise
 * the device): the chunk loop of hdf5_data.py:99-106,153-184 ping-pongs two stores.  Rows are rewritten by the
 * following upload; the padding rows are zeroed here. */
int mmg_geno_reset(mmg_ctx* ctx, mmg_geno* g, int64_t M);
/* Copy SNP rows [m0, m0+rows) from a host [rows x N] int8 C-contiguous buffer. */
int mmg_geno_upload(mmg_ctx* ctx, mmg_geno* g, const int8_t* snps, int64_t m0, int64_t rows);
/* Same from a host float32 / float64 [rows x N] buffer holding small integers (the C3 config's
 * "fp32 genotypes", hdf5_data.py:294 float64 copies), converted to int8 on the device.  Values that are not
 * integers in [-127, 127] (dosages, normalised SNPs, NaN) make the call fail with MMG_E_ARG -- the reference
 * takes arbitrary numeric SNPs (linear_models.py:1317), this int8 store does not, and says so. */
int mmg_geno_upload_f32(mmg_ctx* ctx, mmg_geno* g, const float* snps, int64_t m0, int64_t rows);
int mmg_geno_upload_f64(mmg_ctx* ctx, mmg_geno* g, const double* snps, int64_t m0, int64_t rows);
/* Packed rows: bits = 1 (genotype i of a row = bit i & 7 of byte i >> 3) or 2 (bits 2(i & 3).. of byte i >> 2) --
 * least significant first, the order of numpy.packbits(bitorder='little') and of a PLINK .bed row; row_bytes >=
 * ceil(N * bits / 8) is the stride of the host rows.  lut: int8 value of each of the 2^bits codes, or NULL for
 * code = value (0/1; 0/1/2/3 -- the 0/1/2 coding plink2hdf5.py:171-179 stores).  A .bed row (00 hom A1, 01 missing,
 * 10 het, 11 hom A2) uploads as is with lut = {0, <imputed value>, 1, 2}.  Expanded to the int8 store on the device:
 * the host link carries N/8 or N/4 bytes per SNP instead of N.  Not in the reference (it stores int8 with lzf,
 * plink2hdf5.py:111-118); added because host-resident runs are bound by that link. */
int mmg_geno_upload_packed(mmg_ctx* ctx, mmg_geno* g, const uint8_t* packed, int64_t m0, int64_t rows, int32_t bits,
                           int64_t row_bytes, const int8_t* lut);
int mmg_geno_download(mmg_ctx* ctx, mmg_geno* g, int8_t* snps, int64_t m0, int64_t rows);
/* Gather cnt rows idx[0..cnt) (host int64, any order) into a host [cnt x N] buffer: the top-hit rows of the
 * exact-EMMA refinement (linear_models.py:1365-1370) without moving the whole store over PCIe. */
int mmg_geno_download_rows(mmg_ctx* ctx, mmg_geno* g, const int64_t* idx, int64_t cnt, int8_t* snps);
/* Synthetic Bernoulli genotypes generated on the device (restates simulations.py:21-23 with a
 * counter-based hash so that any SNP range can be regenerated on any rank or on the CPU):
 * s[m][i] = hash64(seed, m_global0 + m, i) >> 48 < thr16 (thr16 = 32768 -> p = 0.5). */
int mmg_geno_fill_hash(mmg_ctx* ctx, mmg_geno* g, uint64_t seed, int64_t m_global0, uint32_t thr16);
/* Structured twin: `npop` contiguous populations (pop(i) = i*npop/N), per-SNP ancestral frequency U[0.1,0.9] plus a
 * per-population deviation of scale spread_q16/65536 (sum of four uniforms), clamped to [0.02,0.98]; 16-bit fixed
 * point from the same hash, reproducible on the host (the test tree holds a numpy twin).  Gives an interior REML
 * optimum and many correlated strong hits -- the data regime of real GWAS. */
int mmg_geno_fill_structured(mmg_ctx* ctx, mmg_geno* g, uint64_t seed, int64_t m_global0, int32_t npop,
                             uint32_t spread_q16);
/* Per-SNP mean and population std (kinship.py:66, hdf5_data.py:99-104). Host outputs, length M. */
int mmg_geno_snp_stats(mmg_ctx* ctx, mmg_geno* g, double* mean, double* std);

/* out[k][m] = s_m . V[k] for nv host fp64 vectors V [nv x N] (row-major); out host [nv x M].
 * The per-SNP linear functionals of the scan: X_j = H s regressors of linear_models.py:1323
 * (with_betas) reduce to such dot products. */
int mmg_geno_matvec(mmg_ctx* ctx, mmg_geno* g, const double* V, int32_t nv, double* out);

/* ---- kinship (replaces the GEMMs of kinship.py:29-44 and kinship.py:63-69) ----------------- */
/* IBS counts C = sum_m (2 s_m - 1)(2 s_m - 1)^T as an exact int8-MFMA GEMM (int32 accumulate).
 * C_out: host int64 [N x N].  Bit-exact with kinship.py:43-44 (whose entries are exact
 * integers carried in float64). */
int mmg_kinship_ibs_i8(mmg_ctx* ctx, mmg_geno* g, int64_t* C_out);
/* The IBS kinship itself (kinship.py:44-51): K = counts / (2 m_total) + 0.5 from the exact counts, scaled != 0: times
 * (N - 1) / (tr K - sum K / N) (scale_k, :94-100), converted and scaled in HBM -- one download of N^2 doubles instead of the
 * int64 counts and three host passes over the matrix (50 of 92 ms of calc_ibs_kinship at N = 5000).  comm != NULL: the counts
 * of all ranks' SNP blocks (m_total = their SNPs together) are summed in HBM first.  The unscaled matrix is bit-identical to
 * the host expression; the scaled one differs from kinship.scale_k by summation order (1e-16). */
int mmg_kinship_ibs_f64(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int64_t m_total, int32_t scaled, double* K_out);
/* The same for 0/1/2 genotypes ('diploid_int', kinship.py:33-41,51): k_ij = (M - 1/2 sum_m |a_m - b_m|) / M off the diagonal,
 * 1 on it, from two exact indicator products ([s >= 1], [s >= 2]: |a - b| = a + b - 2 min(a, b)), combined and -- scaled != 0 --
 * scale_k'd in HBM. */
int mmg_kinship_ibs_diploid_f64(mmg_ctx* ctx, mmg_geno* g, int32_t scaled, double* K_out);
/* Indicator co-occurrence counts C = U U^T, U = [s >= thr] (exact, same kernel).  Two calls
 * (thr = 1, 2) give the 'diploid_int' IBS kinship of kinship.py:33-41:
 * sum_m |a_m - b_m| = r_a + r_b - 2 (C1_ab + C2_ab), r = diag(C1 + C2). */
int mmg_kinship_indicator_i8(mmg_ctx* ctx, mmg_geno* g, int32_t thr, int64_t* C_out);
/* General per-SNP affine kinship C = sum_m x_m x_m^T, x_m = scale[m]*s_m + shift[m], as a dense
 * fp32 MFMA GEMM with the int8 genotypes expanded to fp32 from the LDS tile.  scale/shift: host
 * float arrays of length M, or NULL for scale=2, shift=-1 (the IBS expansion of kinship.py:43;
 * exact while M < 2^24).  GRM (kinship.py:66): scale = 1/std, shift = -mean/std.
 * C_out: host double [N x N] (fp32 partial tiles per K-split, summed in fp64 in fixed order). */
int mmg_kinship_affine_f32(mmg_ctx* ctx, mmg_geno* g, const float* scale, const float* shift,
                           double* C_out);
/* Chunked / streamed genotypes: the `k_mat += x'x` loop of hdf5_data.py:99-106 / kinship.py:63-69 with the
 * N x N sum kept in HBM between chunks.  add: acc += sum_m x_m x_m' over the SNPs of g (same affine map as
 * mmg_kinship_affine_f32; NULL/NULL = 2s-1); fetch: host double [N x N] and the SNP count so far. */
int mmg_kin_acc_create(mmg_ctx* ctx, int32_t N, mmg_kin_acc** acc);
int mmg_kin_acc_add(mmg_ctx* ctx, mmg_kin_acc* acc, mmg_geno* g, const float* scale, const float* shift);
/* GRM chunk (kinship.py:66-69, hdf5_data.py:99-106): acc += sum_m z_m z_m' with z = (s - mean)/std computed per SNP on
 * the device in fp64 -- EXACT route: z z' = a^2 s s' + a b (s 1' + 1 s') + b^2 1 1'; the weighted Gram matrix
 * sum_m s s'/std^2 is 4-5 int8-MFMA GEMMs of the IBS kind (the weight 1/std^2 split into non-negative digits folded into
 * one operand), the rank-one terms fp64 dot products.  ~5x faster than the fp32-MFMA kernel; falls back to that kernel for
 * genotype alphabets beyond -4..4.  A SNP with std == 0 is an error (kinship.py:67).
 * Precision: every weight is rounded to 2^-35 of the call's LARGEST weight for 0/1 stores and for 0/1/2 stores (five
 * planes of 7 bits; a 0/1/2 store is read as s - 1 in {-1, 0, 1}, which leaves z unchanged), 2^-30 for wider alphabets;
 * entries good to ~1e-9 of the float64 result.  0/1 and 0/1/2 stores of >= 65,536 SNPs whose weights span less than a
 * factor 64 (any MAF filter >= 0.004) take FOUR planes, 2^-28 of the largest weight -- the per-SNP roundings are
 * independent and average down as 1/sqrt(M).  The digits are relative to the largest weight of a RUN of calls (below), so
 * the sum depends (at that level, ~1e-10) on how the SNPs are grouped into calls: a multi-rank or differently chunked run
 * is equal to a single-rank one to rounding, not bit for bit (the IBS counts of mmg_kinship_ibs_i8 are).
 * One call may hold at most (2^31 - 1) / (127 smax^2) SNPs (16.9 M binary ones): the digit planes are 32-bit sums.
 * Runs (round 4): the int32 planes of a call stay where they are and the next call adds into them while its largest
 * weight is below the run's cap (the first call's largest weight + 1/16) and above half of it, it needs the same number
 * of planes, and the 32-bit bound holds for the run; otherwise -- and before fetch / scale_k / allreduce read the
 * accumulator -- the planes are combined into the fp64 sum (at N = 50,000 clearing and combining them is 33 ms next to
 * 208 ms of GEMMs per 65,536 SNPs).  A joining call loses at most one bit against a step of its own.  MMG_GRM_DEFER=0:
 * every call is a run of its own.  mmg_kin_acc_pending: SNPs whose sums are still in the planes (0: none).
 * Errors: a call rejected before it adds anything (a monomorphic SNP: MMG_E_ARG) leaves the accumulator as it was; a call that
 * fails AFTER it has begun to add (a launch or stream error) leaves it unusable -- every later add / fetch / scale_k /
 * allreduce / mmg_reml_create_from_acc on it returns MMG_E_STATE (round 5; before, the earlier calls' sums were dropped
 * silently while their SNPs stayed counted). */
int mmg_kin_acc_add_grm(mmg_ctx* ctx, mmg_kin_acc* acc, mmg_geno* g);
int mmg_kin_acc_pending(mmg_ctx* ctx, mmg_kin_acc* acc, int64_t* n_snps_pending);
/* The accumulator's matrix := the IBS kinship of a store (kinship.py:14-56: counts / (2 m_total) + 0.5, scale_k's rule when
 * scaled != 0), formed and KEPT in HBM: followed by mmg_reml_create_from_acc the kinship of an emmax() call never visits the
 * host.  comm / m_total: the SNP blocks of all ranks (RCCL SUM of the counts).  Replaces whatever the accumulator held. */
int mmg_kin_acc_set_ibs(mmg_ctx* ctx, mmg_comm* comm, mmg_kin_acc* acc, mmg_geno* g, int64_t m_total, int32_t scaled);
int mmg_kin_acc_snps(mmg_ctx* ctx, mmg_kin_acc* acc, int64_t* n_snps);      /* SNPs added so far (what fetch reports), without the download */
int mmg_kin_acc_fetch(mmg_ctx* ctx, mmg_kin_acc* acc, double* C_out, int64_t* n_snps);
/* scale_k of the reference (kinship.py:94-100, inlined at hdf5_data.py:108-111) on the device-resident matrix, in place:
 * K *= (N - 1) / (tr K - sum K / N).  The rule is invariant under a prior division of K by the SNP count, so the
 * accumulated sum can be scaled as it is.  *scalar_out = the factor.  (On the host the same takes three passes over a
 * 20 GB matrix at N = 50,000: ~10 s of a 45 s kinship pass.) */
int mmg_kin_acc_scale_k(mmg_ctx* ctx, mmg_kin_acc* acc, double* scalar_out);
int mmg_kin_acc_destroy(mmg_ctx* ctx, mmg_kin_acc* acc);
/* One-shot twin taking host genotypes: upload + mmg_kinship_affine_f32 / _ibs_i8. */
int mmg_kinship_i8(mmg_ctx* ctx, const int8_t* snps, int64_t M, int32_t N,
                   const float* scale, const float* shift, double* C_out);

/* float32-genotype twin (values must be small integers, see mmg_geno_upload_f32) */
int mmg_kinship_f32(mmg_ctx* ctx, const float* snps, int64_t M, int32_t N,
                    const float* scale, const float* shift, double* C_out);

/* ---- eigendecomposition (replaces scipy.linalg.eigh at linear_models.py:594,613) ---------- */
/* Symmetric eigendecomposition on the device (rocSOLVER dsyevd). A: host [N x N] (symmetric,
 * both triangles present).  evals ascending; evecs (may be NULL) host [N x N] row-major whose
 * ROWS are the eigenvectors -- the layout the reference keeps after its transpose at :596
 * (scipy's evecs.T). */
int mmg_eigh_f64(mmg_ctx* ctx, const double* A, int32_t N, double* evals, double* evecs);
/* C[MxN] = op(A) op(B) in fp64 on the device (rocBLAS dgemm; row-major host buffers); used for
 * the O(N^3) products of linear_models.py:610,898,1303.  ta/tb: 0 = as is, 1 = transposed. */
int mmg_dgemm_f64(mmg_ctx* ctx, int ta, int tb, int32_t M, int32_t N, int32_t K,
                  const double* A, const double* B, double* C);

/* ---- REML and the scan model without an eigendecomposition (large N: beyond rocSOLVER's syevd index range) ----
 * The reference evaluates the EMMA likelihood from eigh(K) and eigh(S(K+I)S) (linear_models.py:589-615,794-810);
 * what it consumes per variance ratio delta are four sums -- with H = K + delta I and
 * P = H^-1 - H^-1 X (X'H^-1 X)^-1 X'H^-1:  s1 = y'Py, s2 = log|H| + log|X'H^-1 X| - log|X'X|, s3 = |Py|^2,
 * s4 = tr P -- and sum_sq_etas = |Sy|^2.  mmg_reml_sums evaluates them for nd values of delta, by one of two routes
 * (mmg_reml_sums_ex names it; mmg_reml_sums = AUTO: the band route from N = 256 up, MMG_REML_ROUTE=chol|band overrides):
 *   CHOL  one Cholesky factorisation per delta (potrf_64 + recursive triangular inverse, 2 N^3/3 flops per delta;
 *         the values are independent, so ranks can share a grid);
 *   BAND  K is reduced ONCE to an orthogonally similar band matrix (bandwidth 64; 4 N^3/3 flops of level-3 BLAS, kept
 *         in the workspace), after which every delta costs O(64^2 N): banded Cholesky, banded solves, the band of the
 *         inverse for the trace -- all deltas of a call side by side on the device (csrc/reml_band.hip).
 * mmg_reml_scan_model builds the EMMAX scan model at delta straight on
 * the device -- A = Mp Mp' = P, w = Mp r = Py (linear_models.py:1290-1303 in closed form) -- and returns
 * h0_rss = y'Py (= the Mahalanobis RSS of the null model, :906) and the GLS estimate beta [q] (:902).
 * K: host [N x N] symmetric (scaled as the model holds it); X: host [N x q] row-major (intercept first), q <= 16. */
int mmg_reml_create(mmg_ctx* ctx, int32_t N, int32_t q, const double* K, const double* X, const double* y, mmg_reml** r);
/* The same workspace with K taken from a kinship accumulator as it lies in HBM (a device-to-device copy): the kinship of a
 * streamed pass (mmg_kin_acc_add_grm ... mmg_kin_acc_scale_k) goes into the likelihood search without visiting the host -- at
 * N = 50,000 the download, the host's scale_k and the upload of 20 GB each were ~2 s of a 7.5 s REML stage. */
int mmg_reml_create_from_acc(mmg_ctx* ctx, mmg_kin_acc* acc, int32_t q, const double* X, const double* y, mmg_reml** r);
int mmg_reml_destroy(mmg_ctx* ctx, mmg_reml* r);
int mmg_reml_sums(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s2, double* s3,
                  double* s4, double* sum_sq_etas);
#define MMG_REML_ROUTE_AUTO 0
#define MMG_REML_ROUTE_CHOL 1
#define MMG_REML_ROUTE_BAND 2
int mmg_reml_sums_ex(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s2, double* s3,
                     double* s4, double* sum_sq_etas, int32_t route);
/* Band route, round 5: factor B + delta I for nd variance ratios at once and KEEP the banded factors (at most 256 of them and
 * 2 GB, in the context; beyond that the call keeps nothing and still succeeds).  A later mmg_reml_sums* call on the band route
 * whose variance ratios are ALL among the kept ones (same bits) skips its factor sweep and costs the substitutions and the trace
 * recurrence only.  What this serves: the likelihood search of get_estimates (linear_models.py:814-847) -- its grid is factored
 * together with a four-fold refinement of itself (one workgroup per variance ratio: 4 ms at N = 5000 for 51 or for 227), the
 * sums are taken on the grid, and once the grid has bracketed the optimum the nodes around the bracket are evaluated from the
 * kept factors.  Kept factors belong to ONE workspace per context: the next mmg_reml_band_factor call replaces them.
 * MMG_E_LIB "not positive definite" as mmg_reml_sums. */
int mmg_reml_band_factor(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas);
/* The maximum-likelihood variant (get_ML, linear_models.py:672-683; _ll_ / _dll_ :634-649) sums log(lambda_i + delta) and
 * 1 / (lambda_i + delta) over the spectrum of K itself: log|K + delta I| and tr (K + delta I)^-1, which the same
 * factorisations yield; s1 and s3 as above.  Round 4 (the eigendecomposition-free route evaluated REML only). */
int mmg_reml_sums_ml(mmg_ctx* ctx, mmg_reml* r, int32_t nd, const double* deltas, double* s1, double* s3, double* logdet_h,
                     double* tr_hinv, int32_t route);
/* State of the workspace's band reduction: *band_ready = 1 once K has been reduced (every later AUTO call then takes the
 * band route, whatever N); *householder_fallback = 1 if a Cholesky-QR panel was numerically rank deficient and the
 * reduction was redone with Householder panels (csrc/reml_band.hip); *seconds = what the reduction took.  NULLs allowed. */
int mmg_reml_band_info(mmg_ctx* ctx, mmg_reml* r, int32_t* band_ready, int32_t* householder_fallback, double* seconds);
int mmg_reml_scan_model(mmg_ctx* ctx, mmg_reml* r, double delta, int ndigits, double* h0_rss, double* beta,
                        double* mahalanobis_rss);
/* The same, plus C = (X'V^-1 X)^-1 X'V^-1 (q x N, row-major; V = K + delta I): the matrix the with_betas form of the scan applies to
 * every SNP for the covariates' coefficients (linear_models.py:1300-1303,1323) -- with it emmax(with_betas=True) needs no
 * eigendecomposition either. */
int mmg_reml_scan_model_c(mmg_ctx* ctx, mmg_reml* r, double delta, int ndigits, double* h0_rss, double* beta,
                          double* mahalanobis_rss, double* C_out);
/* H_sqrt_inv without an eigendecomposition (round 5).  The reference's H_sqrt_inv = diag((lambda + delta)^-1/2) U'
 * (linear_models.py:898) is one matrix H with H'H = (K + delta I)^-1; every consumer of it -- the null fit H X, H y
 * (:1141-1147, :1195-1198), the transformed SNPs H s (:1160, :1210), the scan model -- works with any such H (the reference's
 * own is fixed only up to LAPACK's eigenvector signs).  L^-1 of K + delta I = L L' is one: it is what the scan model of the
 * same delta has left in the workspace, else one factorisation + triangular inverse (N = 5000: ~10 ms against 265 ms of
 * rocSOLVER's dsyevd).
 * mmg_reml_linv_apply: out [N x k] = L^-1 V (trans = 0) or L^-T V (trans = 1); V, out column-major N x k on the host.
 * mmg_reml_linv_fetch: H_out [N x N] row-major = L^-1 (lower triangular), for callers that want the matrix itself. */
int mmg_reml_linv_apply(mmg_ctx* ctx, mmg_reml* r, double delta, int32_t trans, const double* V, int32_t k, double* out);
int mmg_reml_linv_fetch(mmg_ctx* ctx, mmg_reml* r, double delta, double* H_out);

/* ---- EMMAX scan (replaces the loop of linear_models.py:1316-1349) ------------------------- */
/* Loads the SNP-independent model onto the device:
 *   A [N x N] symmetric (= Mp Mp^T with Mp = H^T (I - QQ^T), linear_models.py:1300-1303),
 *   w [N]              (= Mp r, r the residualised transformed phenotype, :1293).
 * The off-diagonal of A, shifted into the non-negative range, is quantised to `ndigits` unsigned 7-bit digit planes
 * (exact integer GEMM on the int8 matrix cores; 4 planes = 2^-27 of max|A_ij| per entry); the diagonal and w
 * stay fp64.  ndigits in [2, 6] runs all its planes for every SNP; 0 = default: 4 digits with the adaptive
 * schedule described at mmg_scan_last_stats (p-values within 2.5e-7 relative of the 4-digit scan by construction,
 * bit-identical to it for every SNP that is refined). */
int mmg_scan_set_model(mmg_ctx* ctx, int32_t N, const double* A, const double* w, int ndigits);
/* Scan all M SNPs of g:  num = (s.w)^2, den = s'As, rss = h0_rss - num/den,
 * F = (h0_rss/rss - 1) * df2, p = f.sf(F, 1, df2) (:1345-1349).  Host outputs of length M
 * (any may be NULL).  den == 0 leaves rss = h0_rss, F = 0, p = 1 (:1308,1329). */
int mmg_emmax_scan(mmg_ctx* ctx, mmg_geno* g, double h0_rss, int32_t df2,
                   double* rss, double* F, double* p);
/* What the last mmg_emmax_scan_device did.  The default model (ndigits = 0 in mmg_scan_set_model) runs an ADAPTIVE
 * schedule: the three upper digit planes of the matrix for every SNP (the matrix truncated to 21 bits: den to ~1e-8
 * relative), then the lowest plane only for the SNPs whose p-value could move by more than 2.5e-7 relative at six
 * sigma of that rounding noise (large F, or den small against the noise), which makes those bit-identical to a full
 * 4-plane scan.  sigma_ratio_max = max over the refined SNPs of (observed change of den) / (its six-sigma
 * prediction); if it exceeds 1 the error model is rejected and everything is redone with all planes (fell_back = 1).
 * eps_max = max relative change of den over the refined SNPs.  adaptive = 0 for models with an explicit ndigits. */
int mmg_scan_last_stats(mmg_ctx* ctx, int32_t* adaptive, int64_t* n_refined, double* eps_max,
                        double* sigma_ratio_max, int32_t* fell_back);
/* The exact tier of the last scan: SNPs whose quadratic form was recomputed from the fp64 matrix because the digit planes could
 * not pin it down to the 2.5e-7 target on p (a denominator far below sum s^2 max |A|: a SNP in the span of the kinship's large
 * eigenvalues); -1: more SNPs wanted it than MMG_SCAN_EXACT_MAX_FLOP (4e13) allows -- those keep the planes' values.
 * MMG_SCAN_EXACT=0 turns the tier off (the model then does not keep its fp64 matrix: N^2 doubles). */
int mmg_scan_last_exact(mmg_ctx* ctx, int64_t* n_exact);

/* Same, leaving results in device memory (for RCCL gathers / benchmarking); fetch with
 * mmg_scan_fetch.  Blocks until the kernels finish. */
int mmg_emmax_scan_device(mmg_ctx* ctx, mmg_geno* g, double h0_rss, int32_t df2);
int mmg_scan_fetch(mmg_ctx* ctx, int64_t M, double* rss, double* F, double* p);
/* Raw per-SNP sufficient statistics of the last scan (length M each; any may be NULL):
 * dot = s.w, den = s'As, sum = s.1 -- used by the with_betas / cofactor host paths and tests. */
int mmg_scan_fetch_stats(mmg_ctx* ctx, int64_t M, double* dot, double* den, double* sum);
/* One-shot twin taking host genotypes. */
int mmg_emmax_scan_i8(mmg_ctx* ctx, const int8_t* snps, int64_t M, int32_t N,
                      const double* A, const double* w, double h0_rss, int32_t df2,
                      double* rss, double* F, double* p);

/* float32-genotype twin of the one-shot scan */
int mmg_emmax_scan_f32(mmg_ctx* ctx, const float* snps, int64_t M, int32_t N,
                       const double* A, const double* w, double h0_rss, int32_t df2,
                       double* rss, double* F, double* p);

/* ---- EMMAX permutation test (replaces linear_models.py:1157-1164) ------------------------- */
/* Ht: host [N x N] = H_sqrt_inv (row-major, as the reference holds it), Ys: host [N x P]
 * permuted residual columns (:1150-1154).  For every permutation p returns
 * min_rss[p] = min(h0_rss, min_m ( Ys_p.Ys_p - (t_m.Ys_p)^2 / (t_m.t_m) )), t_m = H (s_m - mean(s_m))
 * (:1159-1164).  min_rss: host [P].  t_m.t_m is the quadratic form of the centred model C H'H C on the int8 matrix
 * cores with the adaptive digit schedule of the scan (within 2.5e-7 of itself; MMG_SCAN_ADAPTIVE=0: every plane). */
int mmg_emmax_perm(mmg_ctx* ctx, mmg_geno* g, int32_t N, const double* Ht, const double* Ys,
                   int32_t P, double h0_rss, int ndigits, double* min_rss);

/* One-shot twin taking host genotypes (mmg_emmax_perm_i8(ctx, snps, M, N, Ht, Ys, P, min_rss)). */
int mmg_emmax_perm_i8(mmg_ctx* ctx, const int8_t* snps, int64_t M, int32_t N, const double* Ht, const double* Ys,
                      int32_t P, double h0_rss, double* min_rss);

/* ---- multi-phenotype scans (replaces a LOOP of emmax() runs over phenotypes that share genotypes and kinship:
 * phenotypeData.py:70-78, hdf5_data.py:262-330 once per phenotype file) ------------------------ */
/* Every phenotype p has its own variance ratio delta_p, hence its own H_p = diag((lambda+delta_p)^-1/2) U' and its
 * own N x N scan matrix; with the eigenvectors U of K shared, everything SNP-dependent is a function of the rotated
 * SNP tau_m = U s_m (linear_models.py:898,1290-1303,1328 in the eigenbasis):
 *     den = sum_i d_p[i] tau_mi^2 - sum_c (sum_i G_p[c][i] tau_mi)^2,   dot = sum_i omega_p[i] tau_mi,
 *     rss = h0_rss_p - dot^2/den,  F = (h0_rss_p/rss - 1) df2,  p = f.sf(F, 1, df2)
 * with d_p = w^2, G_p[c] = Q_p[:,c] * w, omega_p = r_p * w, w = (lambda+delta_p)^-1/2, Q_p an orthonormal basis of
 * the transformed covariates and r_p the residual of the transformed phenotype (host glue: O(N q^2) per phenotype).
 * mmg_rot_create: evecs_rows host [N x N], ROWS are eigenvectors (mmg_eigh_f64's layout); digits them once and
 * allocates T for up to M_cap SNPs (8 N bytes per SNP; fp64, eigen-major inside blocks of 256 SNPs).
 * mmg_rot_load: T = S U' for the SNPs of g (exact int8-MFMA digit GEMM); g may be destroyed afterwards.
 * mmg_emmax_scan_multi: d, omega host [P x N]; G host [P x q x N]; h0_rss host [P]; 1 <= q <= 8 (q > 4: batches of 8 on the matrix-pipe kernel); outputs host
 * [P x M] (any may be NULL).  One HBM-bound pass over T per 8 phenotypes.
 * mmg_rot_fetch: out host [N x rows] = T[:, m0:m0+rows] (tests / diagnostics). */
int mmg_rot_create(mmg_ctx* ctx, int32_t N, const double* evecs_rows, int64_t M_cap, mmg_rot** r);
int mmg_rot_destroy(mmg_ctx* ctx, mmg_rot* r);
int mmg_rot_load(mmg_ctx* ctx, mmg_rot* r, mmg_geno* g);
int mmg_rot_fetch(mmg_ctx* ctx, mmg_rot* r, int64_t m0, int64_t rows, double* out);
int mmg_emmax_scan_multi(mmg_ctx* ctx, mmg_rot* r, int32_t P, int32_t q, const double* d, const double* omega,
                         const double* G, const double* h0_rss, int32_t df2, double* rss, double* F, double* p);

/* ---- p-values (replaces scipy.stats.f.sf at linear_models.py:1349,1172) ------------------- */
/* Upper tail of F(1, df2) evaluated on the device for n values (host in/out). */
int mmg_f_sf(mmg_ctx* ctx, const double* F, int64_t n, int32_t df2, double* p);

/* ---- multi-GPU (RCCL over xGMI; one process per GPU) -------------------------------------- */
/* id: 128-byte ncclUniqueId produced by rank 0 and distributed by the host launcher. */
int mmg_comm_unique_id(unsigned char id[128]);
int mmg_comm_create(mmg_ctx* ctx, const unsigned char id[128], int rank, int world, mmg_comm** c);
int mmg_comm_destroy(mmg_ctx* ctx, mmg_comm* c);
/* all-gather of equal-sized per-rank SNP result blocks that live on the device after
 * mmg_emmax_scan_device: recv_* are host buffers of world*count doubles (rank-major). */
int mmg_comm_allgather_scan(mmg_ctx* ctx, mmg_comm* c, int64_t count,
                            double* rss, double* F, double* p);
/* The same delivery in the background (what the all-gather row of SURVEY 8e becomes when scans are
 * issued back to back, e.g. per chromosome / per phenotype): snapshots the device-resident
 * (rss, F, p) of the last scan, then on a second HIP stream all-gathers them over RCCL (c != NULL;
 * host buffers of world*count doubles) or just downloads this rank's block (c == NULL; count
 * doubles), so that the next mmg_emmax_scan_device overlaps the gather and the PCIe transfer.
 * Host buffers must stay valid -- and should be page-locked (mmg_host_alloc) -- until
 * mmg_scan_deliver_wait returns.  One delivery in flight per context: begin waits for the previous. */
int mmg_scan_deliver_begin(mmg_ctx* ctx, mmg_comm* c, int64_t count, double* rss, double* F, double* p);
int mmg_scan_deliver_wait(mmg_ctx* ctx);
/* Sharded twins of the entry points above: this rank holds a block of the SNP axis; the partial
 * results never leave HBM before the RCCL reduction over xGMI.  comm == NULL (or a 1-rank communicator) makes each
 * identical to its single-GPU form.
 *   mmg_kinship_ibs_i8_sharded  C_out = SUM over ranks of the exact int64 count matrices (all ranks get it)
 *   mmg_kin_acc_allreduce       in-place SUM of the device-resident fp64 accumulator and of its SNP count
 *   mmg_emmax_perm_sharded      min_rss over the SNPs of ALL ranks (RCCL MAX of the per-permutation statistic) */
int mmg_kinship_ibs_i8_sharded(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int64_t* C_out);
int mmg_kin_acc_allreduce(mmg_ctx* ctx, mmg_comm* comm, mmg_kin_acc* acc);
int mmg_emmax_perm_sharded(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int32_t N, const double* Ht, const double* Ys,
                           int32_t P, double h0_rss, int ndigits, double* min_rss);
/* The same test right after an EMMAX scan of the same store with the same H (the flow of hdf5_data.py:315-330): the
 * quadratic form t.t = s~'H'H s~ is rebuilt from the scan's den = s'(H'H - sum_c u_c u_c')s (still in HBM) plus
 * 1 + q dot products per SNP instead of a second O(N^2) pass: HtQ host [q x N], rows u_c = H'Q_c with Q the
 * orthonormal basis of the scan's transformed covariates (linear_models.py:1300).  Fails with MMG_E_STATE unless the
 * last mmg_emmax_scan_device of ctx ran over g in its current state.  den carries the scan's precision (adaptive
 * digit schedule: ~1e-8 relative where it did not refine), i.e. min_rss to ~1e-9 relative instead of 1e-12. */
int mmg_emmax_perm_after_scan(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int32_t N, const double* Ht, const double* Ys,
                              int32_t P, double h0_rss, const double* HtQ, int32_t q, double* min_rss);
/* The SNP-independent half of the test (linear_models.py:1135-1156 plus the operand images: H'H as digit planes,
 * W' = Ys'H as a 4-digit int8 image, v = H'H 1, Ys_p.Ys_p) prepared ONCE and run over any number of genotype stores
 * -- the chunk loop of hdf5_data.py:294-330, the ranks' blocks, repeated tests.  mmg_perm_plan_run: HtQ == NULL is
 * the stand-alone test (exact t.t, all four planes); HtQ [q x N] is the after-scan form above.  The one-shot entry
 * points are create + run + destroy. */
int mmg_perm_plan_create(mmg_ctx* ctx, int32_t N, const double* Ht, const double* Ys, int32_t P, double h0_rss,
                         mmg_perm_plan** plan);
/* flags bit 0: do NOT mean-centre the SNPs before the transform, t_m = Ht s_m as given.  The public
 * LinearMixedModel.emmax_permutations (linear_models.py:1180-1230) centres the TRANSFORMED SNP instead
 * (Xs - mean(Xs), :1211): t_m = C H s_m with C = I - 11'/n, i.e. this plan built on Ht = C H.  Stand-alone runs only. */
int mmg_perm_plan_create_ex(mmg_ctx* ctx, int32_t N, const double* Ht, const double* Ys, int32_t P, double h0_rss,
                            int flags, mmg_perm_plan** plan);
/* The plan with H = L^-1 of K + delta I = L L' taken from a REML workspace as it lies in HBM (mmg_reml_linv_apply above): the
 * permutation test of _emmax_permutations_ / emmax_permutations / hdf5_data.run_emmax_perm without rocSOLVER's dsyevd and
 * without an N x N matrix crossing PCIe.  Ys must be shuffles of residuals computed in the same basis (H X, H y from
 * mmg_reml_linv_apply of this workspace and delta).  flags bit 0 as above; bit 1 (both entry points): H <- C H on the device
 * (the public test's centring of the transformed SNP), so that the caller need not form C H. */
int mmg_perm_plan_create_from_reml(mmg_ctx* ctx, mmg_reml* r, double delta, const double* Ys, int32_t P, double h0_rss,
                                   int flags, mmg_perm_plan** plan);
int mmg_perm_plan_run(mmg_ctx* ctx, mmg_comm* comm, mmg_perm_plan* plan, mmg_geno* g, const double* HtQ, int32_t q,
                      double* min_rss);
int mmg_perm_plan_destroy(mmg_ctx* ctx, mmg_perm_plan* plan);
/* rank / world of the communicator and the rank count RCCL itself reports (ncclCommCount); any may be NULL */
int mmg_comm_info(mmg_comm* c, int* rank, int* world, int* nccl_count);
/* all-gather of equal-sized host blocks (count doubles per rank; recv: world*count, rank-major) */
int mmg_comm_allgather_f64(mmg_ctx* ctx, mmg_comm* c, const double* send, int64_t count, double* recv);
/* in-place all-reduce of host double buffers through device staging (SUM: partial kinship;
 * MIN: permutation minima).  op: 0 = sum, 1 = min, 2 = max. */
int mmg_comm_allreduce_f64(mmg_ctx* ctx, mmg_comm* c, double* buf, int64_t count, int op);
int mmg_comm_allreduce_i64(mmg_ctx* ctx, mmg_comm* c, int64_t* buf, int64_t count, int op);
int mmg_comm_barrier(mmg_ctx* ctx, mmg_comm* c);

#ifdef __cplusplus
}
#endif
#endif
