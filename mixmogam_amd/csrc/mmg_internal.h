// mmg_internal.h -- shared declarations of libmixmogam_hip (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
// The library is built with -fvisibility=hidden (Makefile): what include/mixmogam_hip.h declares is the whole export list --
// `nm -D` of the built library equals the header (tests/test_host.py) -- and helpers shared between the source files
// (mmg_reml_create_dev, the mmg:: namespace) stay inside the shared object.
#pragma GCC visibility push(default)
#include "../../include/mixmogam_hip.h"
#pragma GCC visibility pop

#ifdef MMG_GUARD
// Diagnostic build (make GUARD=1, guard.hip): the library's device buffers get guard bands that are checked when they are freed.
hipError_t mmg_guard_malloc(void** p, size_t bytes, const char* file, int line);
hipError_t mmg_guard_free(void* p);
template <typename T>
static inline hipError_t mmg_guard_malloc_t(T** p, size_t bytes, const char* file, int line) {
  return mmg_guard_malloc((void**)p, bytes, file, line);
}
#define hipMalloc(p, bytes) mmg_guard_malloc_t(p, bytes, __FILE__, __LINE__)
#define hipFree(p) mmg_guard_free((void*)(p))
extern "C" __attribute__((visibility("default"))) void mmg_guard_note(const char* fn);    // breadcrumb: the guard's abort handler prints the last entry points
extern "C" __attribute__((visibility("default"))) void mmg_guard_launched(hipStream_t s); // MMG_GUARD_SYNC=1: wait for the kernel just launched (a fault then names it)
void mmg_guard_launch_check(const char* kernel, dim3 grid, dim3 block, size_t lds);     // a launch the runtime refused: say which (the error stays for the caller)
hipError_t mmg_guard_func_attr(const char* what, int line, const void* f, hipFuncAttribute a, int v);   // hipFuncSetAttribute; a refusal is reported
#define hipFuncSetAttribute(...) mmg_guard_func_attr(#__VA_ARGS__, __LINE__, __VA_ARGS__)   // (kernel names carry template commas)
#define MMG_NOTE_ENTRY() mmg_guard_note(__func__)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                   \
  do {                                                                                                                      \
    mmg_guard_note(#kernelName);                                                                                            \
    hipLaunchKernelGGLInternal((kernelName), numBlocks, numThreads, memPerBlock, streamId, ##__VA_ARGS__);                  \
    mmg_guard_launch_check(#kernelName, dim3(numBlocks), dim3(numThreads), (size_t)(memPerBlock));                          \
    mmg_guard_launched(streamId);                                                                                           \
  } while (0)
#else
#define MMG_NOTE_ENTRY() ((void)0)
#endif

struct mmg_geno {
  int64_t M = 0, Mpad = 0;      // SNPs, padded to 256
  uint64_t version = 0;         // bumped by every write path
  int64_t Mcap = 0;             // rows allocated (mmg_geno_reset may shrink M / Mpad below it and grow them back)
  int32_t N = 0, Npad = 0;      // individuals, padded to 256
  int8_t* d = nullptr;          // [Mpad x Npad] SNP-major, zero padded
  // lazily built bit-packed twin [Mpad x Npad/8] (k_scan_bits.hip); invalidated by every write
  uint8_t* bits = nullptr;
  bool bits_valid = false, binary = false;
  // upper bound of |s| over everything ever written to the store (updated by every write path); the scan uses
  // it to prove that its 32-bit epilogue cannot overflow
  int smax = 0;                 // running bound of max |s| over everything written to the store
  int sneg = 0;                 // ... and of max(-s): 0 = no negative value was ever written
  int* d_smax = nullptr;        // device pair {max |s|, max(-s)}
  // E2M1 twin of the store for the FP4 kinship GEMM (k_kinship.hip:kinship_f4_tr_kernel): [Mcap x Npad / 2] nibbles, bit 0 of
  // every genotype byte as 0x0 / 0x2 (= 0.0 / 1.0), kept in step by every write path -- the pass that folds max |s| of the
  // written rows reads them anyway (k_pack.hip:absmax_i8_kernel<true>), the synthetic generators emit both forms.  Usable
  // as long as the store is binary (smax <= 1, sneg == 0).  nullptr: MMG_FP4_TWIN=0, or the allocation failed (then the
  // kinship call writes a scratch image itself, as in round 3: 1.3 ms of a 6.6 ms call at N = 5000 x M = 1e6).
  uint8_t* fp4 = nullptr;
  // [s = 2] bit image of a store of 0/1/2 codes (/root/reference/plink2hdf5.py:171-179), for the scan's finalize step
  // (k_scan.hip:lin_hi_bits_kernel): [Mcap x Npad / 8], bit k of a row (LSB first) = bit 1 of genotype byte k.  With it
  // sum_i A_ii s_i^2 = sum_i A_ii s_i + 2 sum_i A_ii [s_i = 2] needs 1/8 of the store's bytes instead of all of them.
  // Built on the SECOND scan of the same content (hi2_version == version): a store that is scanned once -- a streamed chunk --
  // keeps the finalize pass over its bytes, which costs what building the image would.
  uint8_t* hi2 = nullptr;
  uint64_t hi2_version = ~0ull;       // the write version the image was built from
  uint64_t scanned_version = ~0ull;   // the write version the scans below saw
  int scans_of_version = 0;
};

enum { EV_KIN = 0, EV_QUAD = 1, EV_FIN = 2, EV_PERM = 3, EV_EIGH = 4, EV_PACK = 5, EV_QUAD2 = 6, EV_ROT = 7, EV_MULTI = 8, EV_GRM = 9, EV_COUNT = 10 };

struct mmg_scan_model {
  int32_t N = 0, Npad = 0, D = 0;
  int8_t* Bq = nullptr;         // [D][Npad][Npad] digits of the strictly-lower triangle of 2A
  double* A64 = nullptr;        // [N][N] the fp64 matrix itself, for the exact tier (api.hip:exact_tier); null: tier off
  mutable bool coherent = false;   // rounding errors of equal entries add up (refused the adaptive schedule, or seen by the
  mutable bool exact_checked = false;   // exact tier's sample check on the first scan of the model: api.hip:exact_tier)
  double* diag = nullptr;       // [Npad] diagonal of A (0 padded)
  double* w = nullptr;          // [Npad] (0 padded)
  double step = 0.0;            // den = step * (q' - offset * sum_{j>k} s_j s_k) + sum_i diag_i s_i^2
  double offset = 0.0;          // 2^(7 D - 1): what quantize_kernel adds to every stored entry (non-negative digits)
  // tile schedule
  int AS = 2, G = 16;           // sub-blocks per cohort, job groups per cohort (AS * G = 32)
  int* job_off = nullptr;       // [G + 1]
  int2* jobs = nullptr;         // (digit, J)
  int njobs = 0;
  // adaptive precision (default model, D = 4): schedules over the upper planes d = 1..D-1 and over plane 0
  bool adaptive = false;
  double mu0 = 0.0;             // mean of the lowest digit over the stored (j > k) entries
  int *job_off_hi = nullptr, *job_off_lo = nullptr;
  int2 *jobs_hi = nullptr, *jobs_lo = nullptr;
  int njobs_hi = 0, njobs_lo = 0;
  // The last, partly filled cohort of a launch runs with fewer SNP blocks per XCD and the jobs split over more groups
  // (AS / 2, AS / 4): a round of it is 2x / 4x shorter.  Schedules per plane range (0 all planes, 1 upper planes,
  // 2 plane 0) and halving (0: AS / 2, 1: AS / 4); `range` says which of them this (shallow copy of the) model runs.
  int range = 0;
  int* tail_off[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  int2* tail_jobs[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  // linear rows (api.hip:add_linear_rows): rows Npad-16 .. Npad-1 of the top digit plane hold digit images of w and
  // diag(A) and a row of ones, so that the quadratic-form GEMM yields s.w, sum diag_i s_i and sum s_i of every SNP as
  // a by-product (binary stores) and the finalize pass need not read the genotype store a second time
  bool lin_rows = false;
  int8_t* lin_tab = nullptr;     // [8][Npad]: the seven digit rows of diag(A) and a row of ones once more, for lin_hi_bits_kernel
  double lin_step_w = 0.0, lin_step_d = 0.0;
};

struct LinOut {                  // where the GEMM leaves the raw accumulators of the linear rows: [Mpad][16] ints
  int* raw;
};

struct mmg_scan_result {
  int64_t cap = 0;              // capacity in SNPs (padded)
  int64_t M = 0;
  unsigned long long* q = nullptr;
  double *rss = nullptr, *F = nullptr, *p = nullptr, *dot = nullptr, *den = nullptr, *sum = nullptr;
  // adaptive precision: sum_i A_ii s_i^2 per SNP, the indices of the SNPs that get the lowest digit plane,
  // their count / the observed max relative den change (device scalars), the quadratic forms of the compact store
  double *dd = nullptr, *ssq = nullptr;   // sum_i A_ii s_i^2 and sum_i s_i^2 per SNP
  int64_t* idx = nullptr;
  unsigned long long* scal = nullptr;      // [0] = count, [1] = max eps bits, [2] = max (observed / 6 sigma) bits
  unsigned long long* q2 = nullptr;
  int64_t q2_cap = 0;
  int* linraw = nullptr;         // [cap][16] raw accumulators of the model's linear rows (k_scan_w4s.hip LIN)
  int* linraw2 = nullptr;        // [cap][8] stores of 0/1/2 codes: digits of sum_i A_ii [s_i = 2] and the count of 2s (lazily)
  const void* geno = nullptr;    // the store the last scan ran on and its write version (mmg_emmax_perm_after_scan)
  uint64_t geno_version = 0;
  // what the last scan did (mmg_scan_last_stats)
  int64_t n_refined = 0;
  double eps_max = 0.0, sigma_ratio_max = 0.0;
  int fell_back = 0, adaptive = 0;
  int64_t n_exact = 0;           // SNPs recomputed from the fp64 matrix (exact tier); -1: wanted for more SNPs than its budget
};

struct mmg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  hipEvent_t ev[EV_COUNT][2];
  bool ev_set[EV_COUNT];
  int n_cu = 0;
  mmg_scan_model model;
  mmg_scan_result res;
  void* rocblas = nullptr;      // rocblas_handle, created lazily
  mmg_geno* sel_geno = nullptr; // compact store of the SNPs refined by the adaptive scan (grown on demand)
  // background delivery of scan results (mmg_scan_deliver_*): second stream, snapshot staging, one in flight
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_snap = nullptr, ev_deliver = nullptr;
  double* dstage = nullptr;
  size_t dstage_elems = 0;
  bool deliver_pending = false;
  double multi_ms_total = 0.0;  // summed pass time of the last mmg_emmax_scan_multi
  double grm_ms_total = 0.0;    // summed digit-plane GEMM time of the last mmg_kin_acc_add_grm
  int2* grp_tab = nullptr;      // workgroup-group table of the perm / rotation GEMM launches (gemm_i8_w4s.h)
  size_t grp_cap = 0;
  std::vector<int2> grp_host;   // host image of grp_tab (source of the asynchronous upload: must outlive it)
  void* jobs = nullptr;         // device job lists of the kinship GEMM launches (k_kinship.hip:job_buffer), grown on demand:
  size_t jobs_cap = 0;          // a hipMalloc / hipFree per launch serialised the streams and leaked on an early return
  void* band_keep = nullptr;    // banded factors kept by mmg_reml_band_factor (<= 2 GB) for the workspace band_keep_owner: the
  size_t band_keep_cap = 0;     // sums at those variance ratios then cost the substitutions and the trace recurrence only
  const void* band_keep_owner = nullptr;
  void* band_ws = nullptr;      // per-delta factors / right-hand sides of reml_band_sums up to 2 GB (650 MB at N = 5000 for the 227
  size_t band_ws_cap = 0;       // variance ratios of a search): a hipMalloc + hipFree per emmax() call was ~8 ms of a 90 ms call
  void* ingest = nullptr;       // device staging of the genotype ingest paths (packed rows, pageable int8 rows); kept:
  size_t ingest_cap = 0;        // hipMalloc / hipFree per chunk would serialise the upload stream with the compute stream
};

namespace mmg {

int set_err(mmg_ctx* ctx, int code, const std::string& msg);
#define MMG_HIP(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return mmg::set_err(ctx, MMG_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__) + " (" + __FILE_NAME__ + ":" + \
                                              std::to_string(__LINE__) + ")");               \
  } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Call-scoped device scratch: everything allocated through it is freed when the entry point returns,
// on every path (the MMG_HIP early returns included).
struct Scratch {
  std::vector<void*> ptrs;
  template <typename T>
  hipError_t alloc(T** out, size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    *out = (T*)p;
    if (e == hipSuccess) ptrs.push_back(p);
    return e;
  }
  ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
};

struct EvScope {  // records the two events of slot `which` around a region on ctx->stream
  mmg_ctx* c; int w;
  EvScope(mmg_ctx* ctx, int which) : c(ctx), w(which) { hipEventRecord(c->ev[w][0], c->stream); }
  ~EvScope() { hipEventRecord(c->ev[w][1], c->stream); c->ev_set[w] = true; }
};

// ---- k_pack.hip
void launch_fill_hash(mmg_ctx*, mmg_geno*, uint64_t seed, int64_t m_global0, uint32_t thr16);     // (+ the store's FP4 twin)
void launch_fill_struct(mmg_ctx*, mmg_geno*, uint64_t seed, int64_t m_global0, int npop, uint32_t spread_q16);
// fp32 / fp64 genotype ingest -> int8; *d_bad |= 1 if any value is not an integer in [-127, 127]
void launch_unpack(mmg_ctx*, const uint8_t* src, int64_t row_bytes, int8_t* dst, int64_t rows, int32_t N, int32_t Npad,
                   int bits, uint32_t lut, uint8_t* x4 = nullptr);   // 1- / 2-bit packed rows -> int8 store rows (+ FP4 twin rows)
void launch_pitch_rows(mmg_ctx*, const int8_t* src, int8_t* dst, int64_t rows, int32_t N, int32_t Npad);
void launch_unpitch_rows(mmg_ctx*, const int8_t* src, int8_t* dst, int64_t rows, int32_t N, int32_t Npad);   // [rows x N] -> [rows x Npad], zero padded
void launch_cvt_f32(mmg_ctx*, const float* src, int8_t* dst, int64_t rows, int32_t N, int64_t ld, int* d_bad);
void launch_cvt_f64(mmg_ctx*, const double* src, int8_t* dst, int64_t rows, int32_t N, int64_t ld, int* d_bad);
// Xt [Npad x Mk] = transpose of S with value map v -> mul*v + add for valid cells, 0 elsewhere.
// (SNP rows [m_begin, m_begin + Mk) of the store; Mk a multiple of 128, m_begin + Mk <= Mpad)
// thr > 0: indicator image [s >= thr] instead of the affine map
void launch_transpose(mmg_ctx*, const mmg_geno*, int8_t* Xt, int64_t Mk, int mul, int add, int64_t m_begin,
                      int thr = 0);
void launch_snp_stats(mmg_ctx*, const mmg_geno*, double* mean, double* sd);
// d_out[0] = max(d_out[0], max |p[i]|), d_out[1] likewise for max(-p[i]); bytes % 16 == 0.  x4 (optional): the E2M1 twin of the
// same bytes (bit 0 of byte i -> nibble i of x4, value 0x0 / 0x2), written in the same pass
void launch_absmax_i8(mmg_ctx*, const int8_t* p, int64_t bytes, int* d_out, uint8_t* x4 = nullptr);

// row sums and diagonal of a row-major fp64 [N x N] matrix (one block per row, fixed order); x[i] *= f
void launch_row_sums_f64(mmg_ctx*, const double* A, int64_t N, double* rows, double* diag);
void launch_scale_f64(mmg_ctx*, double* x, int64_t n, double f);

// ---- k_kinship.hip
int kinship_pick_ksplit(int32_t Npad, int64_t Mk, bool f32);
// C32 [Npad x Npad] int32 += Xt Xt^T (upper-triangular tiles only, mirrored by the caller).
void launch_pack_fp4_on(mmg_ctx*, hipStream_t stream, const int8_t* S, int64_t rows, int32_t Npad, uint8_t* X4, int thr,
                        bool binary);   // [s >= thr] (the genotypes themselves for binary stores, thr 1) as E2M1 nibbles
int run_kinship_f4_tr(mmg_ctx*, mmg::Scratch& sc, const uint8_t* X4, int32_t Npad, int64_t nk4, int* C32);   // enqueues, no sync
int run_kinship_i8_tr(mmg_ctx*, const int8_t* Sp, const int8_t* Sq, int64_t ld, int32_t Npad, int64_t nk, int* C32);
// all four digit planes of the exact GRM of a binary store in one pass (gemm_i8_grm4.h); dig: device [4][dig_stride]
int run_kinship_grm4(mmg_ctx*, const int8_t* S, int64_t ld, int32_t Npad, int64_t nk, const int8_t* dig, int64_t dig_stride,
                     int* C32);
int run_kinship_i8(mmg_ctx*, const int8_t* Xt, int32_t Npad, int64_t Mk, int* C32);
int run_kinship_i8_pq(mmg_ctx*, const int8_t* Xp, const int8_t* Xq, int32_t Npad, int64_t Mk, int* C32);
void launch_grm_combine(mmg_ctx*, const int* C32, int D, int32_t Npad, int32_t N, double step, double base,
                        const double* c1, double c0, double* C, int accumulate);
// SNP-major digit images for the transposed-read kinship GEMM + the weighted column sums of the chunk (k_pack.hip)
int64_t grm_partial_doubles(int64_t Mk, int32_t Npad);
int64_t grm_weight_blocks(int64_t M);
void launch_grm_weight_stats(mmg_ctx*, const double* mean, const double* sd, int64_t M, double* out /*[blocks][4]*/);
void launch_grm_digits(mmg_ctx*, const double* mean, const double* sd, int64_t mb, int64_t M, int64_t Mk, double step, int bd,
                       int D, int8_t* dig, double* coef, int64_t stream_pos = 0);   // stream_pos: SNPs of the accumulator's earlier calls
void launch_add_into_f64(mmg_ctx*, double* dst, const double* src, int64_t n);
void launch_grm_scale_rows(mmg_ctx*, const int8_t* S, int64_t rows_valid, int64_t Mk, int32_t Npad, bool neg, int8_t* Xp,
                           const int8_t* dig, int D, const double* coef, double* partial, double* c1, int32_t n_shift = 0);
void launch_add_scalar_f64(mmg_ctx*, double* x, int64_t n, double v);
void launch_transpose_digits(mmg_ctx*, const mmg_geno*, int8_t* Xq, int8_t* Xp, int64_t Mk, int64_t m_begin,
                             const int8_t* dig, int D);
void launch_snp_dot_raw(mmg_ctx*, const int8_t* S, int64_t ldS, int64_t rows, int32_t len16, const double* v, double* out);
// slabs [ksplit][Npad x Npad] fp32; returns ksplit through *ksplit_out.
int run_kinship_f32(mmg_ctx*, const int8_t* Xt, int32_t Npad, int64_t Mk, const float* scale,
                    const float* shift, float* slabs, int ksplit);
void launch_reduce_slabs(mmg_ctx*, const float* slabs, int ksplit, int32_t Npad, int32_t N, double* C, int accumulate);
void launch_mirror_i32_to_i64(mmg_ctx*, const int* C32, int32_t Npad, int32_t N, int64_t* C);
void launch_ibs_counts_to_f64(mmg_ctx*, const int64_t* C, int64_t n, double two_m, double* K);
void launch_ibs_diploid_combine(mmg_ctx*, const int64_t* c1, const int64_t* c2, int64_t N, double M, double* K);   // c2 may be nullptr: c1 is the sum
void launch_pack_fp4_two(mmg_ctx*, const int8_t* S, int64_t rows, int32_t Npad, uint8_t* X4a, uint8_t* X4b);

// ---- k_scan.hip
void launch_absmax_offdiag(mmg_ctx*, const double* A, int32_t N, unsigned long long* out_bits);
void launch_quantize(mmg_ctx*, const double* A, int32_t N, int32_t Npad, int D, double inv_step, long long offset,
                     int8_t* Bq, double* diag, long long* z0_sum /*dev, accumulated; may be null*/,
                     long long* z0_tile /*dev [Npad/256]^2, accumulated; may be null*/);
// ---- k_scan_w4s.hip: the production quadratic-form GEMM
void launch_scan_quad_w4s(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, unsigned long long* q, const LinOut* lin = nullptr);
// ---- experiments/ (only in a `make EXPERIMENTS=1` library): superseded generations, bit-identical
void launch_scan_quad(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, unsigned long long* q);   // 8-wave family
int ensure_bits(mmg_ctx*, mmg_geno*);
void launch_scan_quad_bits(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, unsigned long long* q);
void launch_scan_quad_w4m(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, unsigned long long* q);
void launch_scan_quad_w4b(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, unsigned long long* q);
// ev_slot: which event pair brackets the kernel (EV_QUAD, or EV_QUAD2 for the refinement pass)
int run_scan_quad(mmg_ctx*, mmg_geno*, const mmg_scan_model&, unsigned long long* q, int ev_slot = EV_QUAD,
                  const LinOut* lin = nullptr);
bool scan_lin_usable(const mmg_geno* g, const mmg_scan_model& md);   // by-products available for this store / model
void launch_scan_finalize(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, mmg_scan_result&,
                          double h0_rss, int32_t df2, double lnbeta, bool with_p = true, double bias = 0.0);
void launch_scan_finalize_lin(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, mmg_scan_result&, double h0_rss,
                              int32_t df2, double lnbeta, bool with_p = true, double bias = 0.0, const int* raw2 = nullptr);
void launch_pack_hi_bits(mmg_ctx*, const mmg_geno*);                                   // g->d -> g->hi2
void launch_lin_hi_bits(mmg_ctx*, const mmg_geno*, const mmg_scan_model&, int* raw2);  // g->hi2 x md.lin_tab -> raw2 [Mpad][8]
inline bool geno_hi2_ready(const mmg_geno* g) { return g->hi2 != nullptr && g->hi2_version == g->version; }
void launch_scan_select(mmg_ctx*, const mmg_scan_result&, int64_t M, double sig_unit, double target, unsigned long long* cnt,
                        bool use_F = true);
void launch_gather_rows(mmg_ctx*, const mmg_geno*, const int64_t* idx, int64_t cnt, int8_t* Sc);
void launch_scan_select_exact(mmg_ctx*, const mmg_scan_result&, int64_t M, double sig_full, double half_step, bool coherent,
                              double target, unsigned long long* cnt, bool use_F);
void launch_rows_to_f64(mmg_ctx*, const int8_t* Sc, int32_t Npad, int32_t N, int64_t rows, double* Sd);
void launch_scan_exact_den(mmg_ctx*, const double* Sd, int32_t N, int64_t cnt, const double* A64, double* part);
void launch_scan_sample_idx(mmg_ctx*, int64_t M, int64_t cnt, int64_t* idx);
void launch_scan_exact_check(mmg_ctx*, const int64_t* idx, int64_t cnt, const double* part, int32_t N, const mmg_scan_result&,
                             double sig_used, unsigned long long* ratio_bits);
void launch_scan_exact_apply(mmg_ctx*, const int64_t* idx, int64_t cnt, const double* part, int32_t N, mmg_scan_result&,
                             double h0_rss, int32_t df2);
void launch_scan_refine(mmg_ctx*, const int64_t* idx, int64_t cnt, const mmg_scan_model&, mmg_scan_result&,
                        const unsigned long long* q2, double sig_unit, double h0_rss, int32_t df2,
                        unsigned long long* eps_bits);
void launch_snp_dot(mmg_ctx*, const mmg_geno*, const double* v /*[Npad] dev*/, double* out /*[M] dev*/);
void launch_f_sf(mmg_ctx*, const double* F, int64_t n, int32_t df2, double lnbeta, double* p);

// ---- k_perm.hip
// mu[m] = sum/N, inv[m] = 1/(den - 2 mu dot + mu^2 c0) (0 for SNPs that are constant after centring)
void launch_perm_center(mmg_ctx*, const mmg_geno*, const mmg_scan_result&, double c0, double* d_mu, double* d_inv);
void launch_perm_center_reuse(mmg_ctx*, const mmg_geno*, const double* den, const double* dots, int q, const double* sum,
                              double c0, double* d_mu, double* d_inv);
// d_maxstat[p] = max_m (s~_m . W_p)^2 * inv[m];  dWt: device [P x N] row-major fp64
// d_ssum [Mpad]: exact genotype sum of every SNP (0 for the padding rows): takes the non-negative digit offset of the
// operand rows out of the accumulators (gemm_i8_w4s.h ROWS_OFFSET)
int run_perm(mmg_ctx*, const mmg_geno*, int32_t N, const double* dWt, int32_t P, const double* d_inv,
             const double* d_mu, const double* d_ssum, int ndigits, double* d_maxstat);

int run_perm_q(mmg_ctx*, const mmg_geno*, const int8_t* Wq, const double* dstep, const double* dcsum, int32_t P,
               const double* d_inv, const double* d_mu, const double* d_ssum, double* d_maxstat);
void launch_colsum(mmg_ctx*, const mmg_geno*, unsigned long long* r);          // r[Npad] += column sums of the store
void launch_mirror_ibs(mmg_ctx*, const int* C32, int32_t Npad, int32_t N, const long long* r, long long Mtot, int64_t* C);
// ---- k_perm.hip: centring of the permutation test's operands, 1 / t.t from a centred quadratic form
void launch_center_sym(mmg_ctx*, double* A, int32_t N, const double* v, double c0);   // A <- C A C given v = A 1, c0 = 1'A 1
void launch_center_rows(mmg_ctx*, double* Wt, int32_t N, int32_t P);                  // rows of Wt [P x N] <- row - mean(row)
void launch_sub_row_mean(mmg_ctx*, double* A, int32_t N, const double* colsum);       // A [N x N] row-major: rows minus colsum / N
void launch_perm_inv(mmg_ctx*, const mmg_geno*, const mmg_scan_result&, double* d_mu, double* d_inv);
int upload_group_table(mmg_ctx*, const std::vector<int2>& tab);   // k_perm.hip; into ctx->grp_tab
int quantize_rows_4digits(mmg_ctx*, const double* dWt, int32_t N, int32_t Npad, int32_t P, int8_t* Wq, double* dstep,
                          double* dcsum);

// ---- k_rot.hip: eigen-rotated genotype store + multi-phenotype scan
// T [Mpad/256][nVT*64][256] (fp64; eigen-major inside 256-SNP blocks): T[m/256][i][m%256] = u_i . s_m for the SNPs
// of g (exact int8 digit GEMM)
int run_rotate(mmg_ctx*, const mmg_geno* g, const int8_t* Vq, const double* dstep, const double* d_ssum /*[Mpad]*/, int nVT,
               double* T);
// one pass over T for PB <= 8 phenotypes with q <= 4 fixed-effect columns each; coef: device [N][PB*(2+q)]
int run_scan_multi(mmg_ctx*, const double* T, int64_t nrows, int32_t N, int64_t M, int PB, int q, const double* coef,
                   const double* h0 /*device [PB]*/, int32_t df2, double lnbeta, double* rss, double* F, double* p,
                   int64_t ldOut);

}  // namespace mmg
