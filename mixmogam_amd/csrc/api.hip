// api.hip -- the extern "C" entry points of libmixmogam_hip.so (see include/mixmogam_hip.h).
#include <rccl/rccl.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "gemm_i8_core.h"
#include "mmg_internal.h"

static thread_local std::string g_last_error;            // (the chunk prefetcher calls in from a thread of its own)

namespace mmg {
int set_err(mmg_ctx* ctx, int code, const std::string& msg) {
  g_last_error = msg;
  if (ctx) ctx->err = msg;
  return code;
}
}  // namespace mmg
using namespace mmg;

#define MMG_CHECK_ARG(ctx, cond)                                                   \
  do {                                                                             \
    if (!(cond)) return set_err(ctx, MMG_E_ARG, std::string("bad argument: ") + #cond); \
  } while (0)
// Every entry point that allocates, copies or launches binds the calling thread to the context's device first: HIP's
// current device is per host thread (and defaults to 0), and a context may be driven from helper threads (the
// chunk prefetcher of hdf5_data.py) or beside contexts of other devices in the same process.
#define MMG_ENTER(ctx)                                                              \
  do {                                                                              \
    MMG_NOTE_ENTRY();                                                               \
    if (!(ctx)) return set_err(nullptr, MMG_E_ARG, "bad argument: ctx != nullptr"); \
    hipError_t e_dev__ = hipSetDevice((ctx)->device);                               \
    if (e_dev__ != hipSuccess)                                                      \
      return set_err(ctx, MMG_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e_dev__)); \
  } while (0)
#define MMG_RB(ctx, call)                                                                      \
  do {                                                                                         \
    rocblas_status s__ = (call);                                                               \
    if (s__ != rocblas_status_success)                                                         \
      return set_err(ctx, MMG_E_LIB, std::string(#call) + ": rocblas status " + std::to_string((int)s__)); \
  } while (0)
#define MMG_NCCL(ctx, call)                                                                    \
  do {                                                                                         \
    ncclResult_t r__ = (call);                                                                 \
    if (r__ != ncclSuccess)                                                                    \
      return set_err(ctx, MMG_E_LIB, std::string(#call) + ": " + ncclGetErrorString(r__));     \
  } while (0)

struct mmg_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int mmg_version(void) { return 100; }

int mmg_device_count(int* n) {
  if (!n) return MMG_E_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *n = 0; return set_err(nullptr, MMG_E_HIP, hipGetErrorString(e)); }
  *n = c;
  return MMG_OK;
}

int mmg_ctx_trim(mmg_ctx* ctx) {
  MMG_ENTER(ctx);
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->band_ws);   ctx->band_ws = nullptr;   ctx->band_ws_cap = 0;
  (void)hipFree(ctx->band_keep); ctx->band_keep = nullptr; ctx->band_keep_cap = 0; ctx->band_keep_owner = nullptr;
  return MMG_OK;
}

const char* mmg_last_error(mmg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int mmg_ctx_create(int device, mmg_ctx** out) {
  if (!out) return MMG_E_ARG;
  *out = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return set_err(nullptr, MMG_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  mmg_ctx* ctx = new mmg_ctx();
  ctx->device = device;
  MMG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  MMG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  MMG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_snap, hipEventDisableTiming));
  MMG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_deliver, hipEventDisableTiming));
  for (int i = 0; i < EV_COUNT; ++i) {
    MMG_HIP(ctx, hipEventCreate(&ctx->ev[i][0]));
    MMG_HIP(ctx, hipEventCreate(&ctx->ev[i][1]));
    ctx->ev_set[i] = false;
  }
  hipDeviceProp_t prop;
  MMG_HIP(ctx, hipGetDeviceProperties(&prop, device));
  ctx->n_cu = prop.multiProcessorCount;
  *out = ctx;
  return MMG_OK;
}

static void free_model(mmg_scan_model& m) {
  hipFree(m.Bq); hipFree(m.A64); hipFree(m.diag); hipFree(m.w); hipFree(m.job_off); hipFree(m.jobs); hipFree(m.lin_tab);
  hipFree(m.job_off_hi); hipFree(m.jobs_hi); hipFree(m.job_off_lo); hipFree(m.jobs_lo);
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 2; ++k) { hipFree(m.tail_off[r][k]); hipFree(m.tail_jobs[r][k]); }
  m = mmg_scan_model();
}
static void free_result(mmg_scan_result& r) {
  hipFree(r.q); hipFree(r.rss); hipFree(r.F); hipFree(r.p); hipFree(r.dot); hipFree(r.den); hipFree(r.sum);
  hipFree(r.dd); hipFree(r.ssq); hipFree(r.idx); hipFree(r.scal); hipFree(r.q2); hipFree(r.linraw); hipFree(r.linraw2);
  r = mmg_scan_result();
}

int mmg_ctx_destroy(mmg_ctx* ctx) {
  if (!ctx) return MMG_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  hipStreamSynchronize(ctx->stream2);
  free_model(ctx->model);
  free_result(ctx->res);
  if (ctx->sel_geno) { hipFree(ctx->sel_geno->d); hipFree(ctx->sel_geno->bits); hipFree(ctx->sel_geno->hi2); hipFree(ctx->sel_geno->fp4); hipFree(ctx->sel_geno->d_smax); delete ctx->sel_geno; }
  hipFree(ctx->dstage);
  hipFree(ctx->grp_tab);
  hipFree(ctx->jobs);
  hipFree(ctx->ingest);
  if (ctx->rocblas) rocblas_destroy_handle((rocblas_handle)ctx->rocblas);
  (void)hipFree(ctx->band_ws);
  (void)hipFree(ctx->band_keep);
  for (int i = 0; i < EV_COUNT; ++i) { hipEventDestroy(ctx->ev[i][0]); hipEventDestroy(ctx->ev[i][1]); }
  hipEventDestroy(ctx->ev_snap);
  hipEventDestroy(ctx->ev_deliver);
  hipStreamDestroy(ctx->stream2);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return MMG_OK;
}

int mmg_device_info(mmg_ctx* ctx, char* name, int name_len, int* n_cu, int64_t* hbm_bytes) {
  MMG_ENTER(ctx);
  hipDeviceProp_t prop;
  MMG_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_len > 0) { std::strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return MMG_OK;
}

int mmg_device_pci_bus_id(mmg_ctx* ctx, char* buf, int buf_len) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, buf != nullptr && buf_len >= 16);
  MMG_HIP(ctx, hipDeviceGetPCIBusId(buf, buf_len, ctx->device));
  return MMG_OK;
}

int mmg_last_kernel_ms(mmg_ctx* ctx, int which, double* ms) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, which >= 0 && which < EV_COUNT && ms != nullptr);
  if (which == EV_GRM) {                                   // summed digit-plane GEMMs of the last exact GRM pass
    if (ctx->grm_ms_total <= 0.0) return set_err(ctx, MMG_E_STATE, "no exact GRM pass has run");
    *ms = ctx->grm_ms_total;
    return MMG_OK;
  }
  if (!ctx->ev_set[which]) return set_err(ctx, MMG_E_STATE, "no kernel of that kind has run");
  float f = 0.f;
  MMG_HIP(ctx, hipEventSynchronize(ctx->ev[which][1]));
  MMG_HIP(ctx, hipEventElapsedTime(&f, ctx->ev[which][0], ctx->ev[which][1]));
  *ms = (double)f;
  if (which == EV_MULTI && ctx->multi_ms_total > 0.0) *ms = ctx->multi_ms_total;   // all batches of the last call

  if (which == EV_QUAD && ctx->ev_set[EV_QUAD2]) {       // adaptive scan: the refinement pass counts too
    MMG_HIP(ctx, hipEventSynchronize(ctx->ev[EV_QUAD2][1]));
    MMG_HIP(ctx, hipEventElapsedTime(&f, ctx->ev[EV_QUAD2][0], ctx->ev[EV_QUAD2][1]));
    *ms += (double)f;
  }
  return MMG_OK;
}

int mmg_host_pin(mmg_ctx* ctx, void* p, int64_t bytes) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, p != nullptr && bytes > 0);
  MMG_HIP(ctx, hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault));
  return MMG_OK;
}

int mmg_host_alloc(mmg_ctx* ctx, int64_t bytes, void** p) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, p != nullptr && bytes > 0);
  *p = nullptr;
  MMG_HIP(ctx, hipHostMalloc(p, (size_t)bytes, hipHostMallocDefault));
  return MMG_OK;
}

int mmg_host_free(mmg_ctx* ctx, void* p) {
  MMG_ENTER(ctx);
  if (!p) return MMG_OK;
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MMG_HIP(ctx, hipHostFree(p));
  return MMG_OK;
}

int mmg_host_unpin(mmg_ctx* ctx, void* p) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, p != nullptr);
  MMG_HIP(ctx, hipHostUnregister(p));
  return MMG_OK;
}

// ------------------------------------------------------------------------- genotype store
static int geno_create(mmg_ctx* ctx, int64_t M, int32_t N, mmg_geno** out, bool twin);
int mmg_geno_create(mmg_ctx* ctx, int64_t M, int32_t N, mmg_geno** out) { return geno_create(ctx, M, N, out, true); }

// twin = false: stores the library keeps for itself and writes with kernels that do not maintain the FP4 twin (the
// compact store of the refined SNPs of an adaptive scan)
static int geno_create(mmg_ctx* ctx, int64_t M, int32_t N, mmg_geno** out, bool twin) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, out != nullptr && M >= 0 && N > 0);
  *out = nullptr;
  mmg_geno* g = new mmg_geno();
  g->M = M; g->N = N;
  g->Mpad = std::max<int64_t>(round_up(M, 256), 256);
  g->Mcap = g->Mpad;
  g->Npad = (int32_t)round_up(N, 256);
  hipError_t e = hipMalloc(&g->d, (size_t)g->Mpad * g->Npad);
  if (e != hipSuccess) { delete g; return set_err(ctx, MMG_E_NOMEM, std::string("hipMalloc genotype store: ") + hipGetErrorString(e)); }
  e = hipMalloc(&g->d_smax, 2 * sizeof(int));
  if (e != hipSuccess) { hipFree(g->d); delete g; return set_err(ctx, MMG_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  MMG_HIP(ctx, hipMemsetAsync(g->d_smax, 0, 2 * sizeof(int), ctx->stream));
  MMG_HIP(ctx, hipMemsetAsync(g->d, 0, (size_t)g->Mpad * g->Npad, ctx->stream));
  // E2M1 twin for the FP4 kinship GEMM: half the store's bytes again, kept in step by the write paths (mmg_internal.h).
  // A failed allocation is not an error: the kinship call then writes its own scratch image.
  static const bool twin_off = [] { const char* e = std::getenv("MMG_FP4_TWIN"); return e && e[0] == '0'; }();
  if (twin && !twin_off) {
    if (hipMalloc(&g->fp4, (size_t)g->Mpad * (g->Npad / 2)) == hipSuccess)
      MMG_HIP(ctx, hipMemsetAsync(g->fp4, 0, (size_t)g->Mpad * (g->Npad / 2), ctx->stream));
    else { g->fp4 = nullptr; (void)hipGetLastError(); }
  }
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = g;
  return MMG_OK;
}

int mmg_geno_reset(mmg_ctx* ctx, mmg_geno* g, int64_t M) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && M >= 0);
  const int64_t Mpad = std::max<int64_t>(round_up(M, 256), 256);
  if (Mpad > g->Mcap) return set_err(ctx, MMG_E_ARG, "mmg_geno_reset: M exceeds the capacity the store was created with");
  // rows [M, Mpad) must read as zeros (every kernel walks whole 256-row blocks); the columns beyond N of the rows an
  // upload rewrites stay zero because uploads only touch the first N bytes of a row
  if (Mpad > M) {
    MMG_HIP(ctx, hipMemsetAsync(g->d + M * (int64_t)g->Npad, 0, (size_t)(Mpad - M) * g->Npad, ctx->stream));
    if (g->fp4) MMG_HIP(ctx, hipMemsetAsync(g->fp4 + M * (int64_t)(g->Npad / 2), 0, (size_t)(Mpad - M) * (g->Npad / 2), ctx->stream));
  }
  MMG_HIP(ctx, hipMemsetAsync(g->d_smax, 0, 2 * sizeof(int), ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  g->M = M; g->Mpad = Mpad; g->smax = 0; g->sneg = 0; g->bits_valid = false; ++g->version;
  return MMG_OK;
}

int mmg_geno_destroy(mmg_ctx* ctx, mmg_geno* g) {
  if (!g) return MMG_OK;
  MMG_NOTE_ENTRY();
  if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
  hipFree(g->d);
  hipFree(g->bits);
  hipFree(g->hi2);
  hipFree(g->fp4);
  hipFree(g->d_smax);
  delete g;
  return MMG_OK;
}

// device staging of the upload / download paths that repack rows; MMG_INGEST_STAGE_MB overrides (A/B runs)
static const int64_t INGEST_STAGE_BYTES = [] { const char* e = std::getenv("MMG_INGEST_STAGE_MB"); const long v = e ? std::atol(e) : 0; return (int64_t)(v > 0 ? v : 256) << 20; }();

static int ensure_ingest(mmg_ctx* ctx, size_t bytes) {
  if (ctx->ingest_cap >= bytes) return MMG_OK;
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->ingest); ctx->ingest = nullptr; ctx->ingest_cap = 0;
  MMG_HIP(ctx, hipMalloc(&ctx->ingest, bytes));
  ctx->ingest_cap = bytes;
  return MMG_OK;
}


// The strided hipMemcpy2DAsync pays per ROW whenever the row width is not a multiple of four bytes: 7.3 us each, from 199 to
// 20,001 individuals alike -- 0.03 / 0.67 / 2.6 GB/s at N = 199 / 4999 / 20,001, 1.5 s for the 43 MB of the bundled A. thaliana
// set (199 x 214,000; tools/upload_width_check.py, profiles/r4_upload_width_check.txt).  Round 3 measured it at N = 5000
// only, where it runs at the link's 54 GB/s.  Contiguous copy into device staging + pitch_rows_kernel: 43-54 GB/s at every
// width in isolation -- but 30 GB/s in bench.py's ingest record (N = 5000, after a download through the same staging buffer and
// the release of a 7.5 GB store; seen twice, not reproduced at N = 4999 on its own), where the single strided call keeps 55.
// So: rows of a multiple of four bytes, at least UPLOAD_2D_MIN_ROW wide (28 GB/s at 200 bytes, 50 at 1000) take the
// strided copy, everything else the staging buffer.  MMG_UPLOAD_PATH=2d / staged force either.
constexpr int32_t UPLOAD_2D_MIN_ROW = 1000;
static bool upload_2d(int32_t row_bytes) {
  static const int forced = [] { const char* e = std::getenv("MMG_UPLOAD_PATH"); return !e ? 0 : std::string(e) == "staged" ? 1 : std::string(e) == "2d" ? 2 : 0; }();
  return forced == 2 || (forced == 0 && row_bytes % 4 == 0 && row_bytes >= UPLOAD_2D_MIN_ROW);
}

// every write path ends here: fold max |s| of the written rows into the store's running bound
static int refresh_smax(mmg_ctx* ctx, mmg_geno* g, int64_t m0, int64_t rows) {
  launch_absmax_i8(ctx, g->d + m0 * (int64_t)g->Npad, rows * (int64_t)g->Npad, g->d_smax,
                   g->fp4 ? g->fp4 + m0 * (int64_t)(g->Npad / 2) : nullptr);     // ... and keep the FP4 twin of these rows in step
  MMG_HIP(ctx, hipGetLastError());
  int v[2] = {0, 0};
  MMG_HIP(ctx, hipMemcpyAsync(v, g->d_smax, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  g->smax = std::max(g->smax, v[0]);
  g->sneg = std::max(g->sneg, v[1]);
  return MMG_OK;
}

int mmg_geno_upload(mmg_ctx* ctx, mmg_geno* g, const int8_t* snps, int64_t m0, int64_t rows) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && snps && m0 >= 0 && rows >= 0 && m0 + rows <= g->M);
  if (rows == 0) return MMG_OK;
  g->bits_valid = false; ++g->version;
  // individual counts that are multiples of 16 land in place; others by the strided copy or through device staging (see upload_2d)
  if (g->N == g->Npad) {
    MMG_HIP(ctx, hipMemcpyAsync(g->d + m0 * (int64_t)g->Npad, snps, (size_t)rows * g->N, hipMemcpyHostToDevice, ctx->stream));
  } else if (upload_2d(g->N)) {
    MMG_HIP(ctx, hipMemcpy2DAsync(g->d + m0 * (int64_t)g->Npad, g->Npad, snps, g->N, g->N, rows,
                                  hipMemcpyHostToDevice, ctx->stream));
  } else {
    const int64_t chunk = std::max<int64_t>(1, INGEST_STAGE_BYTES / g->N);
    int rc0 = ensure_ingest(ctx, (size_t)std::min(chunk, rows) * g->N + 32);
    if (rc0) return rc0;
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
      const int64_t nr = std::min(chunk, rows - r0);
      MMG_HIP(ctx, hipMemcpyAsync(ctx->ingest, snps + r0 * g->N, (size_t)nr * g->N, hipMemcpyHostToDevice, ctx->stream));
      launch_pitch_rows(ctx, (const int8_t*)ctx->ingest, g->d + (m0 + r0) * (int64_t)g->Npad, nr, g->N, g->Npad);
      MMG_HIP(ctx, hipGetLastError());
      if (r0 + chunk < rows) MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the staging is reused by the next piece
    }
  }
  const int prev[2] = {g->smax, g->sneg};
  int rc = refresh_smax(ctx, g, m0, rows);
  if (rc) return rc;
  if (g->smax > 127) {
    // -128 has no negation in int8.  The running bounds are maxima, so a bad block must not stay behind: zero the rows
    // just written and put the bounds (host and device copies) back, or every later upload into this store would fail
    // the same check on valid data (advisor r2).
    MMG_HIP(ctx, hipMemsetAsync(g->d + m0 * (int64_t)g->Npad, 0, (size_t)rows * g->Npad, ctx->stream));
    if (g->fp4) MMG_HIP(ctx, hipMemsetAsync(g->fp4 + m0 * (int64_t)(g->Npad / 2), 0, (size_t)rows * (g->Npad / 2), ctx->stream));
    MMG_HIP(ctx, hipMemcpyAsync(g->d_smax, prev, 2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    g->smax = prev[0]; g->sneg = prev[1];
    return set_err(ctx, MMG_E_ARG, "genotype value -128 is outside the store's range [-127, 127]; the rows of this "
                                   "upload were zeroed");
  }
  return MMG_OK;
}

// Packed ingest: 1 bit (0/1) or 2 bits (0..3, e.g. 0/1/2) per genotype cross PCIe / come off the disk instead of a
// byte each -- the host link is the roof of every run whose genotypes are not resident (64 GB/s = 12.8 M SNPs/s at
// N = 5000 for int8 rows; 8x / 4x that packed).  The rows are expanded into the int8 store on the device.
int mmg_geno_upload_packed(mmg_ctx* ctx, mmg_geno* g, const uint8_t* packed, int64_t m0, int64_t rows, int32_t bits,
                           int64_t row_bytes, const int8_t* lut) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && packed && m0 >= 0 && rows >= 0 && m0 + rows <= g->M && (bits == 1 || bits == 2));
  MMG_CHECK_ARG(ctx, row_bytes >= ((int64_t)g->N * bits + 7) / 8);
  if (rows == 0) return MMG_OK;
  uint32_t lut32 = bits == 1 ? 0x00000100u : 0x03020100u;               // identity: code = value
  if (lut) {
    lut32 = 0;
    for (int c = 0; c < (1 << bits); ++c) {
      if (lut[c] == -128) return set_err(ctx, MMG_E_ARG, "mmg_geno_upload_packed: lut values must lie in [-127, 127]");
      lut32 |= (uint32_t)(uint8_t)lut[c] << (8 * c);
    }
  }
  g->bits_valid = false; ++g->version;
  const int64_t chunk = std::max<int64_t>(1, INGEST_STAGE_BYTES / row_bytes);
  int rc0 = ensure_ingest(ctx, (size_t)std::min(chunk, rows) * row_bytes);
  if (rc0) return rc0;
  uint8_t* tmp = (uint8_t*)ctx->ingest;
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = std::min(chunk, rows - r0);
    MMG_HIP(ctx, hipMemcpyAsync(tmp, packed + r0 * row_bytes, (size_t)nr * row_bytes, hipMemcpyHostToDevice, ctx->stream));
    launch_unpack(ctx, tmp, row_bytes, g->d + (m0 + r0) * (int64_t)g->Npad, nr, g->N, g->Npad, bits, lut32,
                  g->fp4 ? g->fp4 + (m0 + r0) * (int64_t)(g->Npad / 2) : nullptr);
    MMG_HIP(ctx, hipGetLastError());
    if (r0 + chunk < rows) MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // tmp is reused by the next piece
  }
  if (bits == 1) {
    // two codes: the bounds follow from the lut (a pass over the expanded rows would re-read N bytes per SNP); an
    // all-zero block leaves them one too high, which only costs the fast paths that ask for smax == 0 -- none does
    const int a = (int)(int8_t)(lut32 & 0xff), b = (int)(int8_t)((lut32 >> 8) & 0xff);
    g->smax = std::max(g->smax, std::max(std::abs(a), std::abs(b)));
    g->sneg = std::max(g->sneg, std::max(-a, -b));
    const int v[2] = {g->smax, g->sneg};
    MMG_HIP(ctx, hipMemcpyAsync(g->d_smax, v, 2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMG_OK;
  }
  return refresh_smax(ctx, g, m0, rows);
}

extern "C++" {
template <typename T>
static int upload_cvt(mmg_ctx* ctx, mmg_geno* g, const T* snps, int64_t m0, int64_t rows) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && snps && m0 >= 0 && rows >= 0 && m0 + rows <= g->M);
  g->bits_valid = false; ++g->version;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(256 << 20) / ((int64_t)g->N * sizeof(T)));
  T* tmp = nullptr;
  int* dbad = nullptr;
  if (rows == 0) return MMG_OK;
  MMG_HIP(ctx, sc.alloc(&tmp, (size_t)std::min(chunk, rows) * g->N * sizeof(T)));
  MMG_HIP(ctx, sc.alloc(&dbad, sizeof(int)));
  MMG_HIP(ctx, hipMemsetAsync(dbad, 0, sizeof(int), ctx->stream));
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = std::min(chunk, rows - r0);
    MMG_HIP(ctx, hipMemcpyAsync(tmp, snps + r0 * g->N, (size_t)nr * g->N * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    if (sizeof(T) == 4) launch_cvt_f32(ctx, (const float*)tmp, g->d + (m0 + r0) * (int64_t)g->Npad, nr, g->N, g->Npad, dbad);
    else launch_cvt_f64(ctx, (const double*)tmp, g->d + (m0 + r0) * (int64_t)g->Npad, nr, g->N, g->Npad, dbad);
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  int bad = 0;
  MMG_HIP(ctx, hipMemcpy(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost));
  int rc = refresh_smax(ctx, g, m0, rows);
  if (rc) return rc;
  if (bad)
    return set_err(ctx, MMG_E_ARG, "genotype values must be integers in [-127, 127] (the store is int8); "
                                   "non-integral, out-of-range or NaN values were found and left unwritten");
  return MMG_OK;
}
}  // extern C++
int mmg_geno_upload_f32(mmg_ctx* ctx, mmg_geno* g, const float* snps, int64_t m0, int64_t rows) {
  return upload_cvt<float>(ctx, g, snps, m0, rows);
}
int mmg_geno_upload_f64(mmg_ctx* ctx, mmg_geno* g, const double* snps, int64_t m0, int64_t rows) {
  return upload_cvt<double>(ctx, g, snps, m0, rows);
}

int mmg_geno_download(mmg_ctx* ctx, mmg_geno* g, int8_t* snps, int64_t m0, int64_t rows) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && snps && m0 >= 0 && rows >= 0 && m0 + rows <= g->M);
  if (rows == 0) return MMG_OK;
  if (g->N == g->Npad) {
    MMG_HIP(ctx, hipMemcpyAsync(snps, g->d + m0 * (int64_t)g->Npad, (size_t)rows * g->N, hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMG_OK;
  }
  const int64_t chunk = std::max<int64_t>(1, INGEST_STAGE_BYTES / g->N);   // rows packed on the device, one transfer per piece
  int rc0 = ensure_ingest(ctx, (size_t)std::min(chunk, rows) * g->N + 32);
  if (rc0) return rc0;
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = std::min(chunk, rows - r0);
    launch_unpitch_rows(ctx, g->d + (m0 + r0) * (int64_t)g->Npad, (int8_t*)ctx->ingest, nr, g->N, g->Npad);
    MMG_HIP(ctx, hipGetLastError());
    MMG_HIP(ctx, hipMemcpyAsync(snps + r0 * g->N, ctx->ingest, (size_t)nr * g->N, hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return MMG_OK;
}

int mmg_geno_download_rows(mmg_ctx* ctx, mmg_geno* g, const int64_t* idx, int64_t cnt, int8_t* snps) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && idx && snps && cnt >= 0);
  if (cnt == 0) return MMG_OK;
  for (int64_t i = 0; i < cnt; ++i) MMG_CHECK_ARG(ctx, idx[i] >= 0 && idx[i] < g->M);
  int64_t* didx = nullptr;
  int8_t* dS = nullptr;
  MMG_HIP(ctx, sc.alloc(&didx, cnt * sizeof(int64_t)));
  MMG_HIP(ctx, sc.alloc(&dS, (size_t)cnt * g->Npad));
  MMG_HIP(ctx, hipMemcpyAsync(didx, idx, cnt * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  launch_gather_rows(ctx, g, didx, cnt, dS);
  MMG_HIP(ctx, hipGetLastError());
  int8_t* dP = dS;
  if (g->N != g->Npad) {
    MMG_HIP(ctx, sc.alloc(&dP, (size_t)cnt * g->N));
    launch_unpitch_rows(ctx, dS, dP, cnt, g->N, g->Npad);
    MMG_HIP(ctx, hipGetLastError());
  }
  MMG_HIP(ctx, hipMemcpyAsync(snps, dP, (size_t)cnt * g->N, hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int mmg_geno_fill_hash(mmg_ctx* ctx, mmg_geno* g, uint64_t seed, int64_t m_global0, uint32_t thr16) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g != nullptr && thr16 <= 65536);
  if (g->M == 0) return MMG_OK;
  g->bits_valid = false; ++g->version;
  {
    EvScope ev(ctx, EV_PACK);
    launch_fill_hash(ctx, g, seed, m_global0, thr16);
  }
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  g->smax = std::max(g->smax, 1);                     // the generator writes 0 / 1
  return MMG_OK;
}

int mmg_geno_fill_structured(mmg_ctx* ctx, mmg_geno* g, uint64_t seed, int64_t m_global0, int32_t npop,
                             uint32_t spread_q16) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g != nullptr && npop >= 1 && npop <= 64 && spread_q16 <= 65536);
  if (g->M == 0) return MMG_OK;
  g->bits_valid = false; ++g->version;
  {
    EvScope ev(ctx, EV_PACK);
    launch_fill_struct(ctx, g, seed, m_global0, npop, spread_q16);
  }
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  g->smax = std::max(g->smax, 1);
  return MMG_OK;
}

int mmg_geno_snp_stats(mmg_ctx* ctx, mmg_geno* g, double* mean, double* sd) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && mean && sd);
  if (g->M == 0) return MMG_OK;
  double *dm = nullptr, *ds = nullptr;
  MMG_HIP(ctx, sc.alloc(&dm, g->M * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&ds, g->M * sizeof(double)));
  launch_snp_stats(ctx, g, dm, ds);
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipMemcpyAsync(mean, dm, g->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipMemcpyAsync(sd, ds, g->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

// ------------------------------------------------------------------------- kinship
// SNP rows per kinship pass: bounds the transposed image (N x chunk bytes) and keeps 256*ld below
// the 32-bit buffer-offset limit of the staging loads (gemm_i8_core.h MAX_LD).
static int64_t kin_chunk() {
  int64_t ch = int64_t(4) << 20;
  if (const char* e = std::getenv("MMG_KIN_CHUNK")) ch = std::max<int64_t>(128, round_up(std::atoll(e), 128));
  return std::min<int64_t>(ch, MAX_LD / 128 * 128);
}

// f64: the IBS kinship itself instead of the counts (mmg_kinship_ibs_f64): K = counts / (2 m_total) + 0.5, scale_k on request
struct IbsF64 { double* K_out; int64_t m_total; bool scaled; int64_t* dev64 = nullptr; double* dK_dev = nullptr; };   // dev64: the counts stay in HBM (device buffer); dK_dev: the kinship does (mmg_kin_acc_set_ibs)
static int kinship_counts_i8(mmg_ctx* ctx, mmg_geno* g, int mul, int add, int thr, int64_t* C_out,
                             mmg_comm* comm = nullptr, const IbsF64* f64 = nullptr);

int mmg_kinship_ibs_i8(mmg_ctx* ctx, mmg_geno* g, int64_t* C_out) {
  return kinship_counts_i8(ctx, g, 2, -1, 0, C_out);    // X = 2S - 1 (kinship.py:43)
}

int mmg_kinship_ibs_i8_sharded(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int64_t* C_out) {
  return kinship_counts_i8(ctx, g, 2, -1, 0, C_out, comm);
}

int mmg_kinship_indicator_i8(mmg_ctx* ctx, mmg_geno* g, int32_t thr, int64_t* C_out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, thr >= 1 && thr <= 127);
  return kinship_counts_i8(ctx, g, 0, 0, thr, C_out);   // X = [S >= thr]
}

static int kinship_counts_i8(mmg_ctx* ctx, mmg_geno* g, int mul, int add, int thr, int64_t* C_out, mmg_comm* comm, const IbsF64* f64) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && (C_out || (f64 && (f64->K_out || f64->dev64 || f64->dK_dev))) && g->M > 0);
  // IBS (X = 2S - 1, kinship.py:43) runs on the RAW genotypes: X X' = 4 S S' - 2 (r 1' + 1 r') + M with r the column
  // sums of S -- the same exact integers, but the GEMM operands are 0/1 bytes instead of +-1: the matrix pipe draws
  // less power on mostly-zero operands and the power-limited chip clocks higher (measured at N = 5000, M = 1e6:
  // 9.2 ms against 11.1 ms for the +-1 operands, same kernel).
  const bool ibs = (mul == 2 && add == -1 && thr == 0);
  const int64_t vmax = ibs ? std::max(g->smax, 1) : (thr > 0 ? 1 : 127);
  MMG_CHECK_ARG(ctx, vmax * vmax * g->M < (int64_t(1) << 31));   // int32 accumulators: |C_ij| <= max|x|^2 M
  const int64_t CH = kin_chunk();
  const int64_t Mk_max = std::min(round_up(g->M, BK), CH);
  // Round 3: the raw-genotype product reads the SNP-major store as it lies (gemm_i8_w4tr.h: transposed LDS reads) --
  // no individual-major image, no transposition pass.  MMG_KIN_KERNEL=w4 / w8: the transposed-image generations.
  static const bool tr_off = [] { const char* e = std::getenv("MMG_KIN_KERNEL"); return e && (std::string(e) == "w4" || std::string(e) == "w8"); }();
  // 0/1 operands -- the raw genotypes of a binary store, or an indicator [s >= thr] of any store -- run on FP4 (twice
  // the MACs per MFMA, half the bytes per LDS fill); MMG_KIN_FP4=0 keeps int8
  static const bool fp4_off = [] { const char* e = std::getenv("MMG_KIN_FP4"); return e && e[0] == '0'; }();
  const bool binary = g->smax <= 1 && g->sneg == 0;
  const bool fp4 = !fp4_off && !tr_off && ((ibs && binary) || thr > 0);
  bool direct = (ibs && !tr_off) || fp4;
  int8_t* Xt = nullptr;
  int* C32 = nullptr;
  int64_t* C64 = nullptr;
  MMG_HIP(ctx, sc.alloc(&C32, (size_t)g->Npad * g->Npad * sizeof(int)));
  MMG_HIP(ctx, sc.alloc(&C64, (size_t)g->N * g->N * sizeof(int64_t)));
  MMG_HIP(ctx, hipMemsetAsync(C32, 0, (size_t)g->Npad * g->Npad * sizeof(int), ctx->stream));
  int rc = MMG_OK;
  double kin_ms = 0.0, pack_ms = 0.0;
  if (direct) {
    rc = MMG_E_STATE;
    if (fp4 && ibs && binary && g->fp4) {
      // the store's own E2M1 twin (kept in step by the write paths): no image pass at all
      rc = MMG_OK;
      {
        EvScope ev(ctx, EV_KIN);
        rc = run_kinship_f4_tr(ctx, sc, g->fp4, g->Npad, g->Mpad / 256, C32);
      }
      MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ctx->ev_set[EV_PACK] = false;
      if (rc == MMG_E_STATE) MMG_HIP(ctx, hipMemsetAsync(C32, 0, (size_t)g->Npad * g->Npad * sizeof(int), ctx->stream));
    } else if (fp4) {
      uint8_t* X4 = nullptr;
      if (sc.alloc(&X4, (size_t)g->Mpad * (g->Npad / 2)) == hipSuccess) {
        // The nibble image can be written in SNP chunks on the second stream while the GEMM of the previous chunk runs on
        // the first (MMG_KIN_FP4_CHUNKS; the GEMM's waves leave registers and LDS for the image kernel's).  Measured at
        // C3: 1 chunk 6.66 ms (1.3 + 5.4), 2: 6.67, 4: 7.3, 8: 8.7 -- the passes do not hide behind a GEMM that sits at
        // the power cap, and shorter contraction ranges cost more than the overlap returns.  Default: one chunk.
        const int64_t ld4 = g->Npad / 2;
        int nch = 1;
        if (const char* e = std::getenv("MMG_KIN_FP4_CHUNKS")) nch = std::max(1, std::min(16, std::atoi(e)));
        const int64_t rows_ch = round_up((g->Mpad + nch - 1) / nch, 256);
        struct EvGuard {                                   // destroyed on every path out of this block (advisor r3)
          std::vector<hipEvent_t> v;
          hipEvent_t first = nullptr;
          ~EvGuard() { for (hipEvent_t e : v) (void)hipEventDestroy(e); if (first) (void)hipEventDestroy(first); }
        } guard;
        std::vector<hipEvent_t>& evs = guard.v;
        hipEvent_t& ev0 = guard.first;
        MMG_HIP(ctx, hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
        MMG_HIP(ctx, hipEventRecord(ev0, ctx->stream));                  // the store's writers are on the first stream
        MMG_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ev0, 0));
        if (nch == 1) MMG_HIP(ctx, hipEventRecord(ctx->ev[EV_PACK][0], ctx->stream2));   // the image pass on its own
        for (int64_t r0 = 0; r0 < g->Mpad; r0 += rows_ch) {
          const int64_t nr = std::min(rows_ch, g->Mpad - r0);
          launch_pack_fp4_on(ctx, ctx->stream2, g->d + r0 * (int64_t)g->Npad, nr, g->Npad, X4 + r0 * ld4,
                             thr > 0 ? thr : 1, binary);
          hipEvent_t ev = nullptr;
          MMG_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
          MMG_HIP(ctx, hipEventRecord(ev, ctx->stream2));
          evs.push_back(ev);
        }
        if (nch == 1) MMG_HIP(ctx, hipEventRecord(ctx->ev[EV_PACK][1], ctx->stream2));
        rc = MMG_OK;
        {
          EvScope ev(ctx, EV_KIN);                                        // inclusive: image chunks + GEMMs
          size_t c = 0;
          for (int64_t r0 = 0; r0 < g->Mpad && rc == MMG_OK; r0 += rows_ch, ++c) {
            const int64_t nr = std::min(rows_ch, g->Mpad - r0);
            MMG_HIP(ctx, hipStreamWaitEvent(ctx->stream, evs[c], 0));
            rc = run_kinship_f4_tr(ctx, sc, X4 + r0 * ld4, g->Npad, nr / 256, C32);
          }
        }
        hipError_t es = hipStreamSynchronize(ctx->stream);
        if (es == hipSuccess) es = hipStreamSynchronize(ctx->stream2);
        if (es != hipSuccess) return set_err(ctx, MMG_E_HIP, hipGetErrorString(es));
        ctx->ev_set[EV_PACK] = (nch == 1);                                // its time is inside EV_KIN as well
        if (rc == MMG_E_STATE) MMG_HIP(ctx, hipMemsetAsync(C32, 0, (size_t)g->Npad * g->Npad * sizeof(int), ctx->stream));
      } else {
        (void)hipGetLastError();
      }
    }
    if (rc == MMG_E_STATE && ibs && !tr_off) {
      ctx->ev_set[EV_PACK] = false;                     // no image pass in this call
      rc = run_kinship_i8_tr(ctx, g->d, g->d, g->Npad, g->Npad, g->Mpad / BK, C32);   // rows M..Mpad are zero
    } else if (rc == MMG_E_STATE) {
      direct = false;                                   // an indicator product beyond the FP4 range: transposed image
      rc = MMG_OK;
    }
  }
  if (!direct) {
    hipError_t e = sc.alloc(&Xt, (size_t)g->Npad * Mk_max);
    if (e != hipSuccess) return set_err(ctx, MMG_E_NOMEM, "hipMalloc transposed genotype image");
  }
  for (int64_t mb = 0; !direct && mb < g->M && rc == MMG_OK; mb += CH) {
    const int64_t Mk = round_up(std::min(CH, g->M - mb), BK);
    {
      EvScope ev(ctx, EV_PACK);
      launch_transpose(ctx, g, Xt, Mk, ibs ? 1 : mul, ibs ? 0 : add, mb, thr);   // zero in the padding
    }
    MMG_HIP(ctx, hipGetLastError());
    rc = run_kinship_i8(ctx, Xt, g->Npad, Mk, C32);
    if (rc == MMG_OK && g->M > CH) {                    // several passes: report the summed kernel time
      double a = 0, b = 0;
      mmg_last_kernel_ms(ctx, EV_KIN, &a); mmg_last_kernel_ms(ctx, EV_PACK, &b);
      kin_ms += a; pack_ms += b;
    }
  }
  (void)kin_ms; (void)pack_ms;
  if (rc == MMG_OK) {
    if (ibs) {
      unsigned long long* dr = nullptr;
      MMG_HIP(ctx, sc.alloc(&dr, (size_t)g->Npad * sizeof(unsigned long long)));
      MMG_HIP(ctx, hipMemsetAsync(dr, 0, (size_t)g->Npad * sizeof(unsigned long long), ctx->stream));
      launch_colsum(ctx, g, dr);
      launch_mirror_ibs(ctx, C32, g->Npad, g->N, (const long long*)dr, (long long)g->M, C64);
    } else {
      launch_mirror_i32_to_i64(ctx, C32, g->Npad, g->N, C64);
    }
    if (comm && comm->world > 1) {
      // the partial counts of this rank's SNP block never leave HBM: one in-place RCCL SUM over xGMI, one download
      ncclResult_t r = ncclAllReduce(C64, C64, (size_t)g->N * g->N, ncclInt64, ncclSum, comm->comm, ctx->stream);
      if (r != ncclSuccess) return set_err(ctx, MMG_E_LIB, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    }
    if (f64 && f64->dev64) {
      MMG_HIP(ctx, hipMemcpyAsync(f64->dev64, C64, (size_t)g->N * g->N * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
      MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return rc;
    }
    if (f64) {
      // the kinship leaves the device as the matrix the caller wants: conversion and scale_k's rule (kinship.py:94-100; the
      // sums of mmg_kin_acc_scale_k) in HBM instead of three host passes over N^2 doubles
      const int64_t N = g->N;
      double *dK = f64->dK_dev, *drow = nullptr;
      if (!dK) MMG_HIP(ctx, sc.alloc(&dK, (size_t)N * N * sizeof(double)));
      launch_ibs_counts_to_f64(ctx, C64, N * N, 2.0 * (double)f64->m_total, dK);
      if (f64->scaled) {
        MMG_HIP(ctx, sc.alloc(&drow, 2 * N * sizeof(double)));
        launch_row_sums_f64(ctx, dK, N, drow, drow + N);
        std::vector<double> hsum((size_t)2 * N);
        MMG_HIP(ctx, hipMemcpyAsync(hsum.data(), drow, 2 * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double total = 0.0, trace = 0.0;
        for (int64_t i = 0; i < N; ++i) { total += hsum[i]; trace += hsum[N + i]; }
        const double c = trace - total / (double)N;
        if (!(c > 0.0) || !std::isfinite(c)) return set_err(ctx, MMG_E_ARG, "scale_k: tr K - sum K / N is not positive");
        launch_scale_f64(ctx, dK, N * N, (double)(N - 1) / c);
      }
      if (f64->dK_dev) {                                       // stays in HBM
        MMG_HIP(ctx, hipGetLastError());
        MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return rc;
      }
      hipError_t e3 = hipMemcpyAsync(f64->K_out, dK, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
      if (e3 == hipSuccess) e3 = hipStreamSynchronize(ctx->stream);
      if (e3 != hipSuccess) rc = set_err(ctx, MMG_E_HIP, hipGetErrorString(e3));
      return rc;
    }
    hipError_t e2 = hipMemcpyAsync(C_out, C64, (size_t)g->N * g->N * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
    if (e2 != hipSuccess) rc = set_err(ctx, MMG_E_HIP, hipGetErrorString(e2));
  }
  return rc;
}

// scale_k's rule on a device-resident N x N matrix, then its download
static int scale_and_fetch(mmg_ctx* ctx, double* dK, int64_t N, bool scaled, double* K_out) {
  if (scaled) {
    Scratch sc;
    double* drow = nullptr;
    MMG_HIP(ctx, sc.alloc(&drow, 2 * N * sizeof(double)));
    launch_row_sums_f64(ctx, dK, N, drow, drow + N);
    std::vector<double> hsum((size_t)2 * N);
    MMG_HIP(ctx, hipMemcpyAsync(hsum.data(), drow, 2 * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0.0, trace = 0.0;
    for (int64_t i = 0; i < N; ++i) { total += hsum[i]; trace += hsum[N + i]; }
    const double c = trace - total / (double)N;
    if (!(c > 0.0) || !std::isfinite(c)) return set_err(ctx, MMG_E_ARG, "scale_k: tr K - sum K / N is not positive");
    launch_scale_f64(ctx, dK, N * N, (double)(N - 1) / c);
  }
  MMG_HIP(ctx, hipMemcpyAsync(K_out, dK, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

// 'diploid_int' IBS (kinship.py:33-41,51): k_ij = (M - 1/2 sum_m |a_m - b_m|) / M off the diagonal, 1 on it;
// |a - b| = a + b - 2 min(a, b), min(a, b) = [a >= 1][b >= 1] + [a >= 2][b >= 2] for 0/1/2: two exact indicator products, their
// sum c12 with diagonal r, combined and scaled in HBM
int mmg_kinship_ibs_diploid_f64(mmg_ctx* ctx, mmg_geno* g, int32_t scaled, double* K_out) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && K_out && g->M > 0);
  const int64_t N = g->N;
  int64_t *c1 = nullptr, *c2 = nullptr;
  double* dK = nullptr;
  // Round 6: both indicator images from one read of the store, stacked into ONE 2 M-row FP4 image, so that c12 = u'u + v'v is a
  // single launch of the FP4 kinship GEMM (twice the contraction length of a binary store's) instead of two image passes, two
  // GEMMs, two mirrors and two N^2 copies.  MMG_IBS_DIPLOID_FUSED=0 / a range the fp32 accumulators cannot hold: the two calls.
  static const bool fused_off = [] { const char* e = std::getenv("MMG_IBS_DIPLOID_FUSED"); return e && e[0] == '0'; }();
  static const bool fp4_off_d = [] { const char* e = std::getenv("MMG_KIN_FP4"); return e && e[0] == '0'; }();
  if (!fused_off && !fp4_off_d && g->sneg == 0 && g->smax <= 2) {
    Scratch sf;                                               // (freed when this attempt ends, whichever way)
    uint8_t* X4 = nullptr;
    int* C32 = nullptr;
    const int64_t ld4 = g->Npad / 2;
    if (sf.alloc(&X4, (size_t)2 * g->Mpad * ld4) == hipSuccess && sf.alloc(&C32, (size_t)g->Npad * g->Npad * sizeof(int)) == hipSuccess &&
        sf.alloc(&c1, (size_t)N * N * sizeof(int64_t)) == hipSuccess && sf.alloc(&dK, (size_t)N * N * sizeof(double)) == hipSuccess) {
      MMG_HIP(ctx, hipMemsetAsync(C32, 0, (size_t)g->Npad * g->Npad * sizeof(int), ctx->stream));
      {
        EvScope ev(ctx, EV_PACK);
        launch_pack_fp4_two(ctx, g->d, g->Mpad, g->Npad, X4, X4 + g->Mpad * ld4);
      }
      int rc = MMG_OK;
      {
        EvScope ev(ctx, EV_KIN);
        rc = run_kinship_f4_tr(ctx, sf, X4, g->Npad, 2 * g->Mpad / 256, C32);
      }
      if (rc == MMG_OK) {
        launch_mirror_i32_to_i64(ctx, C32, g->Npad, g->N, c1);
        launch_ibs_diploid_combine(ctx, c1, nullptr, N, (double)g->M, dK);
        MMG_HIP(ctx, hipGetLastError());
        return scale_and_fetch(ctx, dK, N, scaled != 0, K_out);
      }
      if (rc != MMG_E_STATE) return rc;
      MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
      (void)hipGetLastError();
    }
    c1 = nullptr; dK = nullptr;
  }
  MMG_HIP(ctx, sc.alloc(&c1, (size_t)N * N * sizeof(int64_t)));
  MMG_HIP(ctx, sc.alloc(&c2, (size_t)N * N * sizeof(int64_t)));
  MMG_HIP(ctx, sc.alloc(&dK, (size_t)N * N * sizeof(double)));
  IbsF64 f{nullptr, g->M, false, c1};
  int rc = kinship_counts_i8(ctx, g, 0, 0, 1, nullptr, nullptr, &f);
  if (rc) return rc;
  f.dev64 = c2;
  rc = kinship_counts_i8(ctx, g, 0, 0, 2, nullptr, nullptr, &f);
  if (rc) return rc;
  launch_ibs_diploid_combine(ctx, c1, c2, N, (double)g->M, dK);
  MMG_HIP(ctx, hipGetLastError());
  return scale_and_fetch(ctx, dK, N, scaled != 0, K_out);
}

int mmg_kinship_ibs_f64(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int64_t m_total, int32_t scaled, double* K_out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && K_out && m_total >= g->M);
  const IbsF64 f{K_out, m_total, scaled != 0};
  return kinship_counts_i8(ctx, g, 2, -1, 0, nullptr, comm, &f);
}

// dC [N x N] (device, fp64) (+)= sum_m x_m x_m', x_m = scale[m] s_m + shift[m]
static int kinship_affine_into(mmg_ctx* ctx, mmg_geno* g, const float* scale, const float* shift, double* dC,
                               bool accumulate) {
  Scratch sc;
  const int64_t CH = kin_chunk();
  const int64_t Mk_all = round_up(g->M, BK);
  const int64_t Mk_max = std::min(Mk_all, CH);
  const int ksplit_max = kinship_pick_ksplit(g->Npad, Mk_max, true);
  int8_t* Xt = nullptr;
  float *dsc = nullptr, *dsh = nullptr, *slabs = nullptr;
  hipError_t e = sc.alloc(&Xt, (size_t)g->Npad * Mk_max);
  if (e != hipSuccess) return set_err(ctx, MMG_E_NOMEM, "hipMalloc transposed genotype image");
  MMG_HIP(ctx, sc.alloc(&dsc, Mk_all * sizeof(float)));
  MMG_HIP(ctx, sc.alloc(&dsh, Mk_all * sizeof(float)));
  MMG_HIP(ctx, sc.alloc(&slabs, (size_t)ksplit_max * g->Npad * g->Npad * sizeof(float)));
  MMG_HIP(ctx, hipMemsetAsync(dsc, 0, Mk_all * sizeof(float), ctx->stream));
  MMG_HIP(ctx, hipMemsetAsync(dsh, 0, Mk_all * sizeof(float), ctx->stream));
  if (scale) {
    MMG_HIP(ctx, hipMemcpyAsync(dsc, scale, g->M * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipMemcpyAsync(dsh, shift, g->M * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    std::vector<float> two((size_t)g->M, 2.0f), neg((size_t)g->M, -1.0f);
    MMG_HIP(ctx, hipMemcpyAsync(dsc, two.data(), g->M * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipMemcpyAsync(dsh, neg.data(), g->M * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  int rc = MMG_OK;
  for (int64_t mb = 0; mb < g->M && rc == MMG_OK; mb += CH) {
    const int64_t Mk = round_up(std::min(CH, g->M - mb), BK);
    const int ksplit = kinship_pick_ksplit(g->Npad, Mk, true);
    {
      EvScope ev(ctx, EV_PACK);
      launch_transpose(ctx, g, Xt, Mk, 1, 0, mb);
    }
    MMG_HIP(ctx, hipGetLastError());
    rc = run_kinship_f32(ctx, Xt, g->Npad, Mk, dsc + mb, dsh + mb, slabs, ksplit);
    if (rc == MMG_OK) launch_reduce_slabs(ctx, slabs, ksplit, g->Npad, g->N, dC, (accumulate || mb > 0) ? 1 : 0);
  }
  hipStreamSynchronize(ctx->stream);
  return rc;
}

int mmg_kinship_affine_f32(mmg_ctx* ctx, mmg_geno* g, const float* scale, const float* shift, double* C_out) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && C_out && g->M > 0);
  MMG_CHECK_ARG(ctx, (scale == nullptr) == (shift == nullptr));
  double* dC = nullptr;
  MMG_HIP(ctx, sc.alloc(&dC, (size_t)g->N * g->N * sizeof(double)));
  int rc = kinship_affine_into(ctx, g, scale, shift, dC, false);
  if (rc == MMG_OK) {
    hipError_t e2 = hipMemcpyAsync(C_out, dC, (size_t)g->N * g->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
    if (e2 != hipSuccess) rc = set_err(ctx, MMG_E_HIP, hipGetErrorString(e2));
  }
  return rc;
}

// Device-resident kinship accumulator for chunked / streamed genotypes (the `k_mat += x'x` loop of
// hdf5_data.py:99-106 and kinship.py:63-69): the N x N sum stays in HBM between chunks.
// workspace of the exact GRM route, kept between chunks: hipMalloc / hipFree of tens of GB per chunk cost more than the
// GEMMs they serve (measured at N = 50,000: 1.4-1.8 s of allocation around 0.2 s of matrix-core work)
struct GrmWorkspace {
  int8_t *Xq = nullptr, *Xp = nullptr, *ddig = nullptr;
  int* C32 = nullptr;
  double *dm = nullptr, *ds = nullptr, *dcoef = nullptr, *dc1 = nullptr, *dc1acc = nullptr, *dpart = nullptr, *dwst = nullptr;
  size_t cap_img = 0, cap_c32 = 0, cap_m = 0, cap_mk = 0, cap_n = 0;
  int capD = 0, cap_nimg = 0;     // planes of C32 / images of Xp (one more than planes for a centred 0 / 1 / 2 store)
  bool direct = false;            // SNP-major images for the transposed-read GEMM (no plain image Xq)
  // Sums that have not reached the fp64 accumulator yet (round 4): successive calls whose weights fit the digit step
  // already in use keep adding into the SAME int32 planes, and the combine pass (40 GB of planes + 40 GB of fp64 read /
  // written at N = 50,000: 20 ms, plus 13 ms of clearing the planes) runs once per run of such calls, not once per call.
  bool pending = false;
  int p_D = 0, p_bd = 0;
  int p_fused = 0;                // 0: one image GEMM per plane, 1: the one-pass kernel (4 planes), 2: one-pass for planes 1-4 + image GEMM for plane 0
  double p_step = 0.0, p_wcap = 0.0, p_c0 = 0.0, p_smax = 0.0;
  int64_t p_M = 0;
  int32_t p_Npad = 0, p_N = 0;
  int64_t stream_M = 0;           // SNPs of every successful exact-GRM call so far (the dither of the digit rounding indexes the stream)
  bool in_call = false;           // a call has started writing into the planes and has not finished: see mmg_kin_acc_add_grm
  void release() {
    hipFree(Xq); hipFree(Xp); hipFree(ddig); hipFree(C32); hipFree(dm); hipFree(ds); hipFree(dcoef); hipFree(dc1); hipFree(dc1acc);
    hipFree(dpart); hipFree(dwst);
    const int64_t sm = stream_M;
    const bool ic = in_call;
    *this = GrmWorkspace();
    stream_M = sm; in_call = ic;
  }
};
struct mmg_kin_acc {
  int32_t N = 0; double* dC = nullptr; int64_t n_snps = 0; GrmWorkspace ws;
  bool broken = false;            // a call failed after it had begun to add: the sum is neither with nor without its SNPs
};
#define MMG_ACC_USABLE(ctx, a)                                                                                         \
  do {                                                                                                                 \
    if ((a)->broken)                                                                                                   \
      return set_err(ctx, MMG_E_STATE, "the kinship accumulator is unusable: an earlier mmg_kin_acc_add_grm / _set_ibs " \
                                       "failed part-way through (its matrix is half built); create a new accumulator");   \
  } while (0)

int mmg_kin_acc_create(mmg_ctx* ctx, int32_t N, mmg_kin_acc** out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, out && N > 0);
  mmg_kin_acc* a = new mmg_kin_acc();
  a->N = N;
  hipError_t e = hipMalloc(&a->dC, (size_t)N * N * sizeof(double));
  if (e != hipSuccess) { delete a; return set_err(ctx, MMG_E_NOMEM, "hipMalloc kinship accumulator"); }
  MMG_HIP(ctx, hipMemsetAsync(a->dC, 0, (size_t)N * N * sizeof(double), ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = a;
  return MMG_OK;
}

int mmg_kin_acc_add(mmg_ctx* ctx, mmg_kin_acc* a, mmg_geno* g, const float* scale, const float* shift) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a && g && g->N == a->N && (scale == nullptr) == (shift == nullptr));
  MMG_ACC_USABLE(ctx, a);
  if (g->M == 0) return MMG_OK;
  int rc = kinship_affine_into(ctx, g, scale, shift, a->dC, true);
  if (rc == MMG_OK) a->n_snps += g->M;
  return rc;
}

// Exact GRM (kinship.py:63-69, hdf5_data.py:99-106): K (+)= sum_m z_m z_m', z_m = (s_m - mean_m) / std_m.
//     z z' = a^2 s s' + a b (s 1' + 1 s') + b^2 1 1',   a = 1/std, b = -mean/std
// The first term is a Gram matrix weighted per SNP by omega_m = 1/std_m^2.  omega is written as D non-negative digits
// whose range makes digit * s fit int8 (7 bits for alphabets within +-1, 6 within +-2, 5 within +-4); each digit plane is then
// ONE exact int8-MFMA GEMM of the IBS kind (digit image x plain image, upper tiles only) -- 4-5 planes at 32x the
// fp32-MFMA rate instead of the fp32 GEMM, and entries good to ~1e-9 instead of fp32 products.  The rank-one terms
// are two dot-product passes in fp64.  Returns MMG_E_STATE (caller falls back to the fp32 kernel) when the genotype
// alphabet is too wide for int8 digit products.
// dC += the sums the int32 planes hold (see GrmWorkspace::pending)
static int grm_flush(mmg_ctx* ctx, GrmWorkspace& ws, double* dC) {
  if (!ws.pending) return MMG_OK;
  launch_grm_combine(ctx, ws.C32, ws.p_D, ws.p_Npad, ws.p_N, ws.p_step, (double)(1 << ws.p_bd), ws.dc1acc, ws.p_c0, dC, 1);
  ws.pending = false;
  ws.p_M = 0;
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

static int kinship_grm_i8_into(mmg_ctx* ctx, mmg_geno* g, double* dC, GrmWorkspace& ws) {
  const bool verbose = std::getenv("MMG_KIN_VERBOSE") != nullptr;
  auto now = [&]() { (void)hipStreamSynchronize(ctx->stream); return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tv0 = verbose ? now() : 0.0;
  double tv_alloc = 0.0, tv_pack = 0.0, tv_gemm = 0.0, tv_tail = 0.0;
  // NON-NEGATIVE digits of the (positive) weight: digit * s must fit int8 for every genotype value, negative codes
  // included (|s| <= smax covers both signs: the store tracks max|s|), so digits are 7 / 6 / 5 bits wide for
  // alphabets within +-1 / +-2 / +-4.  Round 2 used balanced digits one bit wider -- their most negative value times a
  // negative genotype wrapped (advisor r2), and two's-complement negatives cost ~6 % of the power-limited MFMA rate
  // (tools/probe/mfma_digit_range.hip).  >= 30 bits of the largest weight; binary stores of >= 2^16 SNPs take 4
  // planes (28 bits: the roundings are independent per SNP and average down as 1/sqrt(M)).
  // Round 4: a store of 0 / 1 / 2 (diploid counts, plink2hdf5.py's output) is read as s - 1 in {-1, 0, 1} -- K only sees
  // s - mean, so the shift changes nothing, and 7-bit digits fit again: the plane counts of a binary store (4 from 2^16
  // SNPs on) instead of five 6-bit planes.  The images are written shifted (grm_scale_rows_kernel<.., SHIFT>) together with
  // a plain shifted image that replaces the store as the GEMMs' second operand.  MMG_GRM_CENTRE=0: the 6-bit planes.
  // (MMG_KIN_KERNEL=w4 / w8: the individual-major generations of the GEMM, which read transposed images -- no centring there)
  static const bool tr_off = [] { const char* e = std::getenv("MMG_KIN_KERNEL"); return e && (std::string(e) == "w4" || std::string(e) == "w8"); }();
  static const bool centre_off = [] { const char* e = std::getenv("MMG_GRM_CENTRE"); return e && e[0] == '0'; }();
  const bool centre = g->smax == 2 && g->sneg == 0 && !tr_off && !centre_off;
  const int smax_eff = centre ? 1 : g->smax;
  int bd = 0;
  if (smax_eff <= 1) bd = 7; else if (smax_eff <= 2) bd = 6; else if (smax_eff <= 4) bd = 5;
  if (bd == 0) return MMG_E_STATE;
  int D = (30 + bd - 1) / bd;                              // 5, 5, 6 planes
  // ... the SNPs of the whole run count: a short call that can join a four-plane run of >= 2^16 SNPs takes four as well
  // (the tail group of a chromosome; five planes there meant one GEMM per plane and a workspace of another shape)
  const int D_alone = (bd == 7 && g->M >= 65536) ? 4 : D;
  if (bd == 7 && (g->M >= 65536 || (ws.pending && ws.p_D == 4 && ws.p_M + g->M >= 65536))) D = 4;
  const bool planes_env = std::getenv("MMG_GRM_PLANES") != nullptr;
  if (planes_env) { const int v = std::atoi(std::getenv("MMG_GRM_PLANES")); if (v >= 3 && v <= 6) D = v; }   // 3: experiments only
  const double base = (double)(1 << bd);
  // every int32 plane sums digit * s_i * s_j over ALL SNPs between two combine passes: the digits are non-negative, so a
  // plane only grows -- (2^bd - 1) smax^2 M must stay below 2^31 (advisor r3; 16.9 M binary SNPs, 8.5 M for alphabets
  // within +-2; the chunked drivers pass far fewer per call, and a run of calls is cut where the bound would be reached)
  auto planes_hold = [&](double smax, int64_t m) { return (double)((1 << bd) - 1) * smax * smax * (double)m < 2147483648.0; };
  if (!planes_hold(smax_eff, g->M))
    return set_err(ctx, MMG_E_ARG, "exact GRM: too many SNPs in one call for the 32-bit digit planes (split the call)");
  // Round 3: the digit images are SNP-major like the store (row m scaled by the digit of SNP m) and the GEMM reads both
  // through transposed LDS reads (kinship_i8_tr_kernel) -- no transposition pass, no plain image.
  const bool direct = !tr_off;
  // MMG_GRM_DEFER=0: combine at the end of every call with the step taken from the call's own largest weight (round 3)
  static const bool defer = [] { const char* e = std::getenv("MMG_GRM_DEFER"); return !(e && e[0] == '0'); }();
  // binary store and four planes: ONE pass over the genotypes computes all four (gemm_i8_grm4.h: the scaled operands are
  // formed in registers from the plain tiles) -- no digit images at all.  MMG_GRM_FUSED=0: one GEMM per plane.
  // Five planes of a binary store (a short call, or rare variants stretching the weight range -- calc_ibd_kinship has no MAF
  // filter): the one-pass kernel takes planes 1-4 and only the lowest plane goes through a digit image and a GEMM of its own
  // (mode 2; 60 -> 48 ms at the C3 shape, where five image GEMMs cost 10.6 ms each and the one pass 35).  MMG_GRM_HYBRID=0.
  auto fused_for = [&](int d) -> int {
    const char* e = std::getenv("MMG_GRM_FUSED");
    const char* h = std::getenv("MMG_GRM_HYBRID");
    if (!(direct && g->smax <= 1 && g->sneg == 0) || (e && e[0] == '0')) return 0;
    return d == 4 ? 1 : (d == 5 && !(h && h[0] == '0')) ? 2 : 0;
  };
  int fused = fused_for(D);
  const int64_t M = g->M, CH = kin_chunk();
  const int64_t Mk_max = std::min(round_up(M, BK), CH);
  // grow-only: a stream that alternates between call shapes (100,000-SNP groups and a chromosome's 50,000-SNP tail, four
  // and five planes) re-allocated 50 GB at every change -- 3-5 s each at N = 50,000
  int reallocs = 0;
  auto ensure_ws = [&](int D, int fused) -> int {
    const size_t need_img = fused == 1 ? 16 : (size_t)g->Npad * Mk_max, need_c32 = (size_t)g->Npad * g->Npad;
    const int need_nimg = fused == 2 ? 1 : D + (centre ? 1 : 0);   // + the plain shifted image of a 0 / 1 / 2 store
    if (ws.cap_img < need_img || ws.cap_c32 < need_c32 || ws.capD < D || ws.cap_nimg < need_nimg || ws.cap_m < (size_t)M ||
        ws.cap_mk < (size_t)Mk_max || ws.cap_n < (size_t)g->Npad || ws.direct != direct) {
      int rcf = grm_flush(ctx, ws, dC);                       // the planes are about to be freed
      if (rcf) return rcf;
      const size_t c_img = std::max(ws.cap_img, need_img), c_c32 = std::max(ws.cap_c32, need_c32),
                   c_m = std::max(ws.cap_m, (size_t)M), c_mk = std::max(ws.cap_mk, (size_t)Mk_max),
                   c_n = std::max(ws.cap_n, (size_t)g->Npad);
      const int c_D = std::max(ws.capD, D), c_nimg = std::max(ws.cap_nimg, need_nimg);
      ws.release();
      ++reallocs;
      ws.direct = direct;
      hipError_t e = direct ? hipMalloc(&ws.dpart, (size_t)grm_partial_doubles((int64_t)c_mk, (int32_t)c_n) * sizeof(double))
                            : hipMalloc(&ws.Xq, c_img);
      if (e == hipSuccess) e = hipMalloc(&ws.Xp, (size_t)c_nimg * c_img);
      if (e == hipSuccess) e = hipMalloc(&ws.C32, (size_t)c_D * c_c32 * sizeof(int));
      if (e == hipSuccess) e = hipMalloc(&ws.ddig, (size_t)c_D * c_mk);
      if (e == hipSuccess) e = hipMalloc(&ws.dcoef, c_mk * sizeof(double));
      if (e == hipSuccess) e = hipMalloc(&ws.dc1, c_n * sizeof(double));
      if (e == hipSuccess) e = hipMalloc(&ws.dc1acc, c_n * sizeof(double));
      if (e == hipSuccess) e = hipMalloc(&ws.dm, c_m * sizeof(double));
      if (e == hipSuccess) e = hipMalloc(&ws.ds, c_m * sizeof(double));
      if (e == hipSuccess) e = hipMalloc(&ws.dwst, (size_t)grm_weight_blocks((int64_t)c_m) * 4 * sizeof(double));
      if (e != hipSuccess) { ws.release(); return set_err(ctx, MMG_E_NOMEM, std::string("hipMalloc GRM workspace: ") + hipGetErrorString(e)); }
      ws.cap_img = c_img; ws.cap_c32 = c_c32; ws.capD = c_D; ws.cap_nimg = c_nimg; ws.cap_m = c_m; ws.cap_mk = c_mk; ws.cap_n = c_n;
    }
    return MMG_OK;
  };
  { int rcw = ensure_ws(D, fused); if (rcw) return rcw; }
  // per-SNP mean / std in fp64 on the device; of the weights only four numbers per 4096 SNPs come to the host
  auto snp_stats = [&]() {
    launch_snp_stats(ctx, g, ws.dm, ws.ds);
    if (centre) launch_add_scalar_f64(ctx, ws.dm, M, -1.0);    // the mean of s - 1
  };
  snp_stats();
  const int64_t nwb = grm_weight_blocks(M);
  launch_grm_weight_stats(ctx, ws.dm, ws.ds, M, ws.dwst);
  std::vector<double> wst((size_t)nwb * 4);
  MMG_HIP(ctx, hipMemcpyAsync(wst.data(), ws.dwst, wst.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double wmax = 0.0, wmin = 1e300, c0 = 0.0, bad = 0.0;
  for (int64_t b = 0; b < nwb; ++b) {
    wmax = std::max(wmax, wst[4 * b]); wmin = std::min(wmin, wst[4 * b + 1]); c0 += wst[4 * b + 2]; bad += wst[4 * b + 3];
  }
  if (bad > 0.0) return set_err(ctx, MMG_E_ARG, "monomorphic SNP (std == 0) in the GRM kinship");
  // The step of a run of calls is set by its first call with 1/16 of headroom (0.09 bit); a later call joins the run
  // while its largest weight fits under that cap and is within a factor two of it (at most one bit less than a step of its
  // own would give), the plane count it needs is the run's, and the planes stay inside int32.
  // Four planes quantise a weight to 2^-28 of the LARGEST one: fine while the weights are of one size (a MAF filter of
  // 0.1 keeps wmax / wmin below 2.8), not when rare variants stretch the range (no filter: wmax / wmin ~ N / 4) -- then
  // the fifth plane stays (advisor r3).  The fused one-pass kernel computes four planes; five take one GEMM per plane.
  auto plan_for = [&](double wcap, int& Dn, int& fn) {
    Dn = D; fn = fused;
    if (Dn == 4 && wcap > 64.0 * wmin && !planes_env) { Dn = 5; fn = fused_for(5); }
  };
  int Dn = D, fn = fused;
  if (ws.pending) {
    plan_for(ws.p_wcap, Dn, fn);
    bool joins = defer && Dn == ws.p_D && fn == ws.p_fused && bd == ws.p_bd && g->Npad == ws.p_Npad && g->N == ws.p_N &&
                 wmax <= ws.p_wcap && 2.0 * wmax > ws.p_wcap && planes_hold(std::max(ws.p_smax, (double)smax_eff), ws.p_M + M);
    if (joins) {
      // the run's plan (Dn, fn) can differ from the one the workspace was sized for above (a >= 2^16-SNP call that falls back
      // to five planes because of its weight range needs digit IMAGES, which the fused plan does not allocate): size it for
      // the plan that runs.  A re-allocation flushes the run, and the call then starts one of its own.
      const int before = reallocs;
      int rcw = ensure_ws(Dn, fn);
      if (rcw) return rcw;
      if (reallocs != before) { snp_stats(); joins = false; }
    } else {
      int rcf = grm_flush(ctx, ws, dC);
      if (rcf) return rcf;
    }
    if (!joins && !planes_env) { D = D_alone; fused = fused_for(D); }   // a run of its own: the call's own SNP count decides
  }
  if (!ws.pending) {
    const double wcap = defer ? wmax * 1.0625 : wmax;
    plan_for(wcap, Dn, fn);
    const int before = reallocs;
    { int rcw = ensure_ws(Dn, fn); if (rcw) return rcw; }
    if (reallocs != before) snp_stats();                              // a fifth plane re-allocated the workspace
    ws.p_D = Dn; ws.p_fused = fn; ws.p_bd = bd; ws.p_wcap = wcap; ws.p_Npad = g->Npad; ws.p_N = g->N;
    // D unsigned digits reach B^D - 1: the cap is scaled onto exactly that
    ws.p_step = wcap / (std::pow(base, Dn) - 1.0);
    ws.p_c0 = 0.0; ws.p_M = 0; ws.p_smax = 0.0;
    MMG_HIP(ctx, hipMemsetAsync(ws.C32, 0, (size_t)Dn * g->Npad * g->Npad * sizeof(int), ctx->stream));
    MMG_HIP(ctx, hipMemsetAsync(ws.dc1acc, 0, g->Npad * sizeof(double), ctx->stream));
  }
  D = Dn; fused = fn;
  const double step = ws.p_step;
  double *dm = ws.dm, *ds = ws.ds, *dcoef = ws.dcoef, *dc1 = ws.dc1;
  int8_t *ddig = ws.ddig, *Xq = ws.Xq, *Xp = ws.Xp;
  int* C32 = ws.C32;
  if (verbose) tv_alloc = now();
  int rc = MMG_OK;
  double kin_ms = 0.0;
  ws.pending = true;                                          // from here on the planes hold part of the sum
  ws.in_call = true;                                          // ... and a return before the end leaves a partial call in them
  for (int64_t mb = 0; mb < M && rc == MMG_OK; mb += CH) {
    const int64_t Mk = round_up(std::min(CH, M - mb), BK);
    launch_grm_digits(ctx, dm, ds, mb, M, Mk, step, bd, D, ddig, dcoef, ws.stream_M);
    const double tp0 = verbose ? now() : 0.0;
    const int8_t* Srow = g->d + mb * (int64_t)g->Npad;     // the chunk's rows in the store (rows M..Mpad are zero)
    {
      EvScope ev(ctx, EV_PACK);
      if (direct)   // digit images + c1[i] = sum_m (a b)_m s_mi of the chunk in one pass over the store
        launch_grm_scale_rows(ctx, Srow, std::min<int64_t>(Mk, g->Mpad - mb), Mk, g->Npad, g->sneg > 0, Xp, ddig,
                              fused == 1 ? 0 : fused == 2 ? 1 : D, dcoef, ws.dpart, dc1, centre ? g->N : 0);
      else
        launch_transpose_digits(ctx, g, Xq, Xp, Mk, mb, ddig, D);
    }
    MMG_HIP(ctx, hipGetLastError());
    const double tp1 = verbose ? now() : 0.0;
    tv_pack += tp1 - tp0;
    if (fused) {                                             // planes 0-3 of four, or 1-4 of five
      const size_t up = fused == 2 ? 1 : 0;
      rc = run_kinship_grm4(ctx, Srow, g->Npad, g->Npad, Mk / BK, ddig + up * Mk, Mk, C32 + up * g->Npad * g->Npad);
      double a = 0.0;
      if (rc == MMG_OK && mmg_last_kernel_ms(ctx, EV_KIN, &a) == MMG_OK) kin_ms += a;
    }
    const int n_img_gemms = fused == 1 ? 0 : fused == 2 ? 1 : D;
    for (int d = 0; d < n_img_gemms && rc == MMG_OK; ++d) {
      if (direct)
        rc = run_kinship_i8_tr(ctx, Xp + (size_t)d * g->Npad * Mk, centre ? Xp + (size_t)D * g->Npad * Mk : Srow, g->Npad, g->Npad,
                               Mk / BK, C32 + (size_t)d * g->Npad * g->Npad);
      else
        rc = run_kinship_i8_pq(ctx, Xp + (size_t)d * g->Npad * Mk, Xq, g->Npad, Mk, C32 + (size_t)d * g->Npad * g->Npad);
      double a = 0.0;
      if (rc == MMG_OK && mmg_last_kernel_ms(ctx, EV_KIN, &a) == MMG_OK) kin_ms += a;
    }
    if (rc) break;
    if (verbose) tv_gemm += now() - tp1;
    // c1[i] += sum_m (a b)_m s_mi: one dot product per row of the plain image (the direct path has it already)
    if (!direct) launch_snp_dot_raw(ctx, Xq, Mk, g->Npad, (int32_t)Mk, dcoef, dc1);
    launch_add_into_f64(ctx, ws.dc1acc, dc1, g->Npad);         // chunk after chunk, in order: deterministic
  }
  if (rc) return rc;                                          // in_call stays set: the caller marks the accumulator broken
  ctx->grm_ms_total = kin_ms;
  ws.p_c0 += c0;
  ws.p_M += M;
  ws.p_smax = std::max(ws.p_smax, (double)smax_eff);
  const double tt0 = verbose ? now() : 0.0;
  if (!defer) { int rcf = grm_flush(ctx, ws, dC); if (rcf) return rcf; }
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ws.in_call = false;
  ws.stream_M += M;
  if (verbose) {
    tv_tail = now() - tt0;
    fprintf(stderr, "[grm] N=%d M=%lld D=%d: stats+alloc+memset %.3f s, pack %.3f s, gemms(+launch) %.3f s (kernels %.3f s), "
                    "combine %.3f s%s, total %.3f s\n", g->N, (long long)M, D, tv_alloc - tv0, tv_pack, tv_gemm, kin_ms * 1e-3,
            tv_tail, defer ? " (deferred)" : "", now() - tv0);
  }
  return MMG_OK;
}

int mmg_kin_acc_add_grm(mmg_ctx* ctx, mmg_kin_acc* a, mmg_geno* g) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a && g && g->N == a->N);
  MMG_ACC_USABLE(ctx, a);
  if (g->M == 0) return MMG_OK;
  int rc = kinship_grm_i8_into(ctx, g, a->dC, a->ws);
  // A failure behind the point where the call began to add (a GEMM launch, a stream error) leaves the int32 planes with part
  // of this call on top of the earlier calls of the run: neither dropping the run (the earlier calls' SNPs stay counted in
  // n_snps: advisor r4) nor keeping it gives a sum that matches a count.  The accumulator says so from here on.
  if (rc != MMG_OK && a->ws.in_call) { a->broken = true; return rc; }
  if (rc == MMG_E_STATE) {                                  // genotype alphabet too wide for int8 digit products
    Scratch sc;
    double *dm = nullptr, *ds = nullptr;
    MMG_HIP(ctx, sc.alloc(&dm, g->M * sizeof(double)));
    MMG_HIP(ctx, sc.alloc(&ds, g->M * sizeof(double)));
    launch_snp_stats(ctx, g, dm, ds);
    std::vector<double> mean((size_t)g->M), sd((size_t)g->M);
    MMG_HIP(ctx, hipMemcpyAsync(mean.data(), dm, g->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipMemcpyAsync(sd.data(), ds, g->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<float> sc_((size_t)g->M), sh_((size_t)g->M);
    for (int64_t m = 0; m < g->M; ++m) {
      if (!(sd[m] > 0.0)) return set_err(ctx, MMG_E_ARG, "monomorphic SNP (std == 0) in the GRM kinship");
      sc_[m] = (float)(1.0 / sd[m]); sh_[m] = (float)(-mean[m] / sd[m]);
    }
    rc = kinship_affine_into(ctx, g, sc_.data(), sh_.data(), a->dC, true);
  }
  if (rc == MMG_OK) a->n_snps += g->M;
  return rc;
}

extern "C" int mmg_reml_create_dev(mmg_ctx* ctx, int32_t N, int32_t q, const double* dK, const double* X, const double* y, mmg_reml** out);

int mmg_reml_create_from_acc(mmg_ctx* ctx, mmg_kin_acc* a, int32_t q, const double* X, const double* y, mmg_reml** out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a && X && y && out);
  MMG_ACC_USABLE(ctx, a);
  { int rcf = grm_flush(ctx, a->ws, a->dC); if (rcf) return rcf; }
  return mmg_reml_create_dev(ctx, a->N, q, a->dC, X, y, out);
}

// The accumulator takes the IBS kinship of a store (kinship.py:14-56: counts / (2 m_total) + 0.5, scale_k's rule when scaled) as
// its matrix, formed and kept in HBM: with mmg_reml_create_from_acc behind it the kinship of an emmax() call never visits the
// host (N = 5000: two 200 MB crossings of PCIe and the host's scale_k, ~40 ms of a 170 ms kinship -> threshold job).  comm /
// m_total: the SNP blocks of all ranks.  Replaces whatever the accumulator held.
int mmg_kin_acc_set_ibs(mmg_ctx* ctx, mmg_comm* comm, mmg_kin_acc* a, mmg_geno* g, int64_t m_total, int32_t scaled) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a && g && g->N == a->N && m_total >= g->M);
  MMG_ACC_USABLE(ctx, a);
  // pending GRM sums are dropped with the matrix they belonged to, and the run state of the exact route starts over with it
  a->ws.pending = false; a->ws.p_M = 0; a->ws.p_c0 = 0.0; a->ws.p_smax = 0.0; a->ws.stream_M = 0;
  IbsF64 f{nullptr, m_total, scaled != 0};
  f.dK_dev = a->dC;
  int rc = kinship_counts_i8(ctx, g, 2, -1, 0, nullptr, comm, &f);
  if (rc == MMG_OK) { a->n_snps = m_total; return rc; }
  // the counts are converted INTO dC and scaled there: a failure may have left it half built ('not positive' from scale_k's
  // rule, a HIP error) with n_snps still describing what it replaced -- the accumulator says so from here on (advisor r5)
  a->n_snps = 0;
  a->broken = true;
  return rc;
}

int mmg_kin_acc_snps(mmg_ctx* ctx, mmg_kin_acc* a, int64_t* n_snps) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a && n_snps);
  *n_snps = a->n_snps;
  return MMG_OK;
}

int mmg_kin_acc_pending(mmg_ctx* ctx, mmg_kin_acc* a, int64_t* n_snps_pending) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a && n_snps_pending);
  *n_snps_pending = a->ws.pending ? a->ws.p_M : 0;
  return MMG_OK;
}

int mmg_kin_acc_fetch(mmg_ctx* ctx, mmg_kin_acc* a, double* C_out, int64_t* n_snps) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a && C_out);
  MMG_ACC_USABLE(ctx, a);
  { int rcf = grm_flush(ctx, a->ws, a->dC); if (rcf) return rcf; }
  MMG_HIP(ctx, hipMemcpyAsync(C_out, a->dC, (size_t)a->N * a->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_snps) *n_snps = a->n_snps;
  return MMG_OK;
}

int mmg_kin_acc_scale_k(mmg_ctx* ctx, mmg_kin_acc* a, double* scalar_out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a != nullptr);
  MMG_ACC_USABLE(ctx, a);
  { int rcf = grm_flush(ctx, a->ws, a->dC); if (rcf) return rcf; }
  Scratch sc;
  const int64_t N = a->N;
  double* drow = nullptr;
  MMG_HIP(ctx, sc.alloc(&drow, 2 * N * sizeof(double)));
  launch_row_sums_f64(ctx, a->dC, N, drow, drow + N);          // row sums and the diagonal, fixed summation order
  std::vector<double> h((size_t)2 * N);
  MMG_HIP(ctx, hipMemcpyAsync(h.data(), drow, 2 * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double total = 0.0, trace = 0.0;
  for (int64_t i = 0; i < N; ++i) { total += h[i]; trace += h[N + i]; }
  const double c = trace - total / (double)N;
  if (!(c > 0.0) || !std::isfinite(c)) return set_err(ctx, MMG_E_ARG, "scale_k: tr K - sum K / N is not positive");
  const double scalar = (double)(N - 1) / c;
  launch_scale_f64(ctx, a->dC, N * N, scalar);
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (scalar_out) *scalar_out = scalar;
  return MMG_OK;
}

int mmg_kin_acc_allreduce(mmg_ctx* ctx, mmg_comm* comm, mmg_kin_acc* a) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, a != nullptr);
  MMG_ACC_USABLE(ctx, a);
  { int rcf = grm_flush(ctx, a->ws, a->dC); if (rcf) return rcf; }
  if (!comm || comm->world <= 1) return MMG_OK;
  Scratch sc;
  long long* dn = nullptr;
  MMG_HIP(ctx, sc.alloc(&dn, sizeof(long long)));
  long long n = (long long)a->n_snps;
  MMG_HIP(ctx, hipMemcpyAsync(dn, &n, sizeof(n), hipMemcpyHostToDevice, ctx->stream));
  MMG_NCCL(ctx, ncclGroupStart());
  MMG_NCCL(ctx, ncclAllReduce(a->dC, a->dC, (size_t)a->N * a->N, ncclDouble, ncclSum, comm->comm, ctx->stream));
  MMG_NCCL(ctx, ncclAllReduce(dn, dn, 1, ncclInt64, ncclSum, comm->comm, ctx->stream));
  MMG_NCCL(ctx, ncclGroupEnd());
  MMG_HIP(ctx, hipMemcpyAsync(&n, dn, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  a->n_snps = (int64_t)n;
  return MMG_OK;
}

int mmg_kin_acc_destroy(mmg_ctx* ctx, mmg_kin_acc* a) {
  if (!a) return MMG_OK;
  MMG_NOTE_ENTRY();
  if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
  hipFree(a->dC);
  a->ws.release();
  delete a;
  return MMG_OK;
}

int mmg_kinship_i8(mmg_ctx* ctx, const int8_t* snps, int64_t M, int32_t N, const float* scale, const float* shift,
                   double* C_out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, snps && C_out && M > 0 && N > 0);
  mmg_geno* g = nullptr;
  int rc = mmg_geno_create(ctx, M, N, &g);
  if (rc) return rc;
  rc = mmg_geno_upload(ctx, g, snps, 0, M);
  if (rc == MMG_OK) {
    if (!scale) {
      std::vector<int64_t> c64((size_t)N * N);
      rc = mmg_kinship_ibs_i8(ctx, g, c64.data());
      if (rc == MMG_OK) for (size_t i = 0; i < c64.size(); ++i) C_out[i] = (double)c64[i];
    } else {
      rc = mmg_kinship_affine_f32(ctx, g, scale, shift, C_out);
    }
  }
  mmg_geno_destroy(ctx, g);
  return rc;
}

// ------------------------------------------------------------------------- eigh / dgemm
}  // extern "C" (reopened below)
namespace mmg {
int eigh_block_jacobi(mmg_ctx* ctx, rocblas_handle h, double* dA, int32_t N, int32_t block, double* evals, double* evecs,
                      std::string& err);
}
extern "C" {
static int get_rocblas(mmg_ctx* ctx, rocblas_handle* h) {
  if (!ctx->rocblas) {
    rocblas_handle hh;
    MMG_RB(ctx, rocblas_create_handle(&hh));
    MMG_RB(ctx, rocblas_set_stream(hh, ctx->stream));
    ctx->rocblas = hh;
  }
  *h = (rocblas_handle)ctx->rocblas;
  return MMG_OK;
}

int mmg_eigh_f64(mmg_ctx* ctx, const double* A, int32_t N, double* evals, double* evecs) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, A && evals && N > 0);
  // rocSOLVER 7.2 has no 64-bit-index syevd: element offsets lda*N wrap at N*N >= 2^31 and the solver faults
  // on the device (seen at N = 50000).  Beyond that range -- or when MMG_EIGH_BLOCK=<rows> asks for it (tests) --
  // the block-Jacobi solver of eigh_block.hip runs dsyevd on block pairs that stay inside the range.
  {
    int block = 0;
    if (const char* e = std::getenv("MMG_EIGH_BLOCK")) block = std::atoi(e);
    if ((int64_t)N * N >= (int64_t)1 << 31 && (block <= 0 || block > 23168)) block = 23168;
    if (block > 0 && N > block) {
      rocblas_handle hb;
      int rcb = get_rocblas(ctx, &hb);
      if (rcb) return rcb;
      double* dAb = nullptr;
      MMG_HIP(ctx, sc.alloc(&dAb, (size_t)N * N * sizeof(double)));
      MMG_HIP(ctx, hipMemcpyAsync(dAb, A, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      std::string err;
      int rce;
      {
        EvScope ev(ctx, EV_EIGH);
        rce = eigh_block_jacobi(ctx, hb, dAb, N, block, evals, evecs, err);
      }
      if (rce) return set_err(ctx, rce, err);
      return MMG_OK;
    }
  }
  rocblas_handle h;
  int rc = get_rocblas(ctx, &h);
  if (rc) return rc;
  double *dA = nullptr, *dD = nullptr, *dE = nullptr;
  rocblas_int* dinfo = nullptr;
  MMG_HIP(ctx, sc.alloc(&dA, (size_t)N * N * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dD, N * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dE, N * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dinfo, sizeof(rocblas_int)));
  MMG_HIP(ctx, hipMemcpyAsync(dA, A, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  rocblas_status st;
  {
    EvScope ev(ctx, EV_EIGH);
    // row-major symmetric input == column-major symmetric input; use the lower triangle of the
    // column-major view (= upper triangle of the row-major matrix).
    st = rocsolver_dsyevd(h, evecs ? rocblas_evect_original : rocblas_evect_none, rocblas_fill_lower, N, dA, N, dD,
                          dE, dinfo);
  }
  rocblas_int info = 0;
  hipError_t e = hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(evals, dD, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  // column-major eigenvector matrix V read as row-major is V^T: ROWS are the eigenvectors,
  // which is the layout the reference keeps (linear_models.py:596).
  if (e == hipSuccess && evecs)
    e = hipMemcpyAsync(evecs, dA, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (st != rocblas_status_success) return set_err(ctx, MMG_E_LIB, "rocsolver_dsyevd failed: status " + std::to_string((int)st));
  if (e != hipSuccess) return set_err(ctx, MMG_E_HIP, hipGetErrorString(e));
  if (info != 0) return set_err(ctx, MMG_E_LIB, "rocsolver_dsyevd did not converge: info " + std::to_string((int)info));
  return MMG_OK;
}

int mmg_dgemm_f64(mmg_ctx* ctx, int ta, int tb, int32_t M, int32_t N, int32_t K, const double* A, const double* B,
                  double* C) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, A && B && C && M > 0 && N > 0 && K > 0);
  rocblas_handle h;
  int rc = get_rocblas(ctx, &h);
  if (rc) return rc;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  MMG_HIP(ctx, sc.alloc(&dA, (size_t)M * K * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dB, (size_t)K * N * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dC, (size_t)M * N * sizeof(double)));
  MMG_HIP(ctx, hipMemcpyAsync(dA, A, (size_t)M * K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  MMG_HIP(ctx, hipMemcpyAsync(dB, B, (size_t)K * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const double one = 1.0, zero = 0.0;
  // row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T
  rocblas_status st = rocblas_dgemm_64(h, tb ? rocblas_operation_transpose : rocblas_operation_none,
                                    ta ? rocblas_operation_transpose : rocblas_operation_none, N, M, K, &one, dB,
                                    tb ? K : N, dA, ta ? M : K, &zero, dC, N);
  hipError_t e = hipMemcpyAsync(C, dC, (size_t)M * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (st != rocblas_status_success) return set_err(ctx, MMG_E_LIB, "rocblas_dgemm failed");
  if (e != hipSuccess) return set_err(ctx, MMG_E_HIP, hipGetErrorString(e));
  return MMG_OK;
}

// ------------------------------------------------------------------------- scan
static double ln_beta_half(double a) {  // ln B(a, 1/2)
  const double half_ln_pi = 0.57236494292470008707;
  if (a < 30.0) return std::lgamma(a) + std::lgamma(0.5) - std::lgamma(a + 0.5);
  // ln Gamma(a+1/2) - ln Gamma(a) = 1/2 ln a - 1/(8a) + 1/(192 a^3) - 1/(640 a^5) + 17/(14336 a^7)
  const double i = 1.0 / a, i2 = i * i;
  const double diff = 0.5 * std::log(a) + i * (-1.0 / 8 + i2 * (1.0 / 192 + i2 * (-1.0 / 640 + i2 * (17.0 / 14336))));
  return half_ln_pi - diff;
}

// LPT assignment of the (digit, J) jobs with digit in [d0, d1) to the G job groups of an XCD cohort
static int build_schedule_range(mmg_ctx* ctx, const mmg_scan_model& md, int d0, int d1, int** job_off, int2** jobs_out,
                                int* njobs, int G = 0) {
  const int nJ = md.Npad / TM;
  if (G == 0) G = md.G;
  std::vector<std::pair<int, int>> all;  // (weight = J + 1 k-blocks, id)
  for (int d = d0; d < d1; ++d)
    for (int J = 0; J < nJ; ++J) all.push_back({J + 1, d * nJ + J});
  std::sort(all.begin(), all.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) {
    return a.first != b.first ? a.first > b.first : a.second < b.second;
  });
  std::vector<std::vector<int>> bins(G);
  std::vector<long> load(G, 0);
  for (auto& it : all) {
    int best = 0;
    for (int gI = 1; gI < G; ++gI) if (load[gI] < load[best]) best = gI;
    bins[best].push_back(it.second);
    load[best] += it.first;
  }
  std::vector<int> off(G + 1, 0);
  std::vector<int2> jobs;
  for (int gI = 0; gI < G; ++gI) {
    for (int id : bins[gI]) jobs.push_back(make_int2(id / nJ, id % nJ));
    off[gI + 1] = (int)jobs.size();
  }
  *njobs = (int)jobs.size();
  MMG_HIP(ctx, hipMalloc(job_off, (G + 1) * sizeof(int)));
  MMG_HIP(ctx, hipMalloc(jobs_out, std::max<size_t>(1, jobs.size()) * sizeof(int2)));
  MMG_HIP(ctx, hipMemcpy(*job_off, off.data(), (G + 1) * sizeof(int), hipMemcpyHostToDevice));
  if (!jobs.empty()) MMG_HIP(ctx, hipMemcpy(*jobs_out, jobs.data(), jobs.size() * sizeof(int2), hipMemcpyHostToDevice));
  return MMG_OK;
}

static int build_schedule(mmg_ctx* ctx, mmg_scan_model& md) {
  // 4 SNP blocks x 8 job groups per XCD cohort: the L2-miss traffic of the lock-stepped cohort is
  // proportional to 1/AS + 1/G (measured: AS 1/2/4/8 -> 90/49/32/31 GB at M=400k, N=5000)
  int AS = 4;
  if (const char* s = std::getenv("MMG_SCAN_AS")) AS = std::atoi(s);
  if (AS != 1 && AS != 2 && AS != 4 && AS != 8 && AS != 16 && AS != 32) AS = 4;
  md.AS = AS; md.G = 32 / AS;
  int rc = build_schedule_range(ctx, md, 0, md.D, &md.job_off, &md.jobs, &md.njobs);
  // tail cohorts: the same jobs over 2x / 4x as many groups
  auto tails = [&](int range, int d0, int d1) {
    int n = 0, rct = MMG_OK;
    for (int k = 0; k < 2 && rct == MMG_OK; ++k)
      if ((AS >> (k + 1)) >= 1) rct = build_schedule_range(ctx, md, d0, d1, &md.tail_off[range][k], &md.tail_jobs[range][k], &n, 32 / (AS >> (k + 1)));
    return rct;
  };
  if (rc == MMG_OK) rc = tails(0, 0, md.D);
  if (rc || !md.adaptive) return rc;
  if ((rc = build_schedule_range(ctx, md, 1, md.D, &md.job_off_hi, &md.jobs_hi, &md.njobs_hi))) return rc;
  if ((rc = build_schedule_range(ctx, md, 0, 1, &md.job_off_lo, &md.jobs_lo, &md.njobs_lo))) return rc;
  if ((rc = tails(1, 1, md.D))) return rc;
  return tails(2, 0, 1);
}

// Build a scan model from a DEVICE-resident fp64 matrix dA [N x N] and device vector dw [N].
// Rows Npad-16 .. Npad-1 of the top digit plane (padding individuals: all-zero rows of the matrix, and the genotype
// columns they meet in the GEMM's epilogue are zero too) <- seven balanced base-256 digits of w and of diag(A) and a
// row of ones, laid out for k_scan_w4s.hip (LIN).  52 bits relative to the largest entry, i.e. the vectors as they
// are in double precision.  Needs 16 free rows; otherwise the scan keeps the separate finalize pass over the store.
static int add_linear_rows(mmg_ctx* ctx, mmg_scan_model& md) {
  md.lin_rows = false;
  (void)hipFree(md.lin_tab); md.lin_tab = nullptr;
  if (md.Npad - md.N < 16 || md.D < 2) return MMG_OK;
  const int N = md.N, Npad = md.Npad;
  std::vector<double> w((size_t)N), dg((size_t)N);
  MMG_HIP(ctx, hipMemcpyAsync(w.data(), md.w, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipMemcpyAsync(dg.data(), md.diag, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double mw = 0.0, mdg = 0.0;
  for (int i = 0; i < N; ++i) { mw = std::max(mw, std::fabs(w[i])); mdg = std::max(mdg, std::fabs(dg[i])); }
  if (!std::isfinite(mw) || !std::isfinite(mdg)) return MMG_OK;
  md.lin_step_w = mw > 0.0 ? mw / std::ldexp(1.0, 52) : 1.0;
  md.lin_step_d = mdg > 0.0 ? mdg / std::ldexp(1.0, 52) : 1.0;
  std::vector<int8_t> rows((size_t)16 * Npad, 0);
  static const int row_w[7] = {0, 1, 2, 3, 8, 9, 10}, row_d[7] = {4, 5, 6, 7, 12, 13, 14};   // row - 240
  for (int k = 0; k < N; ++k) {
    long long Zw = std::llrint(w[k] / md.lin_step_w), Zd = std::llrint(dg[k] / md.lin_step_d);
    for (int d = 0; d < 7; ++d) {
      const long long zw = ((Zw + 128) & 255) - 128, zd = ((Zd + 128) & 255) - 128;
      Zw = (Zw - zw) >> 8; Zd = (Zd - zd) >> 8;
      rows[(size_t)row_w[d] * Npad + k] = (int8_t)zw;
      rows[(size_t)row_d[d] * Npad + k] = (int8_t)zd;
    }
    rows[(size_t)11 * Npad + k] = 1;
  }
  int8_t* dst = md.Bq + (size_t)(md.D - 1) * Npad * Npad + (size_t)(Npad - 16) * Npad;
  MMG_HIP(ctx, hipMemcpyAsync(dst, rows.data(), rows.size(), hipMemcpyHostToDevice, ctx->stream));
  // stores of 0/1/2 codes (k_scan.hip:lin_hi_bits_kernel): the digit rows of diag(A) and the ones once more, as an [8 x Npad]
  // table of their own.  Not fatal if it cannot be had: such stores then keep the finalize pass over their bytes.
  std::vector<int8_t> tab((size_t)8 * Npad, 0);
  for (int d = 0; d < 7; ++d) std::memcpy(&tab[(size_t)d * Npad], &rows[(size_t)row_d[d] * Npad], (size_t)Npad);
  std::memcpy(&tab[(size_t)7 * Npad], &rows[(size_t)11 * Npad], (size_t)Npad);
  if (hipMalloc(&md.lin_tab, tab.size()) == hipSuccess)
    MMG_HIP(ctx, hipMemcpyAsync(md.lin_tab, tab.data(), tab.size(), hipMemcpyHostToDevice, ctx->stream));
  else { md.lin_tab = nullptr; (void)hipGetLastError(); }
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  md.lin_rows = true;
  return MMG_OK;
}

static int model_from_device(mmg_ctx* ctx, mmg_scan_model& md, int32_t N, const double* dA, const double* dw,
                             int ndigits, bool adaptive = false) {
  Scratch sc;
  free_model(md);
  md.N = N; md.Npad = (int32_t)round_up(N, 256); md.D = ndigits;
  md.adaptive = adaptive && ndigits == 4;
  unsigned long long* dmax = nullptr;
  MMG_HIP(ctx, sc.alloc(&dmax, sizeof(unsigned long long)));
  MMG_HIP(ctx, hipMalloc(&md.Bq, (size_t)md.D * md.Npad * md.Npad));
  MMG_HIP(ctx, hipMalloc(&md.diag, md.Npad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&md.w, md.Npad * sizeof(double)));
  MMG_HIP(ctx, hipMemsetAsync(md.w, 0, md.Npad * sizeof(double), ctx->stream));
  MMG_HIP(ctx, hipMemsetAsync(dmax, 0, sizeof(unsigned long long), ctx->stream));
  MMG_HIP(ctx, hipMemcpyAsync(md.w, dw, N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  launch_absmax_offdiag(ctx, dA, N, dmax);
  unsigned long long bits = 0;
  MMG_HIP(ctx, hipMemcpyAsync(&bits, dmax, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double maxoff;
  std::memcpy(&maxoff, &bits, sizeof(double));
  if (!(maxoff > 0.0) || !std::isfinite(maxoff)) maxoff = 1.0;   // diagonal matrix: all digits are zero
  // |rint(2 A_jk / step)| <= 2^(7D-1) - 2: shifted by offset = 2^(7D-1) every stored entry is a non-negative number
  // below 2^(7D), i.e. D unsigned 7-bit digits (gemm_i8_core.h: SCAN_DIGIT_BITS)
  md.offset = std::ldexp(1.0, SCAN_DIGIT_BITS * md.D - 1);
  md.step = 2.0 * maxoff / (md.offset - 2.0);
  const int nT = md.Npad / 256;
  long long *dz0 = nullptr, *dzt = nullptr;
  MMG_HIP(ctx, sc.alloc(&dz0, sizeof(long long)));
  MMG_HIP(ctx, sc.alloc(&dzt, (size_t)nT * nT * sizeof(long long)));
  MMG_HIP(ctx, hipMemsetAsync(dz0, 0, sizeof(long long), ctx->stream));
  MMG_HIP(ctx, hipMemsetAsync(dzt, 0, (size_t)nT * nT * sizeof(long long), ctx->stream));
  launch_quantize(ctx, dA, N, md.Npad, md.D, 1.0 / md.step, (long long)md.offset, md.Bq, md.diag, dz0,
                  md.adaptive ? dzt : nullptr);
  MMG_HIP(ctx, hipGetLastError());
  long long z0 = 0;
  std::vector<long long> zt((size_t)nT * nT, 0);
  MMG_HIP(ctx, hipMemcpyAsync(&z0, dz0, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  if (md.adaptive)
    MMG_HIP(ctx, hipMemcpyAsync(zt.data(), dzt, zt.size() * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  md.mu0 = N > 1 ? (double)z0 / (0.5 * (double)N * (double)(N - 1)) : 0.0;
  if (md.adaptive) {
    // The adaptive schedule treats the lowest digit plane as noise with mean mu0 around which nothing is
    // structured.  Guard: the mean of every 256 x 256 tile of that plane must be within 8 sigma of mu0 (a uniform
    // 7-bit digit has sigma 36.95); a matrix with blocks that are constant to 21 bits would fail -- then every plane is
    // run for every SNP.
    for (int tj = 0; tj < nT && md.adaptive; ++tj)
      for (int tk = 0; tk <= tj; ++tk) {
        const int64_t rows = std::min<int64_t>(256, N - 256 * (int64_t)tj), cols = std::min<int64_t>(256, N - 256 * (int64_t)tk);
        if (rows <= 0 || cols <= 0) continue;
        const double cnt = tj == tk ? 0.5 * rows * (rows - 1) : (double)rows * cols;
        if (cnt < 1024) continue;
        const double dev = std::fabs((double)zt[(size_t)tj * nT + tk] - md.mu0 * cnt) / (36.95 * std::sqrt(cnt));
        if (dev > 8.0) { md.adaptive = false; break; }
      }
  }
  // the fp64 matrix stays with the model for the exact tier of the scan (exact_tier below): N^2 doubles next to D N^2 / 2 bytes
  // of digits.  MMG_SCAN_EXACT=0, or a failed allocation: the tier is off and the scan is what its planes give.
  // Only the default model has the tier: an explicit digit count asks for what that many planes give.
  const bool exact_on = [] { const char* e = std::getenv("MMG_SCAN_EXACT"); return !(e && e[0] == '0'); }();   // read per model
  md.coherent = adaptive && ndigits == 4 && !md.adaptive;
  if (exact_on && adaptive && ndigits == 4 && hipMalloc(&md.A64, (size_t)N * N * sizeof(double)) == hipSuccess)
    MMG_HIP(ctx, hipMemcpyAsync(md.A64, dA, (size_t)N * N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  else { (void)hipGetLastError(); md.A64 = nullptr; }
  int rcl = add_linear_rows(ctx, md);
  if (rcl) return rcl;
  rcl = build_schedule(ctx, md);
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));           // dA is the caller's scratch
  return rcl;
}

}  // extern "C"
namespace mmg {
int model_from_device_public(mmg_ctx* ctx, int32_t N, const double* dA, const double* dw, int ndigits, bool adaptive) {
  return model_from_device(ctx, ctx->model, N, dA, dw, ndigits, adaptive);
}
}  // namespace mmg
extern "C" {

int mmg_scan_set_model(mmg_ctx* ctx, int32_t N, const double* A, const double* w, int ndigits) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, A && w && N > 0);
  // ndigits = 0: the default model -- 4 digit planes with the adaptive schedule of mmg_emmax_scan_device (three
  // planes for every SNP, the fourth for those whose F passes a threshold); an explicit count runs all its planes
  // for every SNP.  MMG_SCAN_ADAPTIVE=0 turns the adaptive schedule off.
  bool adaptive = false;
  if (ndigits == 0) {
    ndigits = 4;
    const char* e = std::getenv("MMG_SCAN_ADAPTIVE");
    adaptive = !(e && e[0] == '0');
  }
  MMG_CHECK_ARG(ctx, ndigits >= 2 && ndigits <= 6);
  double *dA = nullptr, *dw = nullptr;
  MMG_HIP(ctx, sc.alloc(&dA, (size_t)N * N * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dw, N * sizeof(double)));
  MMG_HIP(ctx, hipMemcpyAsync(dA, A, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  MMG_HIP(ctx, hipMemcpyAsync(dw, w, N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int rc = model_from_device(ctx, ctx->model, N, dA, dw, ndigits, adaptive);
  return rc;
}

static int ensure_result(mmg_ctx* ctx, mmg_scan_result& r, int64_t Mpad) {
  if (r.cap >= Mpad) return MMG_OK;
  free_result(r);
  MMG_HIP(ctx, hipMalloc(&r.q, Mpad * sizeof(unsigned long long)));
  MMG_HIP(ctx, hipMalloc(&r.rss, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.F, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.p, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.dot, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.den, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.sum, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.dd, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.ssq, Mpad * sizeof(double)));
  MMG_HIP(ctx, hipMalloc(&r.linraw, Mpad * 16 * sizeof(int)));
  MMG_HIP(ctx, hipMalloc(&r.idx, Mpad * sizeof(int64_t)));
  MMG_HIP(ctx, hipMalloc(&r.scal, 4 * sizeof(unsigned long long)));
  r.cap = Mpad;
  return MMG_OK;
}

// The exact tier: SNPs whose den the digit planes cannot pin down to the target (k_scan.hip: scan_select_exact_kernel) get
// s'As from the fp64 matrix itself (scan_exact_den_kernel), in batches of gathered rows.  Budget: MMG_SCAN_EXACT_MAX_FLOP
// (default 4e13, ~1-2 s) -- a data set in which MOST SNPs need it is reported (n_exact = -1) rather than served.
static int exact_tier(mmg_ctx* ctx, mmg_geno* g, const mmg_scan_model& md, mmg_scan_result& res, double h0_rss, int32_t df2,
                      bool f_free, double target, bool full_planes) {
  res.n_exact = 0;
  if (!md.A64 || g->M == 0) return MMG_OK;
  const int N = md.N, nJB = (N + 255) / 256;
  const double sig_full = md.step / std::sqrt(12.0) / std::sqrt(2.0);   // every plane in: entries rounded to +-step / 2
  // what the SNPs that were not refined carry when the adaptive schedule ran: three planes (one digit coarser)
  const double coarse = full_planes ? 1.0 : (double)(1 << SCAN_DIGIT_BITS);
  auto ensure_rows = [&](int64_t rows) -> int {
    const int64_t bpad = round_up(rows, 256);
    if (!ctx->sel_geno || ctx->sel_geno->Mpad < bpad || ctx->sel_geno->N != g->N) {
      if (ctx->sel_geno) { mmg_geno_destroy(ctx, ctx->sel_geno); ctx->sel_geno = nullptr; }
      int rc = geno_create(ctx, bpad, g->N, &ctx->sel_geno, false);
      if (rc) return rc;
    }
    return MMG_OK;
  };
  Scratch sc;
  // ---- once per model: is the error model (independent roundings) true of THIS matrix?  64 SNPs spread over the store,
  // the planes' den against the fp64 one.  A kinship of a dozen genotype classes passes the tile test of the adaptive schedule
  // (the classes are interleaved) and is 100 x six sigma off: equal entries round alike.
  if (!md.exact_checked) {
    md.exact_checked = true;
    const int64_t ns = std::min<int64_t>(64, g->M);
    int rc = ensure_rows(ns);
    if (rc) return rc;
    double *Sd = nullptr, *part = nullptr;
    MMG_HIP(ctx, sc.alloc(&Sd, (size_t)ns * N * sizeof(double)));
    MMG_HIP(ctx, sc.alloc(&part, (size_t)ns * nJB * sizeof(double)));
    MMG_HIP(ctx, hipMemsetAsync(res.scal + 3, 0, sizeof(unsigned long long), ctx->stream));
    launch_scan_sample_idx(ctx, g->M, ns, res.idx);
    launch_gather_rows(ctx, g, res.idx, ns, ctx->sel_geno->d);
    launch_rows_to_f64(ctx, ctx->sel_geno->d, g->Npad, N, ns, Sd);
    launch_scan_exact_den(ctx, Sd, N, ns, md.A64, part);
    launch_scan_exact_check(ctx, res.idx, ns, part, N, res, sig_full * coarse, res.scal + 3);
    unsigned long long rb = 0;
    MMG_HIP(ctx, hipMemcpyAsync(&rb, res.scal + 3, sizeof(rb), hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double ratio;
    std::memcpy(&ratio, &rb, sizeof(double));
    if (ratio > 1.0) md.coherent = true;
  }
  MMG_HIP(ctx, hipMemsetAsync(res.scal + 3, 0, sizeof(unsigned long long), ctx->stream));
  // independent roundings: a SNP the adaptive schedule left at three planes passed this very test at the coarser sigma, so the
  // full-plane sigma is the one that can still flag anything (the refined ones); coherent: the bound of what each SNP carries
  launch_scan_select_exact(ctx, res, g->M, sig_full, 0.5 * md.step * coarse, md.coherent, target, res.scal + 3, !f_free);
  unsigned long long hc = 0;
  MMG_HIP(ctx, hipMemcpyAsync(&hc, res.scal + 3, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int64_t cnt = (int64_t)hc;
  if (cnt == 0) return MMG_OK;
  static const double max_flop = [] { const char* e = std::getenv("MMG_SCAN_EXACT_MAX_FLOP"); return e ? std::atof(e) : 4e13; }();
  if (2.0 * (double)cnt * md.N * md.N > max_flop) { res.n_exact = -1; return MMG_OK; }
  // <= 2 GB of fp64 rows and <= 65,535 groups of 8 SNPs (grid.y of scan_exact_den_kernel)
  const int64_t batch = std::max<int64_t>(8, std::min<int64_t>({cnt, ((int64_t)2 << 30) / ((int64_t)N * 8), (int64_t)65535 * 8}));
  { int rc = ensure_rows(batch); if (rc) return rc; }
  double *Sd = nullptr, *part = nullptr;
  MMG_HIP(ctx, sc.alloc(&Sd, (size_t)batch * N * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&part, (size_t)batch * nJB * sizeof(double)));
  for (int64_t b0 = 0; b0 < cnt; b0 += batch) {
    const int64_t nb = std::min(batch, cnt - b0);
    launch_gather_rows(ctx, g, res.idx + b0, nb, ctx->sel_geno->d);
    launch_rows_to_f64(ctx, ctx->sel_geno->d, g->Npad, N, nb, Sd);
    launch_scan_exact_den(ctx, Sd, N, nb, md.A64, part);
    launch_scan_exact_apply(ctx, res.idx + b0, nb, part, N, res, h0_rss, df2);
    MMG_HIP(ctx, hipGetLastError());
  }
  res.n_exact = cnt;
  return MMG_OK;
}

// Stores of 0/1/2 codes: make the [s = 2] bit image (mmg_internal.h:mmg_geno::hi2) available if this is at least the second
// scan of the store's present content (MMG_SCAN_HI2=1: already on the first, =0: never), and the [Mpad][8] result rows of
// lin_hi_bits_kernel.  Nothing here is fatal: without the image the scan keeps the finalize pass over the store's bytes.
static int hi2_prepare(mmg_ctx* ctx, mmg_geno* g, const mmg_scan_model& md, mmg_scan_result& res) {
  if (g->smax != 2 || g->sneg != 0 || !md.lin_rows || !md.lin_tab) return MMG_OK;
  const char* e = std::getenv("MMG_SCAN_HI2");
  if (e && e[0] == '0') { g->hi2_version = ~0ull; return MMG_OK; }
  if (g->scanned_version != g->version) { g->scanned_version = g->version; g->scans_of_version = 0; }
  const int seen = g->scans_of_version++;
  if (!res.linraw2) {
    if (hipMalloc(&res.linraw2, (size_t)res.cap * 8 * sizeof(int)) != hipSuccess) { res.linraw2 = nullptr; (void)hipGetLastError(); }
  }
  if (!res.linraw2) { g->hi2_version = ~0ull; return MMG_OK; }
  if (geno_hi2_ready(g)) return MMG_OK;
  if (seen < 1 && !(e && e[0] == '1')) return MMG_OK;
  if (!g->hi2 && hipMalloc(&g->hi2, (size_t)g->Mcap * (g->Npad >> 3)) != hipSuccess) { g->hi2 = nullptr; (void)hipGetLastError(); return MMG_OK; }
  launch_pack_hi_bits(ctx, g);
  MMG_HIP(ctx, hipGetLastError());
  g->hi2_version = g->version;
  return MMG_OK;
}

// The scan of g against model md into res: all planes for an explicit digit count, else the adaptive schedule.
// f_free: the refinement criterion ignores F (a quadratic form wanted for its own sake -- the permutation test's t.t --
// is refined wherever six sigma of the first pass exceed `target` of the form itself); with_p: p-values at the end.
static int scan_into(mmg_ctx* ctx, mmg_geno* g, const mmg_scan_model& md, mmg_scan_result& res, double h0_rss,
                     int32_t df2, bool f_free, bool with_p) {
  int rc = MMG_OK;
  const double lnb = ln_beta_half(0.5 * df2);
  res.n_refined = 0; res.eps_max = 0.0; res.sigma_ratio_max = 0.0; res.fell_back = 0; res.adaptive = md.adaptive ? 1 : 0;
  ctx->ev_set[EV_QUAD2] = false;
  MMG_HIP(ctx, hipMemsetAsync(res.q, 0, g->Mpad * sizeof(unsigned long long), ctx->stream));
  // by-products of the GEMM (s.w, sum A_ii s_i, sum s_i; binary stores, k_scan_w4s.hip LIN) replace the finalize
  // pass's second sweep over the genotype store
  if ((rc = hi2_prepare(ctx, g, md, res))) return rc;       // stores of 0/1/2 codes: the [s = 2] bit image, from their second scan on
  const LinOut lin_out{res.linraw};
  const bool lin = scan_lin_usable(g, md);
  const LinOut* linp = lin ? &lin_out : nullptr;
  const bool lin2 = lin && g->smax == 2;                    // s^2 = s + 2 [s = 2]: the correction terms from the image
  bool lin2_done = false;
  auto finalize = [&](bool with_p, double bias) {
    if (lin2 && !lin2_done) { launch_lin_hi_bits(ctx, g, md, res.linraw2); lin2_done = true; }
    if (lin) launch_scan_finalize_lin(ctx, g, md, res, h0_rss, df2, lnb, with_p, bias, lin2 ? res.linraw2 : nullptr);
    else launch_scan_finalize(ctx, g, md, res, h0_rss, df2, lnb, with_p, bias);
  };
  double target = 2.5e-7;                                  // a quarter of the 1e-6 bar on p
  if (const char* e = std::getenv("MMG_SCAN_ADAPT_TARGET")) target = std::atof(e);
  if (!md.adaptive) {
    rc = run_scan_quad(ctx, g, md, res.q, EV_QUAD, linp);
    if (rc) return rc;
    MMG_HIP(ctx, hipGetLastError());
    {
      EvScope ev(ctx, EV_FIN);
      finalize(false, 0.0);
    }
    MMG_HIP(ctx, hipGetLastError());
    rc = exact_tier(ctx, g, md, res, h0_rss, df2, f_free, target, true);
    if (rc) return rc;
    if (with_p && res.p && g->M > 0) launch_f_sf(ctx, res.F, g->M, df2, lnb, res.p);
    MMG_HIP(ctx, hipGetLastError());
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMG_OK;
  }
  // ---- adaptive precision.  Pass 1: the three upper digit planes for every SNP, i.e. the matrix truncated to 21
  // bits (unsigned digits: dropping the lowest one cuts off a value uniform in [0, 128); its mean, mu0 per stored
  // entry, goes back in as a bias term of the finalize kernels).  That leaves den off by a sum of
  // independent errors whose sigma is known per SNP (k_scan.hip:scan_select_kernel); a SNP whose p could move
  // by more than `target` at six sigma -- large F, or a den that is small against the rounding noise -- gets the
  // lowest plane added to the same exact integer in pass 2, which makes it bit-identical to a full 4-plane scan.
  // The refined SNPs double as a check of the error model: if any of them moved by more than its six-sigma
  // prediction, everything is redone with all planes.
  const double sig_unit = md.step * (double)(1 << SCAN_DIGIT_BITS) / std::sqrt(12.0) / std::sqrt(2.0);   // sigma = sig_unit * sum s^2
  mmg_scan_model hi = md, lo = md;                        // shallow copies with the schedule swapped
  hi.job_off = md.job_off_hi; hi.jobs = md.jobs_hi; hi.njobs = md.njobs_hi; hi.range = 1;
  lo.job_off = md.job_off_lo; lo.jobs = md.jobs_lo; lo.njobs = md.njobs_lo; lo.range = 2;
  rc = run_scan_quad(ctx, g, hi, res.q, EV_QUAD, linp);
  if (rc) return rc;
  MMG_HIP(ctx, hipGetLastError());
  {
    EvScope ev(ctx, EV_FIN);
    finalize(false, md.step * md.mu0);
  }
  MMG_HIP(ctx, hipMemsetAsync(res.scal, 0, 4 * sizeof(unsigned long long), ctx->stream));
  launch_scan_select(ctx, res, g->M, sig_unit, target, res.scal, !f_free);
  unsigned long long hs[3] = {0, 0, 0};
  MMG_HIP(ctx, hipMemcpyAsync(hs, res.scal, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int64_t cnt = (int64_t)hs[0];
  bool all = cnt > g->M / 2;                               // a selection this large is not worth compacting
  if (!all && cnt > 0) {
    const int64_t cpad = round_up(cnt, 256);
    if (!ctx->sel_geno || ctx->sel_geno->Mpad < cpad || ctx->sel_geno->N != g->N) {
      if (ctx->sel_geno) { mmg_geno_destroy(ctx, ctx->sel_geno); ctx->sel_geno = nullptr; }
      rc = geno_create(ctx, cpad + cpad / 4, g->N, &ctx->sel_geno, false);
      if (rc) return rc;
    }
    if (res.q2_cap < ctx->sel_geno->Mpad) {
      hipFree(res.q2); res.q2 = nullptr; res.q2_cap = 0;
      MMG_HIP(ctx, hipMalloc(&res.q2, ctx->sel_geno->Mpad * sizeof(unsigned long long)));
      res.q2_cap = ctx->sel_geno->Mpad;
    }
    mmg_geno view = *ctx->sel_geno;                        // a cpad-row window of the compact store
    view.M = cnt; view.Mpad = cpad; view.smax = g->smax; view.sneg = g->sneg; view.bits = nullptr; view.bits_valid = false; view.hi2 = nullptr;
    launch_gather_rows(ctx, g, res.idx, cnt, view.d);
    MMG_HIP(ctx, hipMemsetAsync(res.q2, 0, cpad * sizeof(unsigned long long), ctx->stream));
    rc = run_scan_quad(ctx, &view, lo, res.q2, EV_QUAD2);
    if (view.bits) (void)hipFree(view.bits);               // only the bit-packed diagnostic kernels build this twin
    if (rc) return rc;
    launch_scan_refine(ctx, res.idx, cnt, md, res, res.q2, sig_unit, h0_rss, df2, res.scal + 1);
    MMG_HIP(ctx, hipGetLastError());
    MMG_HIP(ctx, hipMemcpyAsync(hs, res.scal, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(&res.eps_max, &hs[1], sizeof(double));
    std::memcpy(&res.sigma_ratio_max, &hs[2], sizeof(double));
    res.n_refined = cnt;
    if (res.sigma_ratio_max > 1.0) all = true;            // a rounding error beyond six sigma: the error model failed
    if (all) {
      // undo nothing: the refined SNPs already carry plane 0; run it for the others by running it for all rows of
      // a store view whose refined rows are skipped -- simplest exact way: subtract is impossible on atomics, so
      // zero q and redo both passes over everything
      MMG_HIP(ctx, hipMemsetAsync(res.q, 0, g->Mpad * sizeof(unsigned long long), ctx->stream));
      rc = run_scan_quad(ctx, g, md, res.q, EV_QUAD, linp);
      if (rc) return rc;
    }
  } else if (all) {
    rc = run_scan_quad(ctx, g, lo, res.q, EV_QUAD2);       // plane 0 on top of planes 1-3: the full 4-plane integer
    if (rc) return rc;
  }
  if (all) {
    res.fell_back = 1;
    finalize(false, 0.0);
  }
  rc = exact_tier(ctx, g, md, res, h0_rss, df2, f_free, target, all);
  if (rc) return rc;
  if (with_p && res.p && g->M > 0) launch_f_sf(ctx, res.F, g->M, df2, lnb, res.p);
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}


int mmg_emmax_scan_device(mmg_ctx* ctx, mmg_geno* g, double h0_rss, int32_t df2) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g != nullptr && df2 > 0);
  if (!ctx->model.Bq) return set_err(ctx, MMG_E_STATE, "mmg_scan_set_model has not been called");
  if (ctx->model.N != g->N) return set_err(ctx, MMG_E_ARG, "model N does not match the genotype store");
  int rc = ensure_result(ctx, ctx->res, g->Mpad);
  if (rc) return rc;
  ctx->res.M = g->M;
  ctx->res.geno = g; ctx->res.geno_version = g->version;
  if (g->M == 0) return MMG_OK;
  return scan_into(ctx, g, ctx->model, ctx->res, h0_rss, df2, false, true);
}

int mmg_scan_last_stats(mmg_ctx* ctx, int32_t* adaptive, int64_t* n_refined, double* eps_max, double* sigma_ratio_max,
                        int32_t* fell_back) {
  MMG_ENTER(ctx);
  if (adaptive) *adaptive = ctx->res.adaptive;
  if (n_refined) *n_refined = ctx->res.n_refined;
  if (eps_max) *eps_max = ctx->res.eps_max;
  if (sigma_ratio_max) *sigma_ratio_max = ctx->res.sigma_ratio_max;
  if (fell_back) *fell_back = ctx->res.fell_back;
  return MMG_OK;
}

int mmg_scan_last_exact(mmg_ctx* ctx, int64_t* n_exact) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, n_exact != nullptr);
  *n_exact = ctx->res.n_exact;
  return MMG_OK;
}

static int fetch(mmg_ctx* ctx, double* dst, const double* src, int64_t M) {
  if (!dst || M == 0) return MMG_OK;
  MMG_HIP(ctx, hipMemcpyAsync(dst, src, M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  return MMG_OK;
}

int mmg_scan_fetch(mmg_ctx* ctx, int64_t M, double* rss, double* F, double* p) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, M == ctx->res.M);
  int rc;
  if ((rc = fetch(ctx, rss, ctx->res.rss, M))) return rc;
  if ((rc = fetch(ctx, F, ctx->res.F, M))) return rc;
  if ((rc = fetch(ctx, p, ctx->res.p, M))) return rc;
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int mmg_scan_fetch_stats(mmg_ctx* ctx, int64_t M, double* dot, double* den, double* sum) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, M == ctx->res.M);
  int rc;
  if ((rc = fetch(ctx, dot, ctx->res.dot, M))) return rc;
  if ((rc = fetch(ctx, den, ctx->res.den, M))) return rc;
  if ((rc = fetch(ctx, sum, ctx->res.sum, M))) return rc;
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int mmg_emmax_scan(mmg_ctx* ctx, mmg_geno* g, double h0_rss, int32_t df2, double* rss, double* F, double* p) {
  int rc = mmg_emmax_scan_device(ctx, g, h0_rss, df2);
  if (rc) return rc;
  return mmg_scan_fetch(ctx, g->M, rss, F, p);
}

int mmg_emmax_scan_i8(mmg_ctx* ctx, const int8_t* snps, int64_t M, int32_t N, const double* A, const double* w,
                      double h0_rss, int32_t df2, double* rss, double* F, double* p) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, snps && M >= 0 && N > 0);
  int rc = mmg_scan_set_model(ctx, N, A, w, 0);
  if (rc) return rc;
  mmg_geno* g = nullptr;
  rc = mmg_geno_create(ctx, M, N, &g);
  if (rc) return rc;
  rc = mmg_geno_upload(ctx, g, snps, 0, M);
  if (rc == MMG_OK) rc = mmg_emmax_scan(ctx, g, h0_rss, df2, rss, F, p);
  mmg_geno_destroy(ctx, g);
  return rc;
}

// One-shot twins over caller-owned host genotypes: float32 genotypes (the C3 config's "fp32 [M x N]") and
// the permutation test.  Each uploads into a temporary store and calls the resident form.
extern "C++" {
template <typename F>
static int with_temp_geno_f32(mmg_ctx* ctx, const float* snps, int64_t M, int32_t N, F&& body) {
  mmg_geno* g = nullptr;
  int rc = mmg_geno_create(ctx, M, N, &g);
  if (rc) return rc;
  rc = mmg_geno_upload_f32(ctx, g, snps, 0, M);
  if (rc == MMG_OK) rc = body(g);
  mmg_geno_destroy(ctx, g);
  return rc;
}
}  // extern C++

int mmg_kinship_f32(mmg_ctx* ctx, const float* snps, int64_t M, int32_t N, const float* scale, const float* shift,
                    double* C_out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, snps && C_out && M > 0 && N > 0 && (scale == nullptr) == (shift == nullptr));
  return with_temp_geno_f32(ctx, snps, M, N, [&](mmg_geno* g) {
    if (scale) return mmg_kinship_affine_f32(ctx, g, scale, shift, C_out);
    std::vector<int64_t> c64((size_t)N * N);
    int rc = mmg_kinship_ibs_i8(ctx, g, c64.data());
    if (rc == MMG_OK) for (size_t i = 0; i < c64.size(); ++i) C_out[i] = (double)c64[i];
    return rc;
  });
}

int mmg_emmax_scan_f32(mmg_ctx* ctx, const float* snps, int64_t M, int32_t N, const double* A, const double* w,
                       double h0_rss, int32_t df2, double* rss, double* F, double* p) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, snps && M >= 0 && N > 0);
  int rc = mmg_scan_set_model(ctx, N, A, w, 0);
  if (rc) return rc;
  if (M == 0) return MMG_OK;
  return with_temp_geno_f32(ctx, snps, M, N, [&](mmg_geno* g) { return mmg_emmax_scan(ctx, g, h0_rss, df2, rss, F, p); });
}

int mmg_emmax_perm_i8(mmg_ctx* ctx, const int8_t* snps, int64_t M, int32_t N, const double* Ht, const double* Ys,
                      int32_t P, double h0_rss, double* min_rss) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, snps && M >= 0 && N > 0);
  mmg_geno* g = nullptr;
  int rc = mmg_geno_create(ctx, M, N, &g);
  if (rc) return rc;
  if (M > 0) rc = mmg_geno_upload(ctx, g, snps, 0, M);
  if (rc == MMG_OK) rc = mmg_emmax_perm(ctx, g, N, Ht, Ys, P, h0_rss, 0, min_rss);
  mmg_geno_destroy(ctx, g);
  return rc;
}

int mmg_geno_matvec(mmg_ctx* ctx, mmg_geno* g, const double* V, int32_t nv, double* out) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && V && out && nv > 0);
  if (g->M == 0) return MMG_OK;
  double *dv = nullptr, *dout = nullptr;
  MMG_HIP(ctx, sc.alloc(&dv, g->Npad * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dout, g->M * sizeof(double)));
  for (int k = 0; k < nv; ++k) {
    MMG_HIP(ctx, hipMemsetAsync(dv, 0, g->Npad * sizeof(double), ctx->stream));
    MMG_HIP(ctx, hipMemcpyAsync(dv, V + (int64_t)k * g->N, g->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_snp_dot(ctx, g, dv, dout);
    MMG_HIP(ctx, hipGetLastError());
    MMG_HIP(ctx, hipMemcpyAsync(out + (int64_t)k * g->M, dout, g->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return MMG_OK;
}

int mmg_f_sf(mmg_ctx* ctx, const double* F, int64_t n, int32_t df2, double* p) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, F && p && n >= 0 && df2 > 0);
  if (n == 0) return MMG_OK;
  double *dF = nullptr, *dp = nullptr;
  MMG_HIP(ctx, sc.alloc(&dF, n * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dp, n * sizeof(double)));
  MMG_HIP(ctx, hipMemcpyAsync(dF, F, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  launch_f_sf(ctx, dF, n, df2, ln_beta_half(0.5 * df2), dp);
  MMG_HIP(ctx, hipGetLastError());
  MMG_HIP(ctx, hipMemcpyAsync(p, dp, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

// ------------------------------------------------------------------------- permutations
// row-major C[M x N] = op(A) op(B) on device pointers
static int dgemm_dev(mmg_ctx* ctx, int ta, int tb, int M, int N, int K, const double* dA, const double* dB, double* dC) {
  rocblas_handle h;
  int rc = get_rocblas(ctx, &h);
  if (rc) return rc;
  const double one = 1.0, zero = 0.0;
  MMG_RB(ctx, rocblas_dgemm_64(h, tb ? rocblas_operation_transpose : rocblas_operation_none,
                            ta ? rocblas_operation_transpose : rocblas_operation_none, N, M, K, &one, dB,
                            tb ? K : N, dA, ta ? M : K, &zero, dC, N));
  return MMG_OK;
}

// ------------------------------------------------------------------------- permutation plan
// Everything of _emmax_permutations_ that does not depend on the SNPs (linear_models.py:1135-1156 + the operand images
// of the two GEMMs): H, W' = Ys'H as 4-digit int8 image, the quadratic-form model of H'H, v = H'H 1, c0 = 1'H'H 1,
// Ys_p.Ys_p.  Built once per (H, Ys) -- per phenotype -- and run over any number of genotype stores (chunks).
struct mmg_perm_plan {
  int32_t N = 0, Npad = 0, P = 0, Ppad = 0;
  double h0_rss = 0.0, c0 = 0.0;
  std::vector<double> yy;
  int8_t* Wq = nullptr;
  double *dstep = nullptr, *dcsum = nullptr, *dv = nullptr /*[Npad] v = H'H 1 (zero padded)*/;
  mmg_scan_model pm;             // digit planes of H'H (t.t quadratic form), w slot = v
  mmg_scan_result pr;            // per-SNP work arrays of the quadratic form
  double *dmu = nullptr, *dinv = nullptr, *dmax = nullptr, *dvecs = nullptr, *ddots = nullptr;
  double* dssum = nullptr;       // [cap] exact genotype sum per SNP, 0 in the padding rows (digit offset of the GEMM operand)
  int64_t cap = 0;               // SNP capacity of dmu / dinv / ddots / dssum
  int qcap = 0;
  bool centred = true;           // SNPs are mean-centred before the transform (:1159); false: t = Ht s as given
};

static void perm_plan_free(mmg_perm_plan* p) {
  hipFree(p->Wq); hipFree(p->dstep); hipFree(p->dcsum); hipFree(p->dv); hipFree(p->dmu); hipFree(p->dinv); hipFree(p->dmax);
  hipFree(p->dvecs); hipFree(p->ddots); hipFree(p->dssum);
  free_model(p->pm); free_result(p->pr);
  delete p;
}

int mmg_perm_plan_create(mmg_ctx* ctx, int32_t N, const double* Ht, const double* Ys, int32_t P, double h0_rss,
                         mmg_perm_plan** out) {
  return mmg_perm_plan_create_ex(ctx, N, Ht, Ys, P, h0_rss, 0, out);
}

// The plan from H on the device.  dH: [N x N]; h_transposed: the buffer is H' row-major (= H column-major: the L^-1 a REML
// workspace holds).  flags bit 0: no SNP centring; bit 1: H <- C H first (the public permutation test centres the TRANSFORMED
// SNP, linear_models.py:1211 -- column-centring of H; dH is modified).
static int perm_plan_build(mmg_ctx* ctx, int32_t N, double* dH, bool h_transposed, const double* Ys, int32_t P, double h0_rss,
                           int flags, mmg_perm_plan** out) {
  Scratch sc;
  *out = nullptr;
  mmg_perm_plan* p = new mmg_perm_plan();
  p->centred = !(flags & 1);
  p->N = N; p->Npad = (int32_t)round_up(N, 256); p->P = P; p->Ppad = (int)round_up(P, 64); p->h0_rss = h0_rss;
  p->yy.assign((size_t)P, 0.0);
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < P; ++k) p->yy[k] += Ys[(size_t)i * P + k] * Ys[(size_t)i * P + k];
  double *dYs = nullptr, *dA = nullptr, *dWt = nullptr, *dones = nullptr, *dh1 = nullptr;
  const int nPT = p->Ppad / 64;
  hipError_t e = hipMalloc(&p->Wq, (size_t)nPT * TM * p->Npad);
  if (e == hipSuccess) e = hipMalloc(&p->dstep, p->Ppad * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&p->dcsum, p->Ppad * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&p->dv, p->Npad * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&p->dmax, p->Ppad * sizeof(double));
  if (e == hipSuccess) e = sc.alloc(&dYs, (size_t)N * P * sizeof(double));
  if (e == hipSuccess) e = sc.alloc(&dA, (size_t)N * N * sizeof(double));
  if (e == hipSuccess) e = sc.alloc(&dWt, (size_t)P * N * sizeof(double));
  if (e == hipSuccess) e = sc.alloc(&dones, N * sizeof(double));
  if (e == hipSuccess) e = sc.alloc(&dh1, N * sizeof(double));
  if (e != hipSuccess) { perm_plan_free(p); return set_err(ctx, MMG_E_NOMEM, std::string("hipMalloc permutation plan: ") + hipGetErrorString(e)); }
  int rc = MMG_OK;
  auto fail = [&](int code) { perm_plan_free(p); return code; };
  std::vector<double> ones((size_t)N, 1.0), h1((size_t)N);
  if (hipMemcpyAsync(dYs, Ys, (size_t)N * P * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(dones, ones.data(), N * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemsetAsync(p->dv, 0, p->Npad * sizeof(double), ctx->stream) != hipSuccess)
    return fail(set_err(ctx, MMG_E_HIP, "permutation plan: upload"));
  const int T = h_transposed ? 1 : 0;                                      // op(dH) = H either way
  if (flags & 2) {
    // C H: every column of H minus its mean = every ROW of the transposed buffer; of the plain one, the mean row is taken off
    if (h_transposed) launch_center_rows(ctx, dH, N, N);
    else {
      double* dmean = nullptr;
      if (sc.alloc(&dmean, N * sizeof(double)) != hipSuccess) return fail(set_err(ctx, MMG_E_NOMEM, "hipMalloc permutation plan"));
      rc = dgemm_dev(ctx, 1, 0, N, 1, N, dH, dones, dmean);                  // H'1 = column sums
      if (rc) return fail(rc);
      launch_sub_row_mean(ctx, dH, N, dmean);
    }
  }
  rc = dgemm_dev(ctx, 1 - T, T, N, N, N, dH, dH, dA);                       // A' = H'H      (t.t = s~' A' s~, :1160,1163)
  if (rc == MMG_OK) rc = dgemm_dev(ctx, 1, T, P, N, N, dYs, dH, dWt);      // W' = Ys'H [P x N]  (t.Ys_p = s~ . W_p)
  if (rc == MMG_OK) rc = dgemm_dev(ctx, T, 0, N, 1, N, dH, dones, dh1);    // H 1
  if (rc == MMG_OK) rc = dgemm_dev(ctx, 1 - T, 0, N, 1, N, dH, dh1, p->dv);   // v = H'(H 1) = A' 1
  if (rc) return fail(rc);
  if (hipMemcpyAsync(h1.data(), dh1, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(set_err(ctx, MMG_E_HIP, "permutation plan: H 1"));
  for (int i = 0; i < N; ++i) p->c0 += h1[i] * h1[i];                       // 1'H'H 1
  // centred operands (k_perm.hip): A'' = C A' C is the model of the stand-alone test's t.t (adaptive digit schedule
  // as in the scan unless MMG_SCAN_ADAPTIVE=0), W'' = C W the GEMM operand of both paths (s~.W_p = s.(C W_p))
  if (p->centred) {
    launch_center_sym(ctx, dA, N, p->dv, p->c0);
    launch_center_rows(ctx, dWt, N, P);
  }
  {
    const char* e = std::getenv("MMG_SCAN_ADAPTIVE");
    rc = model_from_device(ctx, p->pm, N, dA, p->dv, 4, !(e && e[0] == '0'));
  }
  if (rc == MMG_OK) rc = quantize_rows_4digits(ctx, dWt, N, p->Npad, P, p->Wq, p->dstep, p->dcsum);
  if (rc == MMG_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(ctx, MMG_E_HIP, "permutation plan: setup");
  if (rc) return fail(rc);
  *out = p;
  return MMG_OK;
}

int mmg_perm_plan_create_ex(mmg_ctx* ctx, int32_t N, const double* Ht, const double* Ys, int32_t P, double h0_rss,
                            int flags, mmg_perm_plan** out) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, out && Ht && Ys && N > 0 && P > 0 && (flags & ~3) == 0);
  double* dH = nullptr;
  if (sc.alloc(&dH, (size_t)N * N * sizeof(double)) != hipSuccess) return set_err(ctx, MMG_E_NOMEM, "hipMalloc permutation plan");
  MMG_HIP(ctx, hipMemcpyAsync(dH, Ht, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return perm_plan_build(ctx, N, dH, false, Ys, P, h0_rss, flags, out);
}

}  // extern "C" (reopened below)
namespace mmg { int reml_linv_device_opaque(mmg_ctx* ctx, mmg_reml* r, double delta, const double** dLinv, int32_t* N); }
extern "C" {

// The same plan with H = L^-1 of K + delta I = L L', taken from a REML workspace as it lies in HBM (what the scan model of the
// same delta left there, or one factorisation + triangular inverse): the permutation test without an eigendecomposition and
// without an N x N matrix crossing PCIe.  Any H with H'H = (K + delta I)^-1 is a valid H_sqrt_inv (linear_models.py:898 is fixed
// only up to LAPACK's eigenvector signs); the shuffled residuals Ys live in the basis of the H that is used, so they come from
// mmg_reml_linv_apply of the same workspace.
int mmg_perm_plan_create_from_reml(mmg_ctx* ctx, mmg_reml* r, double delta, const double* Ys, int32_t P, double h0_rss,
                                   int flags, mmg_perm_plan** out) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, out && r && Ys && P > 0 && (flags & ~3) == 0);
  const double* dLinv = nullptr;
  int32_t N = 0;
  int rc = reml_linv_device_opaque(ctx, r, delta, &dLinv, &N);
  if (rc) return rc;
  double* dH = nullptr;                                                      // a copy: the builder may centre it
  if (sc.alloc(&dH, (size_t)N * N * sizeof(double)) != hipSuccess) return set_err(ctx, MMG_E_NOMEM, "hipMalloc permutation plan");
  MMG_HIP(ctx, hipMemcpyAsync(dH, dLinv, (size_t)N * N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return perm_plan_build(ctx, N, dH, true, Ys, P, h0_rss, flags, out);
}

int mmg_perm_plan_destroy(mmg_ctx* ctx, mmg_perm_plan* p) {
  if (!p) return MMG_OK;
  MMG_NOTE_ENTRY();
  if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
  perm_plan_free(p);
  return MMG_OK;
}

int mmg_perm_plan_run(mmg_ctx* ctx, mmg_comm* comm, mmg_perm_plan* p, mmg_geno* g, const double* HtQ, int32_t q,
                      double* min_rss) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, p && g && min_rss && g->N == p->N && (HtQ == nullptr || q >= 1));
  const bool reduce = comm && comm->world > 1;
  const bool reuse = HtQ != nullptr;
  if (reuse && !p->centred)
    return set_err(ctx, MMG_E_ARG, "mmg_perm_plan_run: the after-scan form needs a plan with centred SNPs");
  if (reuse && (ctx->res.geno != g || ctx->res.geno_version != g->version || ctx->res.M != g->M || ctx->model.N != p->N))
    return set_err(ctx, MMG_E_STATE, "mmg_perm_plan_run: the last mmg_emmax_scan_device of this context was not over this "
                                     "genotype store in its current state");
  MMG_HIP(ctx, hipMemsetAsync(p->dmax, 0, p->Ppad * sizeof(double), ctx->stream));
  if (g->M > 0) {
    if (p->cap < g->Mpad || p->qcap < (reuse ? q : 0)) {
      hipFree(p->dmu); hipFree(p->dinv); hipFree(p->ddots); hipFree(p->dvecs); hipFree(p->dssum);
      p->dmu = p->dinv = p->ddots = p->dvecs = p->dssum = nullptr; p->cap = 0;
      const int qq = std::max(reuse ? q : 0, p->qcap);
      MMG_HIP(ctx, hipMalloc(&p->dmu, g->Mpad * sizeof(double)));
      MMG_HIP(ctx, hipMalloc(&p->dinv, g->Mpad * sizeof(double)));
      MMG_HIP(ctx, hipMalloc(&p->dssum, g->Mpad * sizeof(double)));
      MMG_HIP(ctx, hipMalloc(&p->ddots, (size_t)(1 + qq) * g->Mpad * sizeof(double)));
      MMG_HIP(ctx, hipMalloc(&p->dvecs, (size_t)(1 + qq) * p->Npad * sizeof(double)));
      p->cap = g->Mpad; p->qcap = qq;
    }
    if (reuse) {
      // t.t from the quadratic forms of the scan that just ran over g with the same H (see mmg_emmax_perm_after_scan)
      MMG_HIP(ctx, hipMemsetAsync(p->dvecs, 0, (size_t)(1 + q) * p->Npad * sizeof(double), ctx->stream));
      MMG_HIP(ctx, hipMemcpyAsync(p->dvecs, p->dv, p->Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      for (int c = 0; c < q; ++c)
        MMG_HIP(ctx, hipMemcpyAsync(p->dvecs + (size_t)(1 + c) * p->Npad, HtQ + (size_t)c * p->N, p->N * sizeof(double),
                                    hipMemcpyHostToDevice, ctx->stream));
      for (int k = 0; k < 1 + q; ++k) launch_snp_dot(ctx, g, p->dvecs + (size_t)k * p->Npad, p->ddots + (size_t)k * g->Mpad);
      launch_perm_center_reuse(ctx, g, ctx->res.den, p->ddots, q, ctx->res.sum, p->c0, p->dmu, p->dinv);
    } else {
      int rc = ensure_result(ctx, p->pr, g->Mpad);
      if (rc) return rc;
      p->pr.M = g->M;
      rc = scan_into(ctx, g, p->pm, p->pr, 1.0, 1, true, false);         // den = s'(C A' C)s = t.t   (:1159-1163)
      if (rc) return rc;
      launch_perm_inv(ctx, g, p->pr, p->dmu, p->dinv);                    // 1 / t.t, mu = 0
    }
    MMG_HIP(ctx, hipGetLastError());
    // the finalize kernels of the scan wrote sum(s) of the M real SNPs (exact integers); the padding rows stay 0
    MMG_HIP(ctx, hipMemsetAsync(p->dssum, 0, g->Mpad * sizeof(double), ctx->stream));
    MMG_HIP(ctx, hipMemcpyAsync(p->dssum, reuse ? ctx->res.sum : p->pr.sum, g->M * sizeof(double), hipMemcpyDeviceToDevice,
                                ctx->stream));
    int rc = run_perm_q(ctx, g, p->Wq, p->dstep, p->dcsum, p->P, p->dinv, p->dmu, p->dssum, p->dmax);
    if (rc) return rc;
  }
  if (reduce)   // per-permutation maxima of this rank's SNP block stay in HBM: RCCL MAX over xGMI (exact, order independent)
    MMG_NCCL(ctx, ncclAllReduce(p->dmax, p->dmax, (size_t)p->Ppad, ncclDouble, ncclMax, comm->comm, ctx->stream));
  std::vector<double> mx((size_t)p->Ppad, 0.0);
  MMG_HIP(ctx, hipMemcpyAsync(mx.data(), p->dmax, p->Ppad * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < p->P; ++k) min_rss[k] = std::min(p->h0_rss, p->yy[k] - mx[k]);   // :1164 running min from h0_rss (:1156)
  return MMG_OK;
}

int mmg_emmax_perm(mmg_ctx* ctx, mmg_geno* g, int32_t N, const double* Ht, const double* Ys, int32_t P, double h0_rss,
                   int ndigits, double* min_rss) {
  return mmg_emmax_perm_sharded(ctx, nullptr, g, N, Ht, Ys, P, h0_rss, ndigits, min_rss);
}

int mmg_emmax_perm_sharded(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int32_t N, const double* Ht, const double* Ys,
                           int32_t P, double h0_rss, int ndigits, double* min_rss) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && Ht && Ys && min_rss && P > 0 && N == g->N);
  MMG_CHECK_ARG(ctx, ndigits == 0 || ndigits == 4);
  mmg_perm_plan* plan = nullptr;
  int rc = mmg_perm_plan_create(ctx, N, Ht, Ys, P, h0_rss, &plan);
  if (rc) return rc;
  rc = mmg_perm_plan_run(ctx, comm, plan, g, nullptr, 0, min_rss);
  mmg_perm_plan_destroy(ctx, plan);
  return rc;
}

int mmg_emmax_perm_after_scan(mmg_ctx* ctx, mmg_comm* comm, mmg_geno* g, int32_t N, const double* Ht, const double* Ys,
                              int32_t P, double h0_rss, const double* HtQ, int32_t q, double* min_rss) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, g && Ht && Ys && HtQ && min_rss && P > 0 && q >= 1 && N == g->N);
  mmg_perm_plan* plan = nullptr;
  int rc = mmg_perm_plan_create(ctx, N, Ht, Ys, P, h0_rss, &plan);
  if (rc) return rc;
  rc = mmg_perm_plan_run(ctx, comm, plan, g, HtQ, q, min_rss);
  mmg_perm_plan_destroy(ctx, plan);
  return rc;
}

// ------------------------------------------------------------------------- eigen-rotated store, multi-phenotype scan
struct mmg_rot {
  int32_t N = 0, Npad = 0, nVT = 0;
  int64_t Mcap = 0;              // SNP capacity (multiple of 256)
  int64_t M = 0;                 // SNPs currently loaded
  int8_t* Vq = nullptr;          // [nVT][256][Npad] digits of the eigenvectors (operand layout of k_perm.hip)
  double* dstep = nullptr;       // [nVT*64] per-eigenvector step
  double *dssum = nullptr, *dones = nullptr;   // [Mcap] genotype sum per loaded SNP (digit offset of the operand); [Npad] ones
  double* T = nullptr;           // [Mcap/256][nVT*64][256] fp64: T[m/256][i][m%256] = u_i . s_m
};

int mmg_rot_create(mmg_ctx* ctx, int32_t N, const double* evecs_rows, int64_t M_cap, mmg_rot** out) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, out && evecs_rows && N > 0 && M_cap > 0);
  *out = nullptr;
  mmg_rot* r = new mmg_rot();
  r->N = N; r->Npad = (int32_t)round_up(N, 256); r->nVT = (N + 63) / 64;
  r->Mcap = round_up(M_cap, 256);
  double *dV = nullptr, *dcsum = nullptr;
  hipError_t e = hipMalloc(&r->T, (size_t)r->nVT * 64 * r->Mcap * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&r->Vq, (size_t)r->nVT * TM * r->Npad);
  if (e == hipSuccess) e = hipMalloc(&r->dstep, (size_t)r->nVT * 64 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&r->dssum, (size_t)r->Mcap * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&r->dones, (size_t)r->Npad * sizeof(double));
  if (e == hipSuccess) e = sc.alloc(&dV, (size_t)N * N * sizeof(double));
  if (e == hipSuccess) e = sc.alloc(&dcsum, (size_t)r->nVT * 64 * sizeof(double));
  auto release = [&]() { hipFree(r->T); hipFree(r->Vq); hipFree(r->dstep); hipFree(r->dssum); hipFree(r->dones); delete r; };
  if (e != hipSuccess) {
    release();
    return set_err(ctx, MMG_E_NOMEM, std::string("hipMalloc rotated store: ") + hipGetErrorString(e));
  }
  int rc = MMG_OK;
  std::vector<double> ones((size_t)r->Npad, 0.0);
  std::fill(ones.begin(), ones.begin() + N, 1.0);
  e = hipMemcpyAsync(dV, evecs_rows, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(r->dones, ones.data(), (size_t)r->Npad * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) rc = set_err(ctx, MMG_E_HIP, hipGetErrorString(e));
  if (rc == MMG_OK) rc = quantize_rows_4digits(ctx, dV, N, r->Npad, N, r->Vq, r->dstep, dcsum);
  if (rc == MMG_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(ctx, MMG_E_HIP, "rotated store setup");
  if (rc) { release(); return rc; }
  *out = r;
  return MMG_OK;
}

int mmg_rot_destroy(mmg_ctx* ctx, mmg_rot* r) {
  if (!r) return MMG_OK;
  MMG_NOTE_ENTRY();
  if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
  hipFree(r->T); hipFree(r->Vq); hipFree(r->dstep); hipFree(r->dssum); hipFree(r->dones);
  delete r;
  return MMG_OK;
}

int mmg_rot_load(mmg_ctx* ctx, mmg_rot* r, mmg_geno* g) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, r && g && g->N == r->N && g->Mpad <= r->Mcap);
  // |sum_k digit * s| <= 127 * smax * Npad must fit the int32 accumulators of the digit GEMM
  MMG_CHECK_ARG(ctx, (int64_t)128 * std::max(g->smax, 1) * r->Npad < ((int64_t)1 << 31));
  r->M = g->M;
  if (g->M == 0) return MMG_OK;
  // sum(s) per SNP (exact integers in fp64; 0 for the padding rows): the rotation GEMM's operand digits are shifted into
  // the non-negative range and its epilogue takes 2^27 sum(s) back out
  MMG_HIP(ctx, hipMemsetAsync(r->dssum, 0, (size_t)g->Mpad * sizeof(double), ctx->stream));
  launch_snp_dot(ctx, g, r->dones, r->dssum);
  MMG_HIP(ctx, hipGetLastError());
  int rc = run_rotate(ctx, g, r->Vq, r->dstep, r->dssum, r->nVT, r->T);
  if (rc) return rc;
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int mmg_rot_fetch(mmg_ctx* ctx, mmg_rot* r, int64_t m0, int64_t rows, double* out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, r && out && m0 >= 0 && rows >= 0 && m0 + rows <= r->M);
  if (rows == 0) return MMG_OK;
  // out[i][k] = T[(m0+k)/256][i][(m0+k)%256], i < N: one strided 2-D copy per 256-SNP block touched
  const int64_t nrows = (int64_t)r->nVT * 64;
  for (int64_t m = m0; m < m0 + rows;) {
    const int64_t sb = m / 256, o = m % 256, w = std::min<int64_t>(256 - o, m0 + rows - m);
    MMG_HIP(ctx, hipMemcpy2DAsync(out + (m - m0), rows * sizeof(double), r->T + sb * nrows * 256 + o, 256 * sizeof(double),
                                  w * sizeof(double), r->N, hipMemcpyDeviceToHost, ctx->stream));
    m += w;
  }
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int mmg_emmax_scan_multi(mmg_ctx* ctx, mmg_rot* r, int32_t P, int32_t q, const double* d, const double* omega,
                         const double* G, const double* h0_rss, int32_t df2, double* rss, double* F, double* p) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, r && d && omega && G && h0_rss && P > 0 && q >= 1 && q <= 8 && df2 > 0);
  const int64_t M = r->M;
  if (M == 0) return MMG_OK;
  const int N = r->N;
  ctx->multi_ms_total = 0.0;
  const int64_t ldOut = round_up(M, 256);
  // 16 phenotypes per pass with the linear columns on the fp64 matrix pipe (k_rot.hip:scan_multi_mfma_kernel), 8 with
  // the all-VALU kernel (MMG_MULTI_KERNEL=valu, MMG_MULTI_PB=8)
  int PBmax = q <= 2 ? 16 : 8;
  if (const char* e = std::getenv("MMG_MULTI_KERNEL")) if (std::string(e) == "valu") PBmax = 8;
  if (const char* e = std::getenv("MMG_MULTI_PB")) if (std::atoi(e) == 8) PBmax = 8;
  const int NCmax = PBmax * (2 + q);
  // Two sets of coefficient / output buffers: the download of a batch's results (second stream; a blocking copy when
  // the caller's arrays are pageable) runs while the next batch's pass is on the device.
  double *dcoef[2] = {nullptr, nullptr}, *dh0[2] = {nullptr, nullptr}, *dout[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  double* host[3] = {rss, F, p};
  const int nbatch = (P + PBmax - 1) / PBmax;
  const int nset = nbatch > 1 ? 2 : 1;
  for (int sI = 0; sI < nset; ++sI) {
    MMG_HIP(ctx, sc.alloc(&dcoef[sI], (size_t)N * NCmax * sizeof(double)));
    MMG_HIP(ctx, sc.alloc(&dh0[sI], PBmax * sizeof(double)));
    for (int k = 0; k < 3; ++k)
      if (host[k]) MMG_HIP(ctx, sc.alloc(&dout[sI][k], (size_t)PBmax * ldOut * sizeof(double)));
  }
  std::vector<double> coef[2], h0b[2];
  for (int sI = 0; sI < nset; ++sI) { coef[sI].resize((size_t)N * NCmax); h0b[sI].resize(PBmax); }
  const double lnb = ln_beta_half(0.5 * df2);
  struct Events {                                      // per batch: kernel start / end; per set: results downloaded
    std::vector<hipEvent_t> k0, k1;
    hipEvent_t done[2] = {nullptr, nullptr};
    ~Events() { for (auto e : k0) hipEventDestroy(e); for (auto e : k1) hipEventDestroy(e); for (auto e : done) if (e) hipEventDestroy(e); }
  } evs;
  evs.k0.resize(nbatch); evs.k1.resize(nbatch);
  for (int b = 0; b < nbatch; ++b) { MMG_HIP(ctx, hipEventCreate(&evs.k0[b])); MMG_HIP(ctx, hipEventCreate(&evs.k1[b])); }
  for (int sI = 0; sI < nset; ++sI) MMG_HIP(ctx, hipEventCreateWithFlags(&evs.done[sI], hipEventDisableTiming));
  auto download = [&](int b) -> int {
    const int sI = b & (nset - 1), p0 = b * PBmax, nb = std::min(PBmax, P - p0);
    MMG_HIP(ctx, hipStreamWaitEvent(ctx->stream2, evs.k1[b], 0));
    for (int k = 0; k < 3; ++k)
      if (host[k])
        MMG_HIP(ctx, hipMemcpy2DAsync(host[k] + (size_t)p0 * M, M * sizeof(double), dout[sI][k], ldOut * sizeof(double),
                                      M * sizeof(double), nb, hipMemcpyDeviceToHost, ctx->stream2));
    MMG_HIP(ctx, hipEventRecord(evs.done[sI], ctx->stream2));
    return MMG_OK;
  };
  for (int b = 0; b < nbatch; ++b) {
    const int sI = b & (nset - 1), p0 = b * PBmax;
    const int nb = std::min(PBmax, P - p0);
    int PB = q > 4 ? 8 : 1;                          // more than 4 fixed-effect columns: always the 8-wide matrix-pipe kernel
    while (PB < nb) PB *= 2;                         // 1, 2, 4, 8, 16: unused columns carry zero coefficients
    const int NC = PB * (2 + q);
    std::vector<double>& cf = coef[sI];
    if (b >= 2) MMG_HIP(ctx, hipEventSynchronize(evs.done[sI]));   // batch b-2 has left this set's buffers (host image too)
    std::fill(cf.begin(), cf.begin() + (size_t)N * NC, 0.0);
    for (int k = 0; k < PB; ++k) h0b[sI][k] = k < nb ? h0_rss[p0 + k] : 1.0;
    for (int k = 0; k < nb; ++k) {
      const double* dk = d + (size_t)(p0 + k) * N;
      const double* wk = omega + (size_t)(p0 + k) * N;
      const double* gk = G + (size_t)(p0 + k) * q * N;
      for (int i = 0; i < N; ++i) {
        double* row = cf.data() + (size_t)i * NC;
        row[k] = dk[i];
        row[PB + k * (1 + q)] = wk[i];
        for (int c = 0; c < q; ++c) row[PB + k * (1 + q) + 1 + c] = gk[(size_t)c * N + i];
      }
    }
    MMG_HIP(ctx, hipMemcpyAsync(dcoef[sI], cf.data(), (size_t)N * NC * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipMemcpyAsync(dh0[sI], h0b[sI].data(), PB * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MMG_HIP(ctx, hipEventRecord(evs.k0[b], ctx->stream));
    int rc = run_scan_multi(ctx, r->T, (int64_t)r->nVT * 64, N, M, PB, q, dcoef[sI], dh0[sI], df2, lnb, dout[sI][0],
                            dout[sI][1], dout[sI][2], ldOut);
    if (rc) return rc;
    MMG_HIP(ctx, hipEventRecord(evs.k1[b], ctx->stream));
    if (b >= 1) { rc = download(b - 1); if (rc) return rc; }      // overlaps the pass just queued
  }
  { int rc = download(nbatch - 1); if (rc) return rc; }
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream2));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double ms_total = 0.0;
  for (int b = 0; b < nbatch; ++b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, evs.k0[b], evs.k1[b]) == hipSuccess) ms_total += ms;
  }
  ctx->multi_ms_total = ms_total;
  return MMG_OK;
}

// ------------------------------------------------------------------------- RCCL
int mmg_comm_unique_id(unsigned char id[128]) {
  ncclUniqueId uid;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclResult_t r = ncclGetUniqueId(&uid);
  if (r != ncclSuccess) return set_err(nullptr, MMG_E_LIB, ncclGetErrorString(r));
  std::memcpy(id, &uid, 128);
  return MMG_OK;
}

int mmg_comm_create(mmg_ctx* ctx, const unsigned char id[128], int rank, int world, mmg_comm** out) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, id && out && world >= 1 && rank >= 0 && rank < world);
  ncclUniqueId uid;
  std::memcpy(&uid, id, 128);
  mmg_comm* c = new mmg_comm();
  c->rank = rank; c->world = world;
  MMG_HIP(ctx, hipSetDevice(ctx->device));
  ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) { delete c; return set_err(ctx, MMG_E_LIB, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
  fflush(stdout);   // RCCL prints a version banner through stdio; do not let it trail the caller's output
  *out = c;
  return MMG_OK;
}

int mmg_comm_destroy(mmg_ctx* ctx, mmg_comm* c) {
  if (!c) return MMG_OK;
  MMG_NOTE_ENTRY();
  if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); hipStreamSynchronize(ctx->stream2); ctx->deliver_pending = false; }
  if (c->comm) ncclCommDestroy(c->comm);
  fflush(stdout);
  delete c;
  return MMG_OK;
}

int mmg_scan_deliver_wait(mmg_ctx* ctx) {
  MMG_ENTER(ctx);
  if (!ctx->deliver_pending) return MMG_OK;
  ctx->deliver_pending = false;
  MMG_HIP(ctx, hipEventSynchronize(ctx->ev_deliver));
  return MMG_OK;
}

int mmg_scan_deliver_begin(mmg_ctx* ctx, mmg_comm* c, int64_t count, double* rss, double* F, double* p) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, count >= 0 && count <= ctx->res.cap);
  int rc = mmg_scan_deliver_wait(ctx);              // one delivery in flight: the staging is reused
  if (rc) return rc;
  if (count == 0) return MMG_OK;
  const int world = c ? c->world : 1;
  // snapshot [3][count] (so the next scan may overwrite the result arrays) + gathered [world][3][count]
  const size_t need = (size_t)(world + 1) * 3 * count;
  if (ctx->dstage_elems < need) {
    hipFree(ctx->dstage);
    ctx->dstage = nullptr; ctx->dstage_elems = 0;
    MMG_HIP(ctx, hipMalloc(&ctx->dstage, need * sizeof(double)));
    ctx->dstage_elems = need;
  }
  double* pack = ctx->dstage;
  double* gathered = c ? ctx->dstage + (size_t)3 * count : pack;
  const double* srcs[3] = {ctx->res.rss, ctx->res.F, ctx->res.p};
  double* dsts[3] = {rss, F, p};
  for (int k = 0; k < 3; ++k)
    MMG_HIP(ctx, hipMemcpyAsync(pack + (size_t)k * count, srcs[k], count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  MMG_HIP(ctx, hipEventRecord(ctx->ev_snap, ctx->stream));
  MMG_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_snap, 0));
  if (c) {
    // one collective for the three result vectors: [3][count] -> [world][3][count]
    ncclResult_t r = ncclAllGather(pack, gathered, (size_t)3 * count, ncclDouble, c->comm, ctx->stream2);
    if (r != ncclSuccess) return set_err(ctx, MMG_E_LIB, std::string("ncclAllGather: ") + ncclGetErrorString(r));
  }
  for (int k = 0; k < 3; ++k) {
    if (!dsts[k]) continue;
    for (int w = 0; w < world; ++w)
      MMG_HIP(ctx, hipMemcpyAsync(dsts[k] + (size_t)w * count, gathered + ((size_t)w * 3 + k) * count,
                                  count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream2));
  }
  MMG_HIP(ctx, hipEventRecord(ctx->ev_deliver, ctx->stream2));
  ctx->deliver_pending = true;
  return MMG_OK;
}

int mmg_comm_allgather_scan(mmg_ctx* ctx, mmg_comm* c, int64_t count, double* rss, double* F, double* p) {
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, c != nullptr);
  int rc = mmg_scan_deliver_begin(ctx, c, count, rss, F, p);
  if (rc) return rc;
  return mmg_scan_deliver_wait(ctx);
}

extern "C++" {
template <typename T>
static int allreduce_host(mmg_ctx* ctx, mmg_comm* c, T* buf, int64_t count, int op, ncclDataType_t dt) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, c && buf && count >= 0 && op >= 0 && op <= 2);
  if (count == 0) return MMG_OK;
  T* d = nullptr;
  MMG_HIP(ctx, sc.alloc(&d, count * sizeof(T)));
  MMG_HIP(ctx, hipMemcpyAsync(d, buf, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
  ncclResult_t r = ncclAllReduce(d, d, count, dt, ops[op], c->comm, ctx->stream);
  if (r != ncclSuccess) { return set_err(ctx, MMG_E_LIB, std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
  MMG_HIP(ctx, hipMemcpyAsync(buf, d, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

}  // extern C++
int mmg_comm_info(mmg_comm* c, int* rank, int* world, int* nccl_count) {
  if (!c) return MMG_E_ARG;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (nccl_count) {
    int n = 0;
    ncclResult_t r = ncclCommCount(c->comm, &n);
    if (r != ncclSuccess) return set_err(nullptr, MMG_E_LIB, std::string("ncclCommCount: ") + ncclGetErrorString(r));
    *nccl_count = n;
  }
  return MMG_OK;
}

int mmg_comm_allgather_f64(mmg_ctx* ctx, mmg_comm* c, const double* send, int64_t count, double* recv) {
  Scratch sc;
  MMG_ENTER(ctx);
  MMG_CHECK_ARG(ctx, c && send && recv && count >= 0);
  if (count == 0) return MMG_OK;
  double *ds = nullptr, *dr = nullptr;
  MMG_HIP(ctx, sc.alloc(&ds, count * sizeof(double)));
  MMG_HIP(ctx, sc.alloc(&dr, (size_t)c->world * count * sizeof(double)));
  MMG_HIP(ctx, hipMemcpyAsync(ds, send, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  MMG_NCCL(ctx, ncclAllGather(ds, dr, (size_t)count, ncclDouble, c->comm, ctx->stream));
  MMG_HIP(ctx, hipMemcpyAsync(recv, dr, (size_t)c->world * count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MMG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MMG_OK;
}

int mmg_comm_allreduce_f64(mmg_ctx* ctx, mmg_comm* c, double* buf, int64_t count, int op) {
  return allreduce_host<double>(ctx, c, buf, count, op, ncclDouble);
}
int mmg_comm_allreduce_i64(mmg_ctx* ctx, mmg_comm* c, int64_t* buf, int64_t count, int op) {
  return allreduce_host<int64_t>(ctx, c, buf, count, op, ncclInt64);
}
int mmg_comm_barrier(mmg_ctx* ctx, mmg_comm* c) {
  int64_t one = 1;
  return mmg_comm_allreduce_i64(ctx, c, &one, 1, 0);
}

}  // extern "C"
