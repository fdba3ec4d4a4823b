// guard.hip -- diagnostic build only (make GUARD=1 -> ../lib/libmixmogam_hip_guard.so; select it with MMG_LIB=<path>).
// There is no GPU address sanitizer on this pool (xnack+ code objects are refused), so the guard build routes every device
// allocation of the library through here (mmg_internal.h, only under -DMMG_GUARD).  Three modes, MMG_GUARD_MODE=
//   bands (default)  GUARD_BYTES of 0xA5 on both sides of every buffer, checked when the buffer is freed and whenever
//                    mmg_guard_check() is called: a kernel (or copy) that WRITES past either end is named by the file:line of the
//                    allocation it ran over.  Freed buffers are filled with 0xA5 and held back from hipFree in a quarantine
//                    (MMG_GUARD_QUARANTINE_MB, default 1024): a use-after-free READ returns poison that a checked result shows.
//   fence            (round 5) every buffer is a virtual-memory mapping of its own whose END sits on the last mapped byte
//                    (rounded up to MMG_GUARD_ALIGN, default 256) with unmapped address space behind it, and a freed buffer is
//                    unmapped while its addresses stay reserved for the life of the process: a READ or write past the end, or
//                    of a freed buffer, is a GPU memory fault at that access instead of a silent neighbour read.  The slack in
//                    front of the buffer (same mapping) is 0xA5 and checked at free; the buffer itself starts as 0xA5 too, so
//                    reads of never-written memory are poison, not the zeros a fresh page happens to hold.
//   fence_left       the mirror image: the buffer starts on the first mapped byte, unmapped space in front (underruns fault).
// On SIGABRT (what the runtime raises after a GPU memory fault) the handler prints the last entry points the library was
// called through (MMG_GUARD_TRACE=1: every one as it happens) and writes the table of live and freed buffers to the file
// MMG_GUARD_DUMP names (default stderr), so that the faulting address the runtime printed can be matched to the file:line of
// the allocation it lies behind, in front of, or in.  Never part of the shipped library.
#include <hip/hip_runtime.h>
#include <signal.h>
#include <stdint.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

// A HIP call of the guard's own whose failure does not change what the guard does next: say so, and take the error off the
// runtime's sticky "last error" -- the library reads that after its launches (MMG_HIP(ctx, hipGetLastError())), and a failure
// in here must not surface there as the library's.
#define GUARD_IGN(call)                                                                                            \
  do {                                                                                                             \
    hipError_t e__ = (call);                                                                                       \
    if (e__ != hipSuccess) {                                                                                       \
      fprintf(stderr, "[mmg guard] %s -> %s (guard.hip:%d, ignored)\n", #call, hipGetErrorString(e__), __LINE__);  \
      (void)hipGetLastError();                                                                                     \
    }                                                                                                              \
  } while (0)

namespace {
constexpr size_t GUARD_BYTES = 256 << 10;
enum Mode { BANDS = 0, FENCE = 1, FENCE_LEFT = 2 };
struct Rec {
  size_t bytes; const char* file; int line;
  // fence modes: the reservation [va, va + span) and the mapping [map, map + map_bytes) inside it
  unsigned char* va; size_t span; unsigned char* map; size_t map_bytes; hipMemGenericAllocationHandle_t h;
};
struct Freed { void* user; Rec r; };
std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;        // user pointer -> record
std::deque<Freed> g_quarantine;               // band buffers: poisoned, not yet hipFree'd
std::deque<Freed> g_fenced_freed;             // fenced buffers: unmapped for good (table only; never evicted).  One deque for both
                                              // was a bug of its own in the fence modes: a freed 3 GB band buffer (above 1 GiB every
                                              // buffer gets bands) pushed the quarantine over its cap, the eviction loop popped FENCE
                                              // records and treated them as band buffers -- hipMemcpy from an unmapped address, hipFree
                                              // of user - 256 KiB --, and the two hipErrorInvalidValue stayed in the runtime's last-error
                                              // slot until the library read it behind its next launch ("the launch was refused":
                                              // profiles/r5e_guard_va_recycling.txt; found with AMD_LOG_LEVEL=3)
size_t g_quarantine_bytes = 0;
long g_bad = 0;

Mode mode() {
  static const Mode m = [] {
    const char* e = std::getenv("MMG_GUARD_MODE");
    const Mode v = !e ? BANDS : std::string(e) == "fence" ? FENCE : std::string(e) == "fence_left" ? FENCE_LEFT : BANDS;
    fprintf(stderr, "[mmg guard] mode: %s\n", v == BANDS ? "bands" : v == FENCE ? "fence" : "fence_left");
    return v;
  }();
  return m;
}
size_t fence_align() {
  static const size_t a = [] { const char* e = std::getenv("MMG_GUARD_ALIGN"); const long v = e ? std::atol(e) : 0; return (size_t)(v >= 1 ? v : 256); }();
  return a;
}
size_t quarantine_cap() {
  static const size_t c = [] { const char* e = std::getenv("MMG_GUARD_QUARANTINE_MB"); const long v = e ? std::atol(e) : -1; return (size_t)(v >= 0 ? v : 1024) << 20; }();
  return c;
}

// ---- the last entry points the library was called through (mmg_guard_note, from MMG_ENTER)
constexpr int RING = 384;
struct Note { const char* fn; unsigned long tid; };
Note g_ring[RING];
std::atomic<unsigned long> g_ring_n{0};

void dump_record(FILE* f, const char* state, void* user, const Rec& r) {
  fprintf(f, "[mmg guard]   %-6s %p .. %p  (%zu bytes)  %s:%d", state, user, (void*)((unsigned char*)user + r.bytes), r.bytes, r.file, r.line);
  if (r.va) fprintf(f, "  mapped %p .. %p", (void*)r.map, (void*)(r.map + r.map_bytes));
  fputc('\n', f);
}

void on_abort(int) {
  // diagnostics only: not async-signal-safe, the process is going down anyway
  const unsigned long n = g_ring_n.load();
  fprintf(stderr, "[mmg guard] SIGABRT -- last entry points (oldest first):\n");
  for (unsigned long i = n > RING ? n - RING : 0; i < n; ++i)
    fprintf(stderr, "[mmg guard]   #%lu thread %lx %s\n", i, g_ring[i % RING].tid, g_ring[i % RING].fn);
  const char* dump = std::getenv("MMG_GUARD_DUMP");
  FILE* f = dump ? fopen(dump, "w") : stderr;
  if (!f) f = stderr;
  if (g_mu.try_lock()) {
    fprintf(f, "[mmg guard] %zu live buffers, %zu freed ones on record:\n", g_live.size(), g_quarantine.size() + g_fenced_freed.size());
    for (auto& kv : g_live) dump_record(f, "live", kv.first, kv.second);
    for (auto& q : g_quarantine) dump_record(f, "freed", q.user, q.r);
    for (auto& q : g_fenced_freed) dump_record(f, "freed", q.user, q.r);
    g_mu.unlock();
  }
  if (f != stderr) fclose(f);
  signal(SIGABRT, SIG_DFL);
  abort();
}
void install_handler() {
  static const bool once = [] { signal(SIGABRT, on_abort); return true; }();
  (void)once;
}

// first / last damaged byte of a guard region (host copy), -1 if intact
bool damaged(const std::vector<unsigned char>& h, size_t* first, size_t* last) {
  bool any = false;
  for (size_t i = 0; i < h.size(); ++i)
    if (h[i] != 0xA5) { if (!any) *first = i; *last = i; any = true; }
  return any;
}

int check_region(const unsigned char* p, size_t n, const char* when, const char* side, const Rec& r, bool before) {
  if (!n) return 0;
  std::vector<unsigned char> h(n);
  if (hipMemcpy(h.data(), p, n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  size_t a = 0, b = 0;
  if (!damaged(h, &a, &b)) return 0;
  if (before) fprintf(stderr, "[mmg guard] %s: bytes %zu..%zu BEFORE the %zu-byte buffer of %s:%d were written\n", when, n - b, n - a, r.bytes, r.file, r.line);
  else fprintf(stderr, "[mmg guard] %s: bytes %zu..%zu %s of the %zu-byte buffer of %s:%d were written\n", when, a, b, side, r.bytes, r.file, r.line);
  GUARD_IGN(hipMemset((void*)p, 0xA5, n));   // report an overrun once: restore the pattern
  return 1;
}

int check_one(void* user, const Rec& r, const char* when) {
  GUARD_IGN(hipDeviceSynchronize());
  unsigned char* u = (unsigned char*)user;
  if (!r.va)
    return check_region(u - GUARD_BYTES, GUARD_BYTES, when, "", r, true) + check_region(u + r.bytes, GUARD_BYTES, when, "PAST the end", r, false);
  // fence modes: the slack of the mapping on the side that is not fenced (at most 256 KiB of it)
  const size_t lo = std::min<size_t>((size_t)(u - r.map), GUARD_BYTES);
  const size_t hi = std::min<size_t>((size_t)(r.map + r.map_bytes - (u + r.bytes)), GUARD_BYTES);
  return check_region(u - lo, lo, when, "", r, true) + check_region(u + r.bytes, hi, when, "PAST the end", r, false);
}

extern "C" void mmg_guard_note(const char* fn);
__global__ void guard_fill_kernel(unsigned long long* p, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0xA5A5A5A5A5A5A5A5ull;
}

hipError_t fence_malloc(void** p, size_t bytes, Rec& r) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess) return e;
  if (gran < 4096) gran = 4096;
  const size_t al = fence_align();
  const size_t padded = (bytes + al - 1) / al * al;
  const size_t map_bytes = (std::max<size_t>(padded, 1) + gran - 1) / gran * gran;
  // unmapped address space on either side (MMG_GUARD_VA_MB, default 64 MiB): an access that far off still lands in THIS
  // buffer's reservation and not in a neighbour's, so the table names the right allocation
  static const size_t va_guard = [] { const char* e = std::getenv("MMG_GUARD_VA_MB"); const long v = e ? std::atol(e) : 0; return (size_t)(v > 0 ? v : 64) << 20; }();
  const size_t vg = (va_guard + gran - 1) / gran * gran;
  const size_t span = map_bytes + 2 * vg;
  void* va = nullptr;
  e = hipMemAddressReserve(&va, span, gran, nullptr, 0);
  if (e != hipSuccess) return e;
  hipMemGenericAllocationHandle_t h;
  e = hipMemCreate(&h, map_bytes, &prop, 0);
  if (e != hipSuccess) { GUARD_IGN(hipMemAddressFree(va, span)); return e; }
  unsigned char* map = (unsigned char*)va + vg;
  e = hipMemMap(map, map_bytes, 0, h, 0);
  if (e != hipSuccess) { GUARD_IGN(hipMemRelease(h)); GUARD_IGN(hipMemAddressFree(va, span)); return e; }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(map, map_bytes, &acc, 1);
  if (e != hipSuccess) { GUARD_IGN(hipMemUnmap(map, map_bytes)); GUARD_IGN(hipMemRelease(h)); GUARD_IGN(hipMemAddressFree(va, span)); return e; }
  static const bool verbose = std::getenv("MMG_GUARD_TRACE") != nullptr;
  if (verbose) fprintf(stderr, "[mmg guard] fence_malloc %zu bytes: mapping %p .. %p, filling\n", bytes, (void*)map, (void*)(map + map_bytes));
  if (bytes >= ((size_t)1 << 30)) mmg_guard_note("(fence_malloc: filling a buffer >= 1 GiB)");
  guard_fill_kernel<<<4096, 256>>>((unsigned long long*)map, map_bytes / 8);      // (map_bytes is a multiple of 4096)
  hipError_t ef = hipGetLastError();
  hipError_t es = hipDeviceSynchronize();
  if (bytes >= ((size_t)1 << 30)) mmg_guard_note("(fence_malloc: filled)");
  if (verbose) fprintf(stderr, "[mmg guard] fence_malloc filled (%d, %d)\n", (int)ef, (int)es);
  r.va = (unsigned char*)va; r.span = span; r.map = map; r.map_bytes = map_bytes; r.h = h;
  *p = mode() == FENCE ? map + (map_bytes - padded) : map;
  return hipSuccess;
}
}  // namespace

hipError_t mmg_guard_malloc(void** p, size_t bytes, const char* file, int line) {
  install_handler();
  Rec r{bytes, file, line, nullptr, 0, nullptr, 0, {}};
  // ROCm 7.2: the first kernel to touch a 17.7 GB mapping made after tens of GB were unmapped and released faulted INSIDE that
  // mapping, at the first page past a 1 GiB boundary of the address space -- in this file's own fill kernel, 4 runs of 4 of
  // tests/test_gpu_bign.py (profiles/r5_guard_fence_vmm_fault.log), with hipMemset as with a plain grid-stride store, with and
  // without a pause after hipMemSetAccess.  Buffers beyond MMG_GUARD_FENCE_MAX_MB (default 1024) therefore get guard bands
  // around a hipMalloc instead of a mapping of their own; the kernels that index them run fenced at every smaller size.
  static const size_t fence_max = [] { const char* e = std::getenv("MMG_GUARD_FENCE_MAX_MB"); const long v = e ? std::atol(e) : 0; return (size_t)(v > 0 ? v : 1024) << 20; }();
  // Address space is never recycled.  Round 5: once freed reservations were handed back (hipMemAddressFree of the oldest beyond
  // 20,000) two threads in the library saw wrong results and damaged slack within a few rounds -- at round 2,490 with that cap, at
  // round 340 with a cap of 5,000, never (4,745 rounds) with 60,000 (profiles/r5e_guard_va_recycling.txt).  No access faulted before
  // the recycling began, so these are not stale pointers of the library meeting new mappings; what the runtime does with a recycled
  // range is its own.
  // Nor should reservations pile up for ever: each leaves mappings in the HOST's address space (a reserved range with a mapped
  // stretch inside it is three; vm.max_map_count is 65,530 by default).  A precaution, never reached in any run so far (the
  // two-thread stress stays under 60 % of the limit through 40,000 fenced buffers): the guard watches /proc/self/maps, and beyond
  // 60 % of the limit (or MMG_GUARD_MAX_RESERVATIONS fenced buffers, if set) the rest of the process gets guard bands.
  static const long max_reservations = [] { const char* e = std::getenv("MMG_GUARD_MAX_RESERVATIONS"); const long v = e ? std::atol(e) : 0; return v > 0 ? v : 0L; }();
  static std::atomic<long> reservations{0};
  static std::atomic<bool> exhausted{false};
  bool fenced = mode() != BANDS && bytes <= fence_max && !exhausted.load();
  if (fenced) {
    const long n = reservations.fetch_add(1);
    bool stop = max_reservations > 0 && n >= max_reservations;
    if (!stop && (n & 127) == 0) {
      static const long limit = [] { long v = 65530; if (FILE* f = fopen("/proc/sys/vm/max_map_count", "r")) { if (fscanf(f, "%ld", &v) != 1) v = 65530; fclose(f); } return v; }();
      long lines = 0;
      if (FILE* f = fopen("/proc/self/maps", "r")) {
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0)
          for (size_t i = 0; i < got; ++i) lines += buf[i] == '\n';
        fclose(f);
      }
      stop = lines > limit * 6 / 10;
      if (stop) fprintf(stderr, "[mmg guard] %ld fenced buffers so far, %ld of %ld host mappings in use: later buffers get guard bands\n", n, lines, limit);
    }
    if (stop) { exhausted.store(true); fenced = false; }
  }
  if (fenced) {
    hipError_t e = fence_malloc(p, bytes, r);
    if (e != hipSuccess) { *p = nullptr; return e == hipErrorOutOfMemory ? e : hipErrorOutOfMemory; }
  } else {
    void* raw = nullptr;
    hipError_t e = hipMalloc(&raw, bytes + 2 * GUARD_BYTES);
    if (e != hipSuccess) {                                      // the quarantine holds memory the library has given back
      std::vector<Freed> q;
      { std::lock_guard<std::mutex> lk(g_mu); q.assign(g_quarantine.begin(), g_quarantine.end()); g_quarantine.clear(); g_quarantine_bytes = 0; }
      (void)hipGetLastError();
      for (auto& f : q) GUARD_IGN(hipFree((unsigned char*)f.user - GUARD_BYTES));
      e = hipMalloc(&raw, bytes + 2 * GUARD_BYTES);
    }
    if (e != hipSuccess) { *p = nullptr; return e; }
    GUARD_IGN(hipMemset(raw, 0xA5, GUARD_BYTES));
    GUARD_IGN(hipMemset((unsigned char*)raw + GUARD_BYTES + bytes, 0xA5, GUARD_BYTES));
    GUARD_IGN(hipDeviceSynchronize());
    *p = (unsigned char*)raw + GUARD_BYTES;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_live[*p] = r;
  return hipSuccess;
}

hipError_t mmg_guard_free(void* p) {
  if (!p) return hipSuccess;
  Rec r{};
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) {
      fprintf(stderr, "[mmg guard] hipFree of %p, which the library did not allocate (or freed before)\n", p);
      ++g_bad;
      return hipErrorInvalidValue;
    }
    r = it->second;
    g_live.erase(it);
  }
  const int bad = check_one(p, r, "at hipFree");   // (synchronises the device, as hipFree does)
  if (bad) { std::lock_guard<std::mutex> lk(g_mu); g_bad += bad; }
  if (r.va) {
    // unmapped and released, the addresses stay reserved: whatever still reads or writes the buffer faults from here on
    GUARD_IGN(hipMemUnmap(r.map, r.map_bytes));
    GUARD_IGN(hipMemRelease(r.h));
    std::lock_guard<std::mutex> lk(g_mu);
    g_fenced_freed.push_back(Freed{p, r});
    return hipSuccess;                                        // (the reservation is never given back: see max_reservations)
  }
  GUARD_IGN(hipMemset(p, 0xA5, r.bytes));             // poison: a later read through a stale pointer does not see plausible data
  GUARD_IGN(hipDeviceSynchronize());
  std::vector<Freed> out;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_quarantine.push_back(Freed{p, r});
    g_quarantine_bytes += r.bytes + 2 * GUARD_BYTES;
    while (g_quarantine_bytes > quarantine_cap() && !g_quarantine.empty()) {
      out.push_back(g_quarantine.front());
      g_quarantine_bytes -= g_quarantine.front().r.bytes + 2 * GUARD_BYTES;
      g_quarantine.pop_front();
    }
  }
  hipError_t e = hipSuccess;
  for (auto& f : out) {
    // a write into a freed buffer while it sat in the quarantine shows as damaged poison
    std::vector<unsigned char> h(std::min<size_t>(f.r.bytes, 1 << 20));
    size_t a = 0, b = 0;
    if (!h.empty() && hipMemcpy(h.data(), f.user, h.size(), hipMemcpyDeviceToHost) == hipSuccess && damaged(h, &a, &b)) {
      fprintf(stderr, "[mmg guard] bytes %zu..%zu of the FREED %zu-byte buffer of %s:%d were written after hipFree\n", a, b, f.r.bytes, f.r.file, f.r.line);
      std::lock_guard<std::mutex> lk(g_mu);
      ++g_bad;
    }
    hipError_t e1 = hipFree((unsigned char*)f.user - GUARD_BYTES);
    if (e1 != hipSuccess) { e = e1; fprintf(stderr, "[mmg guard] hipFree of a quarantined buffer -> %s\n", hipGetErrorString(e1)); (void)hipGetLastError(); }
  }
  return e;
}

// Checks every live buffer; returns the number of damaged guards found since the library was loaded.
extern "C" __attribute__((visibility("default"))) long mmg_guard_check(void) {
  std::vector<std::pair<void*, Rec>> live;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    live.assign(g_live.begin(), g_live.end());
  }
  long bad = 0;
  for (auto& kv : live) bad += check_one(kv.first, kv.second, "mmg_guard_check");
  std::lock_guard<std::mutex> lk(g_mu);
  g_bad += bad;
  return g_bad;
}
extern "C" __attribute__((visibility("default"))) long mmg_guard_live(void) { std::lock_guard<std::mutex> lk(g_mu); return (long)g_live.size(); }
extern "C" __attribute__((visibility("default"))) int mmg_guard_mode(void) { return (int)mode(); }

// entry-point breadcrumbs (MMG_ENTER): the abort handler prints the last ones; MMG_GUARD_TRACE=1 prints every one as it happens
extern "C" __attribute__((visibility("default"))) void mmg_guard_note(const char* fn) {
  static const bool live_trace = std::getenv("MMG_GUARD_TRACE") != nullptr;
  const unsigned long i = g_ring_n.fetch_add(1);
  g_ring[i % RING] = Note{fn, (unsigned long)pthread_self()};
  if (live_trace) fprintf(stderr, "[mmg] %lx %s\n", (unsigned long)pthread_self(), fn);
}

hipError_t mmg_guard_func_attr(const char* what, int line, const void* f, hipFuncAttribute a, int v) {
  const hipError_t e = hipFuncSetAttribute(f, a, v);
  if (e != hipSuccess) fprintf(stderr, "[mmg guard] hipFuncSetAttribute(%s) at line %d -> %s\n", what, line, hipGetErrorString(e));
  return e;
}

void mmg_guard_launch_check(const char* kernel, dim3 grid, dim3 block, size_t lds) {
  const hipError_t e = hipPeekAtLastError();
  if (e != hipSuccess)
    fprintf(stderr, "[mmg guard] the launch of %s (grid %u x %u x %u, block %u x %u x %u, %zu bytes of dynamic LDS) left the runtime's last error at: %s\n",
            kernel, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds, hipGetErrorString(e));
}

extern "C" __attribute__((visibility("default"))) void mmg_guard_launched(hipStream_t s) {
  static const bool sync = [] { const char* e = std::getenv("MMG_GUARD_SYNC"); return e && e[0] == '1'; }();
  if (sync) GUARD_IGN(hipStreamSynchronize(s));
}

// Proof that the bands work: one byte written past a 100-byte buffer and one before it must be reported (returns 2; 1 in the
// fence modes for the side whose neighbour byte is unmapped: fence_left always, fence when MMG_GUARD_ALIGN leaves no padding).
extern "C" __attribute__((visibility("default"))) long mmg_guard_selftest(void) {
  void* p = nullptr;
  if (mmg_guard_malloc(&p, 100, "guard.hip(selftest)", 0) != hipSuccess) return -1;
  long before;
  { std::lock_guard<std::mutex> lk(g_mu); before = g_bad; }
  if (mode() != FENCE || fence_align() >= 128) GUARD_IGN(hipMemset((unsigned char*)p + 100, 0, 1));   // (fence: inside the alignment padding)
  if (mode() != FENCE_LEFT) GUARD_IGN(hipMemset((unsigned char*)p - 1, 0, 1));
  (void)mmg_guard_free(p);
  std::lock_guard<std::mutex> lk(g_mu);
  const long found = g_bad - before;
  g_bad = before;                       // the self-test's own damage does not count
  return found;
}

// Proof that the fence works (fence modes only; ENDS THE PROCESS with a GPU memory fault when it does): a kernel reads one
// byte past the end (fence) / before the start (fence_left) of a 1000-byte buffer, or, which = 1, a byte of a freed buffer.
__global__ void guard_probe_kernel(const unsigned char* p, unsigned* out) { out[0] = p[0]; }
extern "C" __attribute__((visibility("default"))) long mmg_guard_fault_selftest(int which) {
  if (mode() == BANDS) return -1;
  void *p = nullptr, *o = nullptr;
  if (mmg_guard_malloc(&p, 1024, "guard.hip(fault selftest)", 0) != hipSuccess) return -2;
  if (mmg_guard_malloc(&o, 256, "guard.hip(fault selftest out)", 0) != hipSuccess) return -2;
  const unsigned char* target = mode() == FENCE ? (unsigned char*)p + 1024 : (unsigned char*)p - 1;
  if (which == 1) { target = (unsigned char*)p; (void)mmg_guard_free(p); }
  fprintf(stderr, "[mmg guard] fault self-test: reading %p\n", (const void*)target);
  guard_probe_kernel<<<1, 1>>>(target, (unsigned*)o);
  GUARD_IGN(hipDeviceSynchronize());
  return 0;                             // reached only if the access did NOT fault
}

// tools/guard_big_alloc.py: the guarded allocator by itself (does a fenced mapping of 20 GB work on this runtime?)
extern "C" __attribute__((visibility("default"))) int mmg_guard_test_malloc(void** p, size_t bytes) {
  return (int)mmg_guard_malloc(p, bytes, "guard.hip(test malloc)", 0);
}
extern "C" __attribute__((visibility("default"))) int mmg_guard_test_free(void* p) { return (int)mmg_guard_free(p); }
