// guard.hip -- diagnostic build only (make GUARD=1 -> ../lib/libmixmogam_hip_guard.so; select it with MMG_LIB=<path>).
// There is no GPU address sanitizer on this pool (xnack+ code objects are refused), so the guard build gives every device
// buffer the library allocates GUARD_BYTES of 0xA5 on both sides and checks them when the buffer is freed and whenever
// mmg_guard_check() is called: a kernel (or copy) that WRITES past either end of a buffer is named by the file:line of
// the allocation it ran over.  Reads past an end land in the padding instead of a neighbour's pages.  Never part of the
// shipped library: mmg_internal.h routes hipMalloc / hipFree here only under -DMMG_GUARD.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
constexpr size_t GUARD_BYTES = 256 << 10;
struct Rec { size_t bytes; const char* file; int line; };
std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;        // user pointer -> record
long g_bad = 0;

// first / last damaged byte of a guard region (host copy), -1 if intact
static bool damaged(const std::vector<unsigned char>& h, size_t* first, size_t* last) {
  bool any = false;
  for (size_t i = 0; i < h.size(); ++i)
    if (h[i] != 0xA5) { if (!any) *first = i; *last = i; any = true; }
  return any;
}

static int check_one(void* user, const Rec& r, const char* when) {
  std::vector<unsigned char> lo(GUARD_BYTES), hi(GUARD_BYTES);
  unsigned char* base = (unsigned char*)user - GUARD_BYTES;
  (void)hipDeviceSynchronize();
  if (hipMemcpy(lo.data(), base, GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (hipMemcpy(hi.data(), base + GUARD_BYTES + r.bytes, GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  int bad = 0;
  size_t a = 0, b = 0;
  if (damaged(lo, &a, &b)) {
    fprintf(stderr, "[mmg guard] %s: bytes %zu..%zu BEFORE the %zu-byte buffer of %s:%d were written\n", when,
            GUARD_BYTES - b, GUARD_BYTES - a, r.bytes, r.file, r.line);
    ++bad;
  }
  if (damaged(hi, &a, &b)) {
    fprintf(stderr, "[mmg guard] %s: bytes %zu..%zu PAST the end of the %zu-byte buffer of %s:%d were written\n", when, a, b,
            r.bytes, r.file, r.line);
    ++bad;
  }
  if (bad) {   // report an overrun once: restore the pattern
    (void)hipMemset(base, 0xA5, GUARD_BYTES);
    (void)hipMemset(base + GUARD_BYTES + r.bytes, 0xA5, GUARD_BYTES);
  }
  return bad;
}
}  // namespace

hipError_t mmg_guard_malloc(void** p, size_t bytes, const char* file, int line) {
  void* raw = nullptr;
  hipError_t e = hipMalloc(&raw, bytes + 2 * GUARD_BYTES);
  if (e != hipSuccess) { *p = nullptr; return e; }
  (void)hipMemset(raw, 0xA5, GUARD_BYTES);
  (void)hipMemset((unsigned char*)raw + GUARD_BYTES + bytes, 0xA5, GUARD_BYTES);
  (void)hipDeviceSynchronize();
  *p = (unsigned char*)raw + GUARD_BYTES;
  std::lock_guard<std::mutex> lk(g_mu);
  g_live[*p] = Rec{bytes, file, line};
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
  const int bad = check_one(p, r, "at hipFree");
  if (bad) { std::lock_guard<std::mutex> lk(g_mu); g_bad += bad; }
  return hipFree((unsigned char*)p - GUARD_BYTES);
}

// Checks every live buffer; returns the number of damaged guards found since the library was loaded.
extern "C" long mmg_guard_check(void) {
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
extern "C" long mmg_guard_live(void) { std::lock_guard<std::mutex> lk(g_mu); return (long)g_live.size(); }

// Proof that the bands work: one byte written past a 100-byte buffer and one before it must be reported (returns 2).
extern "C" long mmg_guard_selftest(void) {
  void* p = nullptr;
  if (mmg_guard_malloc(&p, 100, "guard.hip(selftest)", 0) != hipSuccess) return -1;
  long before;
  { std::lock_guard<std::mutex> lk(g_mu); before = g_bad; }
  (void)hipMemset((unsigned char*)p + 100, 0, 1);
  (void)hipMemset((unsigned char*)p - 1, 0, 1);
  (void)mmg_guard_free(p);
  std::lock_guard<std::mutex> lk(g_mu);
  const long found = g_bad - before;
  g_bad = before;                       // the self-test's own damage does not count
  return found;
}
