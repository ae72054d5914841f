// Error plumbing + build identification for the C ABI (include/gangealing_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace gg {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code ? code : -1;
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

// ---- per-(device, stream) scratch -------------------------------------------------------------------------------
// Split reductions (split-K convolutions, K-split weight gradients, grid-wide sums) write their partial results here
// and a finishing pass adds them in a FIXED order: no floating-point atomics on the training path, so every result
// is bitwise reproducible run to run.  Kernels of one stream execute in order, so consecutive launches share the
// buffer.  Buffers only grow (geometrically); outgrown ones stay allocated until gg_scratch_release(), because
// launches already enqueued - or captured into a hipGraph - still reference them.
namespace {
// Where scratch memory comes from.  Default: hipMalloc / hipFree.  A host framework that runs its own caching
// allocator installs it here (gg_set_allocator; gangealing_amd/_lib.py installs torch's): the library's buffers then
// live inside the framework's pool - visible to its accounting, satisfiable after it has reserved most of HBM (it
// releases cached blocks and retries), and legal while a stream is being captured into a hipGraph (torch serves
// capture-time allocations from the graph's private pool).
gg_alloc_fn g_alloc = nullptr;
gg_free_fn g_free = nullptr;

void* dev_alloc(size_t bytes, hipError_t* err) {
  *err = hipSuccess;
  if (g_alloc) {
    void* p = g_alloc((long long)bytes);
    if (!p) *err = hipErrorOutOfMemory;
    return p;
  }
  void* p = nullptr;
  *err = hipMalloc(&p, bytes);
  if (*err != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
  return p;
}
void dev_free(void* p) {
  if (!p) return;
  if (g_free) g_free(p);
  else (void)hipFree(p);
}

struct ScratchEntry {
  void* buf = nullptr;
  size_t bytes = 0;
  unsigned* tickets = nullptr;
  std::vector<void*> retired;
};
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, ScratchEntry> g_scratch;

ScratchEntry* scratch_entry(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  return &g_scratch[std::make_pair(dev, st)];
}
}  // namespace

void* scratch(hipStream_t st, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  ScratchEntry* e = scratch_entry(st);
  if (!e) { fail(-3, "scratch: no current device"); return nullptr; }
  if (e->bytes >= bytes && e->buf) return e->buf;
  size_t want = e->bytes * 2;
  if (want < bytes) want = bytes;
  const size_t gran = (size_t)1 << 24;                       // 16 MiB granules
  want = (want + gran - 1) / gran * gran;
  hipError_t err;
  void* p = dev_alloc(want, &err);                            // (plain hipMalloc is not legal during a capture: see
  if (!p) {                                                   //  gg_set_allocator / gg_scratch_reserve)
    fail((int)err, "scratch: allocation of %zu bytes failed: %s (during a hipGraph capture without an installed "
         "allocator?  run the step eagerly once before capturing, or call gg_scratch_reserve)", want,
         hipGetErrorString(err));
    return nullptr;
  }
  if (e->buf) e->retired.push_back(e->buf);
  e->buf = p;
  e->bytes = want;
  return p;
}

unsigned* tickets(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  ScratchEntry* e = scratch_entry(st);
  if (!e) { fail(-3, "tickets: no current device"); return nullptr; }
  if (e->tickets) return e->tickets;
  // The page must be ZERO before the first kernel that takes a ticket runs.  Inside a hipGraph capture a memset on the
  // capturing stream is only recorded, not executed: an eager launch before the first replay would count arrivals on
  // uninitialised memory (and a replay would clear a page that eager kernels rely on).  So the first use of a stream's
  // ticket page may not happen inside a capture - only the scratch buffer may grow there (gg_set_allocator).
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
  if (cap != hipStreamCaptureStatusNone) {
    fail(-4, "tickets: the ticket page of this stream would be created inside a hipGraph capture; call "
         "gg_scratch_reserve(0, stream) (or run the step once eagerly) on this stream before capturing");
    return nullptr;
  }
  hipError_t err;
  void* p = dev_alloc(sizeof(unsigned) * kTickets, &err);
  if (!p) {
    (void)hipGetLastError();
    fail((int)err, "tickets: allocation failed: %s", hipGetErrorString(err));
    return nullptr;
  }
  // cleared on the stream whose kernels will use them (stream-ordered: executed before any later launch on it)
  err = hipMemsetAsync(p, 0, sizeof(unsigned) * kTickets, st);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    dev_free(p);
    fail((int)err, "tickets: clearing the ticket page failed: %s", hipGetErrorString(err));
    return nullptr;
  }
  e->tickets = reinterpret_cast<unsigned*>(p);
  return e->tickets;
}

}  // namespace gg

extern "C" int gg_scratch_reserve(long long bytes, void* stream) {
  if (bytes < 0) return gg::fail(-2, "scratch_reserve: negative size");
  hipStream_t st = gg::as_stream(stream);
  if (!gg::tickets(st)) return -3;
  if (bytes > 0 && !gg::scratch(st, (size_t)bytes)) return -3;
  return 0;
}

extern "C" int gg_scratch_release(void) {
  std::lock_guard<std::mutex> lock(gg::g_scratch_mu);
  if (hipDeviceSynchronize() != hipSuccess) return gg::fail(-3, "scratch_release: device synchronisation failed");
  for (auto& kv : gg::g_scratch) {
    for (void* p : kv.second.retired) gg::dev_free(p);
    gg::dev_free(kv.second.buf);
    gg::dev_free(kv.second.tickets);
  }
  gg::g_scratch.clear();
  return 0;
}

extern "C" int gg_set_allocator(gg_alloc_fn alloc, gg_free_fn free_fn) {
  std::lock_guard<std::mutex> lock(gg::g_scratch_mu);
  if ((alloc == nullptr) != (free_fn == nullptr)) return gg::fail(-2, "set_allocator: both functions or neither");
  if (!gg::g_scratch.empty())
    return gg::fail(-2, "set_allocator: scratch buffers exist already (install the allocator first, or call "
                        "gg_scratch_release)");
  gg::g_alloc = alloc;
  gg::g_free = free_fn;
  return 0;
}

extern "C" int gg_abi_version(void) { return 6; }
extern "C" const char* gg_last_error(void) { return gg::g_err; }
extern "C" const char* gg_build_arch(void) { return "gfx950"; }
