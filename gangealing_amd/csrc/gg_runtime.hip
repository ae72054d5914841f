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
  void* p = nullptr;
  const hipError_t err = hipMalloc(&p, want);                 // (not legal while the stream is being captured: the
  if (err != hipSuccess || !p) {                              //  eager warm-up iterations size the buffer first)
    (void)hipGetLastError();
    fail((int)err, "scratch: hipMalloc(%zu) failed: %s (during a hipGraph capture?  run the step eagerly once "
         "before capturing, or call gg_scratch_reserve)", want, hipGetErrorString(err));
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
  void* p = nullptr;
  hipError_t err = hipMalloc(&p, sizeof(unsigned) * kTickets);
  if (err == hipSuccess) err = hipMemset(p, 0, sizeof(unsigned) * kTickets);
  if (err != hipSuccess || !p) {
    (void)hipGetLastError();
    fail((int)err, "tickets: allocation failed: %s", hipGetErrorString(err));
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
    for (void* p : kv.second.retired) (void)hipFree(p);
    if (kv.second.buf) (void)hipFree(kv.second.buf);
    if (kv.second.tickets) (void)hipFree(kv.second.tickets);
  }
  gg::g_scratch.clear();
  return 0;
}

extern "C" int gg_abi_version(void) { return 2; }
extern "C" const char* gg_last_error(void) { return gg::g_err; }
extern "C" const char* gg_build_arch(void) { return "gfx950"; }
