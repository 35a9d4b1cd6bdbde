// Test infrastructure (not product code): an "electric fence" device allocator for torch's pluggable-allocator hook.
// Every allocation gets its own virtual-memory mapping whose END coincides with the end of the tensor (up to the
// alignment slack) and is followed by an unmapped guard range: a kernel that reads or writes past the end of any
// tensor - the failure mode a caching allocator hides, because its blocks are rounded up and packed into 2 MiB+
// segments - raises a GPU memory fault at the offending access instead of silently touching a neighbour.
// Used by tests/efence/run_efence.py (see DESIGN.md "memory-fault investigation").
//
// Frees are deferred: the VA stays mapped until the next drain (device synchronise, then unmap), so in-flight kernels
// never lose their memory.  Freed ranges are unmapped but their virtual addresses are NEVER handed out again (default,
// TD_EFENCE_KEEP=1), which turns use-after-free into a fault as well.  TD_EFENCE_KEEP=0 returns the address ranges with
// hipMemAddressFree - measured unusable on ROCm 7.0.2: once ranges are re-reserved and re-mapped, kernels observe stale
// contents / NaNs in tensors nobody wrote to and eventually abort with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION (the same
// tests are clean with KEEP=1 and with the caching allocator), i.e. an artefact of VA recycling, not of the code under test.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/types.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Rec {
  void* base;       // reserved VA: [guard | mapping | guard]
  size_t reserved;  // bytes reserved
  size_t mapped;    // bytes mapped (multiple of the granularity)
  hipMemGenericAllocationHandle_t handle;
  bool vmm;
};
std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;
std::vector<Rec> g_dead;
size_t g_dead_bytes = 0, g_gran = 0;
long g_allocs = 0, g_fallbacks = 0;
int g_mode = -1;  // 1 = VMM available, 0 = plain hipMalloc fallback

size_t env_sz(const char* n, size_t dflt) {
  const char* e = getenv(n);
  return e ? (size_t)strtoull(e, nullptr, 10) : dflt;
}

void drain_locked() {
  if (g_dead.empty()) return;
  (void)hipDeviceSynchronize();
  static const bool keep = env_sz("TD_EFENCE_KEEP", 1) != 0;
  for (auto& r : g_dead) {
    if (r.vmm) {
      (void)hipMemUnmap((char*)r.base + g_gran, r.mapped);
      (void)hipMemRelease(r.handle);
      if (!keep) (void)hipMemAddressFree(r.base, r.reserved);
    } else {
      (void)hipFree(r.base);
    }
  }
  g_dead.clear();
  g_dead_bytes = 0;
}
}  // namespace

extern "C" void* td_efence_malloc(ssize_t size, int device, hipStream_t) {
  std::lock_guard<std::mutex> lk(g_mu);
  static const size_t align = env_sz("TD_EFENCE_ALIGN", 64);
  if (size <= 0) size = 1;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  if (g_mode < 0) {
    g_mode = (hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum) == hipSuccess && g_gran > 0) ? 1 : 0;
    fprintf(stderr, "[efence] mode=%s granularity=%zu align=%zu\n", g_mode ? "vmm" : "hipMalloc-fallback (NO PROTECTION)", g_gran, align);
  }
  ++g_allocs;
  const size_t need = ((size_t)size + align - 1) / align * align;
  if (g_mode == 1) {
    Rec r = {};
    r.vmm = true;
    r.mapped = (need + g_gran - 1) / g_gran * g_gran;
    r.reserved = r.mapped + 2 * g_gran;
    hipError_t e = hipMemAddressReserve(&r.base, r.reserved, g_gran, nullptr, 0);
    if (e == hipSuccess) e = hipMemCreate(&r.handle, r.mapped, &prop, 0);
    if (e == hipSuccess) e = hipMemMap((char*)r.base + g_gran, r.mapped, 0, r.handle, 0);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (e == hipSuccess) e = hipMemSetAccess((char*)r.base + g_gran, r.mapped, &acc, 1);
    if (e == hipSuccess) {
      void* p = (char*)r.base + g_gran + r.mapped - need;  // the tensor ends where the mapping ends
      // TD_EFENCE_POISON=1: fresh memory is filled with 0xFF bytes (NaN as fp32 / bf16, -1 as integers), so a kernel that
      // READS memory nobody has written shows up as NaN in its results instead of depending on what the pages held
      static const bool poison = env_sz("TD_EFENCE_POISON", 0) != 0;
      if (poison) {  // (hipMemset may return before the fill has run: wait, or the fill could land on top of a kernel's output)
        (void)hipMemset((char*)r.base + g_gran, 0xFF, r.mapped);
        (void)hipDeviceSynchronize();
      }
      g_live[p] = r;
      return p;
    }
    fprintf(stderr, "[efence] VMM allocation of %zd bytes failed (%s): falling back to hipMalloc (NO PROTECTION)\n", size, hipGetErrorString(e));
    (void)hipGetLastError();
    drain_locked();
    g_mode = 0;
  }
  ++g_fallbacks;
  Rec r = {};
  if (hipMalloc(&r.base, need) != hipSuccess) {
    drain_locked();
    if (hipMalloc(&r.base, need) != hipSuccess) return nullptr;
  }
  g_live[r.base] = r;
  return r.base;
}

extern "C" void td_efence_free(void* ptr, ssize_t, int, hipStream_t) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_live.find(ptr);
  if (it == g_live.end()) return;
  g_dead.push_back(it->second);
  g_dead_bytes += it->second.mapped;
  g_live.erase(it);
  static const size_t lim_bytes = env_sz("TD_EFENCE_DRAIN_GB", 48) << 30, lim_n = env_sz("TD_EFENCE_DRAIN_N", 4096);
  if (g_dead_bytes > lim_bytes || g_dead.size() > lim_n) drain_locked();
}

// 1 = every allocation so far was fenced, 0 = some (or all) fell back to hipMalloc
extern "C" int td_efence_protected(void) { return g_mode == 1 && g_fallbacks == 0; }
extern "C" long td_efence_allocs(void) { return g_allocs; }
