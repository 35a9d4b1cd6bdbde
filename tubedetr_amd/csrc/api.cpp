// Error plumbing of the C ABI (thread-local last-error string, launch checks).
#include <stdarg.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "td_common.h"

namespace td {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return TD_ERR_LAUNCH;
  }
  return TD_OK;
}

// ---- launch timing for bench.py's roofline leg ----
// Off by default (one relaxed atomic load per launch).  When on, records are appended under a mutex and the record a
// launching thread is filling is remembered per thread, so autograd worker threads may launch concurrently.
struct ProfRec { int family, dtype; double flops; hipEvent_t a, b; int M, N, K, R, stride, mode; double bytes = 0; };
static std::atomic<bool> g_prof{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static thread_local long t_cur = -1;
static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
bool prof_on() { return g_prof.load(std::memory_order_relaxed); }
void prof_begin(int family, int dtype, double flops, hipStream_t st, int M, int N, int K, int R, int stride, int mode) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{family, dtype, flops, get_event(), get_event(), M, N, K, R, stride, mode};
  hipEventRecord(r.a, st);
  t_cur = (long)g_recs.size();
  g_recs.push_back(r);
}
void prof_end(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (t_cur >= 0 && t_cur < (long)g_recs.size()) hipEventRecord(g_recs[t_cur].b, st);
  t_cur = -1;
}
void prof_set_bytes(double bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (t_cur >= 0 && t_cur < (long)g_recs.size()) g_recs[t_cur].bytes = bytes;
}

// the one mode switch of the library: -1 = not yet seeded from the environment
static std::atomic<int> g_det{-1};
bool deterministic() {
  int v = g_det.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("TD_DETERMINISTIC");
    v = (e && e[0] == '1') ? 1 : 0;
    int expected = -1;
    g_det.compare_exchange_strong(expected, v, std::memory_order_relaxed);
    v = g_det.load(std::memory_order_relaxed);
  }
  return v == 1;
}

}  // namespace td

extern "C" int td_set_deterministic(int on) {
  td::g_det.store(on ? 1 : 0, std::memory_order_relaxed);
  return TD_OK;
}
extern "C" int td_get_deterministic(void) { return td::deterministic() ? 1 : 0; }

extern "C" int td_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(td::g_prof_mu);
  td::g_prof.store(on != 0);
  if (on) {
    for (auto& r : td::g_recs) { td::g_pool.push_back(r.a); td::g_pool.push_back(r.b); }
    td::g_recs.clear();
  }
  return TD_OK;
}
extern "C" int td_prof_collect(int family, int dtype, long long* launches, double* ms, double* flops) {
  long long n = 0; double t = 0, f = 0;
  std::lock_guard<std::mutex> lk(td::g_prof_mu);
  for (auto& r : td::g_recs) {
    if (r.family != family || r.dtype != dtype) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) { td::set_error("td_prof_collect: event sync failed"); return TD_ERR_LAUNCH; }
    float e = 0.f;
    hipEventElapsedTime(&e, r.a, r.b);
    n++; t += e; f += r.flops;
  }
  if (launches) *launches = n;
  if (ms) *ms = t;
  if (flops) *flops = f;
  return TD_OK;
}

extern "C" int td_prof_collect_bytes(int family, int dtype, double* bytes) {
  double b = 0;
  std::lock_guard<std::mutex> lk(td::g_prof_mu);
  for (auto& r : td::g_recs)
    if (r.family == family && r.dtype == dtype) b += r.bytes;
  if (bytes) *bytes = b;
  return TD_OK;
}

extern "C" int td_prof_dump(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) { td::set_error("td_prof_dump: cannot open %s", path); return TD_ERR_INVALID; }
  fprintf(f, "family,dtype,M,N,K,R,stride,mode,ms\n");
  std::lock_guard<std::mutex> lk(td::g_prof_mu);
  for (auto& r : td::g_recs) {
    hipEventSynchronize(r.b);
    float e = 0.f;
    hipEventElapsedTime(&e, r.a, r.b);
    fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%.5f\n", r.family, r.dtype, r.M, r.N, r.K, r.R, r.stride, r.mode, e);
  }
  fclose(f);
  return TD_OK;
}

extern "C" const char* td_last_error(void) { return td::g_err; }
extern "C" int td_abi_version(void) { return 9; }  // 9 (round 6): td_pw_chain2, td_set_deterministic / td_get_deterministic
