// af_runtime.hip -- error strings, version, device probe and the hipEvent profiling hook.
#include <atomic>
#include <mutex>
#include <vector>

#include "af_common.h"

static thread_local std::string g_last_error;

void af_set_error(const std::string& msg) { g_last_error = msg; }
int af_fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

extern "C" const char* af_last_error(void) { return g_last_error.c_str(); }
extern "C" int af_version(void) { return 100; }

extern "C" int af_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return af_fail(AF_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

static thread_local std::string g_pending_error;      // a refused attribute, reported by the next af_check_launch()

bool af_allow_dyn_lds(const void* kernel, size_t bytes, bool& done, const char* what) {
  if (done || bytes <= 65536) return true;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    g_pending_error = std::string(what) + ": hipFuncSetAttribute(MaxDynamicSharedMemorySize = " + std::to_string(bytes) + "): " + hipGetErrorString(e);
    return false;
  }
  done = true;
  return true;
}

int af_check_launch(const char* what) {
  if (!g_pending_error.empty()) {
    std::string m;
    m.swap(g_pending_error);
    (void)hipGetLastError();
    return af_fail(AF_E_HIP, m);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return af_fail(AF_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
  return AF_OK;
}

// ---- profiling: per-family (start, stop) event pairs, recorded on the launch stream -------
namespace {
struct ProfState {
  std::mutex mu;
  std::atomic<bool> on{false};
  unsigned epoch = 0;      // bumped by af_prof_reset(): scopes opened before a reset do not record their stop event
  struct Pair {
    hipEvent_t a, b;
  };
  std::vector<Pair> pairs[AF_FAM_COUNT];
  std::vector<Pair> pool;
};
ProfState& prof() {
  static ProfState s;
  return s;
}
}  // namespace

AfLaunchScope::AfLaunchScope(int family_, void* stream_) : family(family_), stream((hipStream_t)stream_), ev_stop(nullptr), epoch(0) {
  ProfState& p = prof();
  if (!p.on.load(std::memory_order_relaxed)) return;
  // events cannot be recorded into a stream that is being captured into a hipGraph: skip (the graph replays carry no events)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  std::lock_guard<std::mutex> lk(p.mu);
  ProfState::Pair pr;
  if (!p.pool.empty()) {
    pr = p.pool.back();
    p.pool.pop_back();
  } else {
    if (hipEventCreate(&pr.a) != hipSuccess) return;
    if (hipEventCreate(&pr.b) != hipSuccess) {
      (void)hipEventDestroy(pr.a);
      return;
    }
  }
  (void)hipEventRecord(pr.a, stream);
  p.pairs[family].push_back(pr);
  ev_stop = pr.b;
  epoch = p.epoch;
}

AfLaunchScope::~AfLaunchScope() {
  if (!ev_stop) return;
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  if (epoch != p.epoch) return;      // af_prof_reset() ran meanwhile: the pair went back to the pool, leave it alone
  (void)hipEventRecord(ev_stop, stream);
}

extern "C" int af_prof_enable(int on) {
  ProfState& p = prof();
  p.on.store(on != 0);
  return AF_OK;
}

extern "C" int af_prof_reset(void) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  ++p.epoch;
  for (int f = 0; f < AF_FAM_COUNT; ++f) {
    for (auto& pr : p.pairs[f]) p.pool.push_back(pr);
    p.pairs[f].clear();
  }
  return AF_OK;
}

extern "C" int af_prof_read(int family, int* launches, double* total_ms) {
  if (family < 0 || family >= AF_FAM_COUNT || !launches || !total_ms) return af_fail(AF_E_BADARG, "af_prof_read: bad argument");
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  double tot = 0.0;
  for (auto& pr : p.pairs[family]) {
    hipError_t e = hipEventSynchronize(pr.b);
    if (e != hipSuccess) return af_fail(AF_E_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, pr.a, pr.b);
    if (e != hipSuccess) return af_fail(AF_E_HIP, std::string("hipEventElapsedTime: ") + hipGetErrorString(e));
    tot += ms;
  }
  *launches = (int)p.pairs[family].size();
  *total_ms = tot;
  return AF_OK;
}
