// Error reporting and version of libsage_hip.so.
#include "common.h"

namespace sage {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// pending probes of this thread: {start, stop, number of contraction launches to let pass first}
struct Probe {
  hipEvent_t start, stop;
  int skip;
};
constexpr int kMaxProbes = 8;
static thread_local Probe g_probe[kMaxProbes];
static thread_local int g_nprobe = 0;
bool add_probe(hipEvent_t start, hipEvent_t stop, int skip) {
  if (g_nprobe >= kMaxProbes) return false;
  g_probe[g_nprobe++] = Probe{start, stop, skip};
  return true;
}
void clear_probes() { g_nprobe = 0; }
bool take_probe(hipEvent_t* start, hipEvent_t* stop) {
  // called once per contraction launch: the probe whose count has run out rides on this launch,
  // every other pending probe has one launch fewer to wait for
  int taken = -1;
  for (int i = 0; i < g_nprobe; ++i) {
    if (g_probe[i].skip == 0 && taken < 0) taken = i;
    else if (g_probe[i].skip > 0) --g_probe[i].skip;
  }
  if (taken < 0) return false;
  *start = g_probe[taken].start;
  *stop = g_probe[taken].stop;
  for (int i = taken + 1; i < g_nprobe; ++i) g_probe[i - 1] = g_probe[i];
  --g_nprobe;
  return true;
}
}  // namespace sage

extern "C" int sage_probe_next_contraction(void* start_event, void* stop_event, int skip) {
  if (start_event == nullptr && stop_event == nullptr) {
    sage::clear_probes();
    return SAGE_OK;
  }
  SAGE_REQUIRE(start_event != nullptr && stop_event != nullptr && skip >= 0, SAGE_EINVAL,
               "sage_probe_next_contraction: pass both events and skip >= 0 (or two NULLs to cancel all)");
  SAGE_REQUIRE(sage::add_probe(static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event), skip),
               SAGE_ELIMIT, "sage_probe_next_contraction: too many pending probes");
  return SAGE_OK;
}

extern "C" int sage_event_record(void* event, void* stream) {
  SAGE_REQUIRE(event != nullptr, SAGE_EINVAL, "sage_event_record: NULL event");
  const hipError_t e = hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream));
  SAGE_REQUIRE(e == hipSuccess, SAGE_ELAUNCH, "sage_event_record: %s", hipGetErrorString(e));
  return SAGE_OK;
}

extern "C" int sage_stream_wait_event(void* stream, void* event) {
  SAGE_REQUIRE(event != nullptr, SAGE_EINVAL, "sage_stream_wait_event: NULL event");
  const hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0);
  SAGE_REQUIRE(e == hipSuccess, SAGE_ELAUNCH, "sage_stream_wait_event: %s", hipGetErrorString(e));
  return SAGE_OK;
}

extern "C" const char* sage_last_error(void) { return sage::g_err; }
extern "C" int sage_abi_version(void) { return 2; }
