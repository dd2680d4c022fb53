// Runtime entry points: error string, device info, HIP-graph capture and event timing.
#include <cstring>

#include "mp_common.h"

namespace mp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mp

extern "C" {

const char* mp_last_error(void) { return mp::g_err; }

int mp_version(void) { return 100; }

int mp_device_info(char* name_host, int name_len, int* num_cu_host, int* lds_bytes_host) {
  int dev = 0;
  MP_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MP_HIP(hipGetDeviceProperties(&prop, dev));
  if (name_host && name_len > 0) {
    snprintf(name_host, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (num_cu_host) *num_cu_host = prop.multiProcessorCount;
  if (lds_bytes_host) *lds_bytes_host = static_cast<int>(prop.maxSharedMemoryPerMultiProcessor);
  return MP_OK;
}

int mp_graph_begin(mpStream_t stream) {
  MP_REQUIRE(stream != nullptr, "mp_graph_begin: capture needs a non-default stream");
  MP_HIP(hipStreamBeginCapture(mp::as_stream(stream), hipStreamCaptureModeThreadLocal));
  return MP_OK;
}

int mp_graph_end(mpStream_t stream, void** graph_exec_out_host) {
  MP_REQUIRE(graph_exec_out_host != nullptr, "mp_graph_end: null output");
  hipGraph_t graph = nullptr;
  MP_HIP(hipStreamEndCapture(mp::as_stream(stream), &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    mp::set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return MP_EHIP;
  }
  *graph_exec_out_host = exec;
  return MP_OK;
}

int mp_graph_launch(void* graph_exec, mpStream_t stream) {
  MP_REQUIRE(graph_exec != nullptr, "mp_graph_launch: null graph");
  MP_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), mp::as_stream(stream)));
  return MP_OK;
}

int mp_graph_destroy(void* graph_exec) {
  if (graph_exec) MP_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
  return MP_OK;
}

int mp_event_create(void** event_out_host) {
  MP_REQUIRE(event_out_host != nullptr, "mp_event_create: null output");
  hipEvent_t ev;
  MP_HIP(hipEventCreate(&ev));
  *event_out_host = ev;
  return MP_OK;
}

int mp_event_record(void* event, mpStream_t stream) {
  MP_REQUIRE(event != nullptr, "mp_event_record: null event");
  MP_HIP(hipEventRecord(static_cast<hipEvent_t>(event), mp::as_stream(stream)));
  return MP_OK;
}

int mp_event_elapsed_ms(void* start, void* stop, float* ms_out_host) {
  MP_REQUIRE(start && stop && ms_out_host, "mp_event_elapsed_ms: null argument");
  MP_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  MP_HIP(hipEventElapsedTime(ms_out_host, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return MP_OK;
}

int mp_event_destroy(void* event) {
  if (event) MP_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return MP_OK;
}

}  // extern "C"
