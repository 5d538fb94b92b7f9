// Error string, hipGraph capture and event timing for libmelogan_hip.
#include "common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void mg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int mg_version(void) { return 100; }
const char* mg_last_error(void) { return g_err; }

int mg_graph_begin(mg_stream_t stream) {
    MG_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
    return MG_OK;
}
int mg_graph_end(mg_stream_t stream, void** graph_exec_out) {
    MG_CHECK_ARG(graph_exec_out != nullptr, "mg_graph_end: null out pointer");
    hipGraph_t graph = nullptr;
    MG_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) {
        mg_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return MG_EHIP;
    }
    *graph_exec_out = (void*)exec;
    return MG_OK;
}
// The same capture instantiated `n` times (an experiment knob: alternating executables of the step graph measured slower
// than replaying one, melo-gan_amd/ops.py::Graph).
static thread_local int g_last_kernel_nodes = -1;
int mg_graph_last_kernel_nodes(void) { return g_last_kernel_nodes; }

static void count_kernel_nodes(hipGraph_t graph) {
    g_last_kernel_nodes = -1;
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess || n == 0 || n > 65536) return;
    hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
    if (!nodes) return;
    if (hipGraphGetNodes(graph, nodes, &n) == hipSuccess) {
        int k = 0;
        for (size_t i = 0; i < n; ++i) {
            hipGraphNodeType t;
            if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k;
        }
        g_last_kernel_nodes = k;
    }
    free(nodes);
}

int mg_graph_end_n(mg_stream_t stream, void** graph_execs_out, int n) {
    MG_CHECK_ARG(graph_execs_out != nullptr && n >= 1 && n <= 8, "mg_graph_end_n: 1..8 executables");
    hipGraph_t graph = nullptr;
    MG_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    count_kernel_nodes(graph);
    for (int i = 0; i < n; ++i) {
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            for (int j = 0; j < i; ++j) hipGraphExecDestroy((hipGraphExec_t)graph_execs_out[j]);
            hipGraphDestroy(graph);
            mg_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
            return MG_EHIP;
        }
        graph_execs_out[i] = (void*)exec;
    }
    hipGraphDestroy(graph);
    return MG_OK;
}
int mg_graph_launch(void* graph_exec, mg_stream_t stream) {
    MG_CHECK_ARG(graph_exec != nullptr, "mg_graph_launch: null graph");
    MG_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return MG_OK;
}
int mg_graph_destroy(void* graph_exec) {
    if (graph_exec) MG_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return MG_OK;
}

int mg_event_create(void** ev) {
    MG_CHECK_ARG(ev != nullptr, "mg_event_create: null");
    hipEvent_t e;
    MG_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return MG_OK;
}
int mg_event_record(void* ev, mg_stream_t stream) {
    MG_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return MG_OK;
}
int mg_event_elapsed_ms(void* start, void* stop, float* ms) {
    MG_HIP(hipEventSynchronize((hipEvent_t)stop));
    MG_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MG_OK;
}
int mg_event_destroy(void* ev) {
    if (ev) MG_HIP(hipEventDestroy((hipEvent_t)ev));
    return MG_OK;
}

}  // extern "C"
