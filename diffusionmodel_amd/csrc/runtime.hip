// Library-level helpers of libdm_amd.so: error reporting and version.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dm_last_error(void) { return g_err; }
extern "C" int dm_version(void) { return 100; }

float* dm_g_ws = nullptr;
int64_t dm_g_ws_bytes = 0;
int* dm_g_counters = nullptr;        // DM_WS_COUNTERS arrival counters (zero between launches) at the tail of the workspace
extern "C" int dm_set_workspace(void* ws, int64_t bytes) {
    DM_CHECK_ARG((ws == nullptr) == (bytes == 0) && bytes >= 0 && ((uintptr_t)ws & 15) == 0, "dm_set_workspace: need a 16-byte aligned buffer and its size (or NULL, 0)");
    DM_CHECK_ARG(ws == nullptr || bytes >= (int64_t)(1 << 20), "dm_set_workspace: the workspace must hold at least 1 MiB");
    dm_g_ws = (float*)ws;
    dm_g_ws_bytes = ws ? bytes - DM_WS_COUNTERS * (int64_t)sizeof(int) : 0;
    dm_g_counters = ws ? (int*)((char*)ws + dm_g_ws_bytes) : nullptr;
    if (ws) {          // the split-K kernels count arrivals per output tile here; the last arriver resets its counter
        hipError_t e = hipMemset(dm_g_counters, 0, DM_WS_COUNTERS * sizeof(int));
        if (e != hipSuccess) { dm_set_error("dm_set_workspace: hipMemset failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    return DM_OK;
}

// ------------------------------------------------------------------------------------------------
// dm_debug_poison_lds: one workgroup per CU fills all 160 KiB of that CU's LDS with `pattern` (test infrastructure for the residue
// audit: LDS survives a kernel boundary, so a kernel that reads bytes it did not write computes on whatever ran before it — its own
// benign residue when a process has the GPU to itself, a stranger's data when it does not.  With NaN bit patterns in every word
// beforehand, such a read turns an exact-integer result into NaN.)  A workgroup that holds 160 KiB has a CU to itself; every
// workgroup waits ~20 us after its fill, so the `n_cu` workgroups of the launch are resident together, i.e. on `n_cu` different CUs.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(uint32_t pattern) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_all[i] = pattern;
    __syncthreads();
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);       // ~ 200 x 127 x 64 cycles / 2.4 GHz = 0.7 ms upper bound; s_sleep is a hint
}
}  // namespace

extern "C" int dm_debug_poison_lds(uint32_t pattern, dm_stream_t stream) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        hipError_t e = hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { ncu = 0; dm_set_error("dm_debug_poison_lds: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(ncu), dim3(256), 160 * 1024, (hipStream_t)stream, pattern);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { dm_set_error("dm_debug_poison_lds: %s", hipGetErrorString(e)); return (int)e; }
    return DM_OK;
}

// ------------------------------------------------------------------------------------------------
// Launch plans (dm_plan_*): replay of a captured step as PLAIN stream launches issued from C.
//
// The train step is ~800 launches; issued from Python one by one the host needs 12-15 ms per step, and a replayed hipGraph
// pays a per-node cost on the device (+0.7 ms at 817 nodes) and cannot contain the data-parallel all-reduce.  A plan keeps
// what a graph is good at — the step is recorded once, memory comes from the capture's private pool, no Python per launch —
// and drops the graph executor: the nodes of the CAPTURED (never instantiated) hipGraph are read back (function, grid, block,
// argument block of every kernel node; memset nodes) in dependency order and re-issued with hipLaunchKernel on an ordinary
// stream.  dm_plan_marker() launches inside the capture split the plan into segments, so the host can do work between them
// (RCCL all-reduce of a gradient bucket) without leaving the replay.
// ------------------------------------------------------------------------------------------------
#include <algorithm>
#include <string>
#include <vector>

namespace {
__global__ void plan_marker_kernel(int id) { (void)id; }

struct PlanOp {
    int kind;                       // 0 kernel, 1 memset, 2 marker
    hipKernelNodeParams k;
    hipMemsetParams ms;
    int marker_id;
};
struct DmPlan {
    std::vector<PlanOp> ops;
    std::vector<int> seg_begin;     // op index where segment s starts (segment s ends at the next marker)
    std::vector<int> seg_marker;    // marker id that ENDS segment s (-1 for the last one)
    int n_kernel = 0, n_memset = 0, n_marker = 0, n_skipped = 0;
    std::vector<hipEvent_t> timed_events;   // dm_plan_run_timed: (start, stop) pairs not read yet
    std::vector<int> timed_ops;
};
}  // namespace

extern "C" int dm_plan_marker(int id, dm_stream_t stream) {
    hipLaunchKernelGGL(plan_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, id);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_plan_from_graph(void* hip_graph, void** plan_out) {
    DM_CHECK_ARG(hip_graph && plan_out, "dm_plan_from_graph: null argument");
    hipGraph_t g = (hipGraph_t)hip_graph;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(g, nullptr, &n);
    DM_CHECK_ARG(e == hipSuccess && n > 0, "dm_plan_from_graph: hipGraphGetNodes failed or empty graph (%s)", hipGetErrorString(e));
    std::vector<hipGraphNode_t> nodes(n);
    e = hipGraphGetNodes(g, nodes.data(), &n);
    DM_CHECK_ARG(e == hipSuccess, "dm_plan_from_graph: hipGraphGetNodes: %s", hipGetErrorString(e));
    // dependency order (Kahn, lowest node index first: a single-stream capture is a chain, so the order is the capture order)
    size_t ne = 0;
    e = hipGraphGetEdges(g, nullptr, nullptr, &ne);
    DM_CHECK_ARG(e == hipSuccess, "dm_plan_from_graph: hipGraphGetEdges: %s", hipGetErrorString(e));
    std::vector<hipGraphNode_t> from(ne), to(ne);
    if (ne) {
        e = hipGraphGetEdges(g, from.data(), to.data(), &ne);
        DM_CHECK_ARG(e == hipSuccess, "dm_plan_from_graph: hipGraphGetEdges: %s", hipGetErrorString(e));
    }
    std::vector<std::pair<hipGraphNode_t, int>> index(n);
    for (size_t i = 0; i < n; ++i) index[i] = {nodes[i], (int)i};
    std::sort(index.begin(), index.end());
    auto idx_of = [&](hipGraphNode_t nd) {
        auto it = std::lower_bound(index.begin(), index.end(), std::make_pair(nd, -1));
        return (it != index.end() && it->first == nd) ? it->second : -1;
    };
    std::vector<std::vector<int>> succ(n);
    std::vector<int> indeg(n, 0);
    for (size_t i = 0; i < ne; ++i) {
        const int a = idx_of(from[i]), b = idx_of(to[i]);
        DM_CHECK_ARG(a >= 0 && b >= 0, "dm_plan_from_graph: edge to an unknown node");
        succ[a].push_back(b);
        ++indeg[b];
    }
    std::vector<int> order, ready;
    for (size_t i = 0; i < n; ++i) if (!indeg[i]) ready.push_back((int)i);
    std::make_heap(ready.begin(), ready.end(), std::greater<int>());
    while (!ready.empty()) {
        std::pop_heap(ready.begin(), ready.end(), std::greater<int>());
        const int u = ready.back();
        ready.pop_back();
        order.push_back(u);
        for (int v : succ[u]) if (--indeg[v] == 0) { ready.push_back(v); std::push_heap(ready.begin(), ready.end(), std::greater<int>()); }
    }
    DM_CHECK_ARG(order.size() == n, "dm_plan_from_graph: the captured graph has a cycle?");
    DmPlan* p = new DmPlan();
    p->seg_begin.push_back(0);
    for (int u : order) {
        hipGraphNodeType ty;
        e = hipGraphNodeGetType(nodes[u], &ty);
        if (e != hipSuccess) { delete p; dm_set_error("dm_plan_from_graph: hipGraphNodeGetType: %s", hipGetErrorString(e)); return DM_EINVAL; }
        PlanOp op{};
        if (ty == hipGraphNodeTypeKernel) {
            e = hipGraphKernelNodeGetParams(nodes[u], &op.k);
            // the replay launches with hipLaunchKernel(func, ..., kernelParams, ...): a node whose arguments live in `extra`
            // (hipModuleLaunchKernel-style) would be launched with a null argument block -> refused, loudly
            if (e != hipSuccess || op.k.func == nullptr || op.k.kernelParams == nullptr) {
                delete p;
                dm_set_error("dm_plan_from_graph: kernel node %d has no kernelParams array (%s%s): not replayable", u, hipGetErrorString(e),
                             (e == hipSuccess && op.k.extra != nullptr) ? "; its arguments are in `extra`" : "");
                return DM_EUNSUPPORTED;
            }
            if (op.k.func == (void*)plan_marker_kernel) {
                op.kind = 2;
                op.marker_id = *(int*)op.k.kernelParams[0];
                p->seg_marker.push_back(op.marker_id);
                p->ops.push_back(op);
                p->seg_begin.push_back((int)p->ops.size());
                ++p->n_marker;
                continue;
            }
            op.kind = 0;
            ++p->n_kernel;
        } else if (ty == hipGraphNodeTypeMemset) {
            e = hipGraphMemsetNodeGetParams(nodes[u], &op.ms);
            if (e != hipSuccess || op.ms.height > 1) { delete p; dm_set_error("dm_plan_from_graph: unsupported memset node (%s)", hipGetErrorString(e)); return DM_EUNSUPPORTED; }
            op.kind = 1;
            ++p->n_memset;
        } else if (ty == hipGraphNodeTypeEmpty || ty == hipGraphNodeTypeEventRecord || ty == hipGraphNodeTypeWaitEvent) {
            ++p->n_skipped;          // ordering-only nodes: the replay is serial on one stream
            continue;
        } else {
            delete p;
            dm_set_error("dm_plan_from_graph: node %d has type %d (memcpy / host / child-graph nodes are not replayable: route copies "
                         "through a library kernel such as dm_cast or dm_scatter_copy)", u, (int)ty);
            return DM_EUNSUPPORTED;
        }
        p->ops.push_back(op);
    }
    p->seg_marker.push_back(-1);
    *plan_out = p;
    return DM_OK;
}

/* info[0..5] = ops, kernels, memsets, markers, skipped ordering nodes, segments */
extern "C" int dm_plan_info(void* plan, int32_t* info) {
    DM_CHECK_ARG(plan && info, "dm_plan_info: null argument");
    DmPlan* p = (DmPlan*)plan;
    info[0] = (int)p->ops.size(); info[1] = p->n_kernel; info[2] = p->n_memset; info[3] = p->n_marker; info[4] = p->n_skipped;
    info[5] = (int)p->seg_begin.size();
    return DM_OK;
}

/* marker id that ends segment `seg` (-1: the last segment, or out of range) */
extern "C" int dm_plan_segment_marker(void* plan, int seg) {
    DmPlan* p = (DmPlan*)plan;
    return (p && seg >= 0 && seg < (int)p->seg_marker.size()) ? p->seg_marker[seg] : -1;
}

/* name of the kernel of op `idx` (empty for memsets / markers); returns the op kind or -1 */
extern "C" int dm_plan_op_name(void* plan, int idx, char* buf, int cap) {
    DmPlan* p = (DmPlan*)plan;
    if (!p || idx < 0 || idx >= (int)p->ops.size() || !buf || cap < 1) return -1;
    buf[0] = 0;
    const PlanOp& op = p->ops[idx];
    if (op.kind == 0) {
        const char* nm = hipKernelNameRefByPtr(op.k.func, nullptr);
        snprintf(buf, cap, "%s", nm ? nm : "?");
    } else if (op.kind == 1) {
        snprintf(buf, cap, "memset %zu bytes", (size_t)op.ms.width * op.ms.elementSize);
    } else {
        snprintf(buf, cap, "marker %d", op.marker_id);
    }
    return op.kind;
}

static int plan_issue(const PlanOp& op, hipStream_t st, int idx = -1) {
    hipError_t e = hipSuccess;
    if (op.kind == 0) {
        e = hipLaunchKernel(op.k.func, op.k.gridDim, op.k.blockDim, op.k.kernelParams, op.k.sharedMemBytes, st);
    } else if (op.kind == 1) {
        if (op.ms.elementSize == 4) e = hipMemsetD32Async((hipDeviceptr_t)op.ms.dst, (int)op.ms.value, op.ms.width, st);
        else if (op.ms.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)op.ms.dst, (unsigned short)op.ms.value, op.ms.width, st);
        else e = hipMemsetAsync(op.ms.dst, (int)op.ms.value, op.ms.width, st);
    }
    if (e != hipSuccess) {
        // (the launches before op `idx` have been issued: the step is half-way — the caller must not keep training on this state)
        dm_set_error("dm_plan_run: op %d (%s) failed: %s; ops before it were issued", idx,
                     op.kind == 0 ? hipKernelNameRefByPtr(op.k.func, nullptr) : "memset", hipGetErrorString(e));
        return (int)e;
    }
    return DM_OK;
}

// does the kernel name contain one of the '|'-separated substrings of `list` ("" matches everything)
static bool name_hit(const char* nm, const char* list) {
    if (!nm) return false;
    if (!*list) return true;
    const char* a = list;
    while (*a) {
        const char* b = strchr(a, '|');
        const size_t n = b ? (size_t)(b - a) : strlen(a);
        if (n > 0 && n < 200) {
            char buf[200];
            memcpy(buf, a, n);
            buf[n] = 0;
            if (strstr(nm, buf)) return true;
        }
        if (!b) break;
        a = b + 1;
    }
    return false;
}

/* issue segments [seg_first, seg_last] on `stream` (marker launches themselves are not replayed) */
extern "C" int dm_plan_run(void* plan, int seg_first, int seg_last, dm_stream_t stream) {
    DmPlan* p = (DmPlan*)plan;
    DM_CHECK_ARG(p, "dm_plan_run: null plan");
    const int nseg = (int)p->seg_begin.size();
    DM_CHECK_ARG(seg_first >= 0 && seg_last < nseg && seg_first <= seg_last, "dm_plan_run: segments [%d, %d] outside [0, %d)", seg_first, seg_last, nseg);
    const int lo = p->seg_begin[seg_first];
    const int hi = seg_last + 1 < nseg ? p->seg_begin[seg_last + 1] : (int)p->ops.size();
    hipStream_t st = (hipStream_t)stream;
    for (int i = lo; i < hi; ++i) {
        if (p->ops[i].kind == 2) continue;
        const int rc = plan_issue(p->ops[i], st, i);
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}

/* dm_plan_run with a HIP event pair (on `stream`) around every kernel whose name contains `substr`; the events stay with the
   plan until dm_plan_timed_results reads them (which synchronises the stream): ms_out[i] = duration of the i-th such launch
   since the last read, in issue order. */
extern "C" int dm_plan_run_timed(void* plan, int seg_first, int seg_last, const char* substr, dm_stream_t stream) {
    DmPlan* p = (DmPlan*)plan;
    DM_CHECK_ARG(p && substr, "dm_plan_run_timed: bad argument");
    const int nseg = (int)p->seg_begin.size();
    DM_CHECK_ARG(seg_first >= 0 && seg_last < nseg && seg_first <= seg_last, "dm_plan_run_timed: segments [%d, %d] outside [0, %d)", seg_first, seg_last, nseg);
    const int lo = p->seg_begin[seg_first];
    const int hi = seg_last + 1 < nseg ? p->seg_begin[seg_last + 1] : (int)p->ops.size();
    hipStream_t st = (hipStream_t)stream;
    for (int i = lo; i < hi; ++i) {
        const PlanOp& op = p->ops[i];
        if (op.kind == 2) continue;
        bool hit = false;
        if (op.kind == 0) {
            const char* nm = hipKernelNameRefByPtr(op.k.func, nullptr);
            hit = name_hit(nm, substr);            // '|'-separated list of substrings; "" = every kernel
        }
        hipEvent_t a = nullptr, b = nullptr;
        if (hit) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
        const int rc = plan_issue(op, st, i);
        if (hit) { (void)hipEventRecord(b, st); p->timed_events.push_back(a); p->timed_events.push_back(b); p->timed_ops.push_back(i); }
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}

/* ms_out[i], op_out[i] (op index in the plan) for the launches timed since the last call; returns their count through *n_out */
extern "C" int dm_plan_timed_results(void* plan, float* ms_out, int32_t* op_out, int cap, int32_t* n_out, dm_stream_t stream) {
    DmPlan* p = (DmPlan*)plan;
    DM_CHECK_ARG(p && ms_out && op_out && n_out && cap >= 0, "dm_plan_timed_results: bad argument");
    (void)hipStreamSynchronize((hipStream_t)stream);
    const int n = (int)p->timed_ops.size();
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, p->timed_events[2 * i], p->timed_events[2 * i + 1]);
        if (i < cap) { ms_out[i] = ms; op_out[i] = p->timed_ops[i]; }
        (void)hipEventDestroy(p->timed_events[2 * i]);
        (void)hipEventDestroy(p->timed_events[2 * i + 1]);
    }
    p->timed_events.clear();
    p->timed_ops.clear();
    *n_out = n < cap ? n : cap;
    return DM_OK;
}

extern "C" int dm_plan_destroy(void* plan) {
    delete (DmPlan*)plan;
    return DM_OK;
}
