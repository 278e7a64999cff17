// Error state, the plan recorder and small queries of libccnet_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

struct PlanOp {
    ccnet_launch_fn fn;
    int kind;
    double flops, bytes;
};
// One instantiated HIP graph of the whole plan per (device, image pointer): a forward is ~200 dependent launches of
// 10-300 us kernels, and replaying them as a graph removes the per-launch host work and most of the dispatch gaps.
struct PlanGraph {
    int device;
    const float* x;
    hipGraphExec_t exec;
    unsigned long long last_use;
};
struct ccnet_plan {
    std::vector<PlanOp> ops;
    bool open = true;
    std::mutex mu;
    std::vector<PlanGraph> graphs;  // small LRU
    std::vector<std::pair<int, const float*>> seen;  // (device, x) pairs that ran eagerly once (warm-up before capture)
    unsigned long long tick = 0;
    ~ccnet_plan() {
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
    }
};

namespace {
thread_local char g_err[512] = "";
thread_local ccnet_plan* g_recording = nullptr;
}  // namespace

int ccnet_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ccnet_submit(ccnet_launch_fn fn, ccnet_stream stream, int kind, double flops, double bytes) {
    if (g_recording) {
        g_recording->ops.push_back(PlanOp{std::move(fn), kind, flops, bytes});
        return CCNET_OK;
    }
    return fn((hipStream_t)stream, nullptr);
}

extern "C" const char* ccnet_last_error(void) { return g_err; }
extern "C" int ccnet_version(void) { return 100; }

extern "C" int ccnet_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ccnet_plan_begin(ccnet_plan** plan) {
    CC_REQUIRE(plan != nullptr, "plan_begin: null output");
    CC_REQUIRE(g_recording == nullptr, "plan_begin: this thread is already recording a plan");
    g_recording = new ccnet_plan();
    *plan = g_recording;
    return CCNET_OK;
}

extern "C" int ccnet_plan_end(void) {
    CC_REQUIRE(g_recording != nullptr, "plan_end: no plan is being recorded on this thread");
    g_recording->open = false;
    g_recording = nullptr;
    return CCNET_OK;
}

extern "C" int ccnet_plan_num_ops(const ccnet_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

namespace {
int run_eager(const ccnet_plan* plan, const float* x_dev, hipStream_t st) {
    for (const auto& op : plan->ops) {
        int rc = op.fn(st, x_dev);
        if (rc != CCNET_OK) return rc;
    }
    return CCNET_OK;
}
bool graphs_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CCNET_NO_GRAPH");
        v = (e && atoi(e) != 0) ? 0 : 1;
    }
    return v == 1;
}
constexpr size_t MAX_GRAPHS = 4;
}  // namespace

extern "C" int ccnet_plan_run(const ccnet_plan* cplan, const float* x_dev, ccnet_stream stream) {
    CC_REQUIRE(cplan != nullptr, "plan_run: null plan");
    CC_REQUIRE(!cplan->open, "plan_run: plan is still being recorded");
    ccnet_plan* plan = const_cast<ccnet_plan*>(cplan);
    hipStream_t st = (hipStream_t)stream;
    if (!graphs_enabled()) return run_eager(plan, x_dev, st);
    int dev = 0;
    CC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(plan->mu);
    ++plan->tick;
    for (auto& g : plan->graphs)
        if (g.device == dev && g.x == x_dev) {
            g.last_use = plan->tick;
            CC_HIP(hipGraphLaunch(g.exec, st));
            return CCNET_OK;
        }
    // first sight of this (device, image pointer): run eagerly (also warms per-kernel caches); second: capture
    bool warm = false;
    for (auto& s : plan->seen) warm = warm || (s.first == dev && s.second == x_dev);
    if (!warm) {
        if (plan->seen.size() >= 64) plan->seen.clear();
        plan->seen.emplace_back(dev, x_dev);
        return run_eager(plan, x_dev, st);
    }
    // capture on a private stream (the caller's may be the legacy default stream, which cannot be captured)
    hipStream_t cap = nullptr;
    CC_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    int rc = CCNET_OK;
    if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipStreamDestroy(cap);
        (void)hipGetLastError();
        return run_eager(plan, x_dev, st);
    }
    rc = run_eager(plan, x_dev, cap);
    hipError_t e = hipStreamEndCapture(cap, &graph);
    (void)hipStreamDestroy(cap);
    if (rc != CCNET_OK || e != hipSuccess || graph == nullptr) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return rc != CCNET_OK ? rc : run_eager(plan, x_dev, st);
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess || exec == nullptr) {
        (void)hipGetLastError();
        return run_eager(plan, x_dev, st);
    }
    if (plan->graphs.size() >= MAX_GRAPHS) {
        size_t victim = 0;
        for (size_t i = 1; i < plan->graphs.size(); ++i)
            if (plan->graphs[i].last_use < plan->graphs[victim].last_use) victim = i;
        (void)hipGraphExecDestroy(plan->graphs[victim].exec);
        plan->graphs.erase(plan->graphs.begin() + victim);
    }
    plan->graphs.push_back(PlanGraph{dev, x_dev, exec, plan->tick});
    CC_HIP(hipGraphLaunch(exec, st));
    return CCNET_OK;
}

extern "C" int ccnet_plan_op_info(const ccnet_plan* plan, int i, int* kind, double* flops, double* bytes) {
    CC_REQUIRE(plan != nullptr && i >= 0 && i < (int)plan->ops.size(), "plan_op_info: bad index");
    if (kind) *kind = plan->ops[i].kind;
    if (flops) *flops = plan->ops[i].flops;
    if (bytes) *bytes = plan->ops[i].bytes;
    return CCNET_OK;
}

extern "C" int ccnet_plan_run_timed(const ccnet_plan* plan, const float* x_dev, ccnet_stream stream, float* ms_per_op) {
    CC_REQUIRE(plan != nullptr && ms_per_op != nullptr, "plan_run_timed: null argument");
    CC_REQUIRE(!plan->open, "plan_run_timed: plan is still being recorded");
    const size_t n = plan->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) CC_HIP(hipEventCreate(&e));
    hipStream_t st = (hipStream_t)stream;
    int rc = CCNET_OK;
    CC_HIP(hipEventRecord(ev[0], st));
    for (size_t i = 0; i < n && rc == CCNET_OK; ++i) {
        rc = plan->ops[i].fn(st, x_dev);
        if (hipEventRecord(ev[i + 1], st) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipEventRecord failed");
    }
    if (rc == CCNET_OK && hipStreamSynchronize(st) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipStreamSynchronize failed");
    for (size_t i = 0; i < n && rc == CCNET_OK; ++i)
        if (hipEventElapsedTime(&ms_per_op[i], ev[i], ev[i + 1]) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipEventElapsedTime failed");
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

extern "C" void ccnet_plan_destroy(ccnet_plan* plan) {
    if (plan == g_recording) g_recording = nullptr;
    delete plan;
}
