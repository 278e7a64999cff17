// Error state, the plan recorder and small queries of libccnet_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

constexpr int MAX_LANES = 4;
struct PlanOp {
    ccnet_launch_fn fn;  // empty for a wait
    int kind;
    double flops, bytes;
    int lane;            // stream the op runs on (0 = the caller's)
    int wait_src;        // >= 0: this op is "lane `lane` waits for everything recorded so far on lane wait_src"
};
// One instantiated HIP graph of the whole plan per (device, image pointer), when graphs are enabled (CCNET_GRAPH=1):
// replaying removes the per-launch host work, but see graphs_enabled() for why eager side lanes are the default.
struct PlanGraph {
    int device;
    const float* x;
    hipGraphExec_t exec;
    unsigned long long last_use;
};
struct ccnet_plan {
    std::vector<PlanOp> ops;
    bool open = true;
    int nlanes = 1;
    std::mutex mu;
    // per device: side streams of lanes 1.. and one event per wait op (created on first use, kept for the plan's life)
    struct DevRes {
        int device;
        hipStream_t side[MAX_LANES - 1];
        std::vector<hipEvent_t> events;
    };
    std::vector<DevRes> res;
    std::vector<PlanGraph> graphs;  // small LRU
    std::vector<std::pair<int, const float*>> seen;  // (device, x) pairs that ran eagerly once (warm-up before capture)
    unsigned long long tick = 0;
    ~ccnet_plan() {
        // a replay or side-lane work of this plan may still be running: let every device that holds one of its graphs or
        // lane streams drain before they are destroyed (destruction is rare; forwards never come here)
        int cur = 0;
        const bool have_cur = hipGetDevice(&cur) == hipSuccess;
        std::vector<int> devs;
        for (auto& g : graphs) devs.push_back(g.device);
        for (auto& r : res) devs.push_back(r.device);
        for (size_t i = 0; i < devs.size(); ++i) {
            bool dup = false;
            for (size_t j = 0; j < i; ++j) dup = dup || devs[j] == devs[i];
            if (!dup && hipSetDevice(devs[i]) == hipSuccess) (void)hipDeviceSynchronize();
        }
        if (have_cur && !devs.empty()) (void)hipSetDevice(cur);
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
        for (auto& r : res) {
            for (int i = 0; i < MAX_LANES - 1; ++i)
                if (r.side[i]) (void)hipStreamDestroy(r.side[i]);
            for (auto e : r.events) (void)hipEventDestroy(e);
        }
    }
};

namespace {
thread_local char g_err[512] = "";
thread_local ccnet_plan* g_recording = nullptr;
thread_local int g_lane = 0;
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
        g_recording->ops.push_back(PlanOp{std::move(fn), kind, flops, bytes, g_lane, -1});
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
    g_lane = 0;
    *plan = g_recording;
    return CCNET_OK;
}

extern "C" int ccnet_plan_end(void) {
    CC_REQUIRE(g_recording != nullptr, "plan_end: no plan is being recorded on this thread");
    g_recording->open = false;
    g_recording = nullptr;
    g_lane = 0;
    return CCNET_OK;
}

extern "C" int ccnet_plan_num_ops(const ccnet_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

extern "C" int ccnet_plan_set_lane(int lane) {
    CC_REQUIRE(g_recording != nullptr, "plan_set_lane: no plan is being recorded on this thread");
    CC_REQUIRE(lane >= 0 && lane < MAX_LANES, "plan_set_lane: lane must be 0..%d", MAX_LANES - 1);
    g_lane = lane;
    if (lane + 1 > g_recording->nlanes) g_recording->nlanes = lane + 1;
    return CCNET_OK;
}

extern "C" int ccnet_plan_wait(int dst_lane, int src_lane) {
    CC_REQUIRE(g_recording != nullptr, "plan_wait: no plan is being recorded on this thread");
    CC_REQUIRE(dst_lane >= 0 && dst_lane < MAX_LANES && src_lane >= 0 && src_lane < MAX_LANES && dst_lane != src_lane, "plan_wait: bad lanes %d <- %d",
               dst_lane, src_lane);
    const int hi = (dst_lane > src_lane ? dst_lane : src_lane) + 1;
    if (hi > g_recording->nlanes) g_recording->nlanes = hi;
    g_recording->ops.push_back(PlanOp{ccnet_launch_fn(), CCNET_OP_OTHER, 0.0, 0.0, dst_lane, src_lane});
    return CCNET_OK;
}

namespace {
bool lanes_enabled() {
    static int v = -1;
    if (v < 0) {
        // side lanes run the recorded independent branches (head levels, the CBAM gates beside the fused Transformer) on
        // their own streams: measured 5.85 vs 6.06 ms per bs64 step when launched eagerly.  CCNET_LANES=0 turns them off.
        const char* e = getenv("CCNET_LANES");
        v = (e && atoi(e) == 0) ? 0 : 1;
    }
    return v == 1;
}
// the caller holds plan->mu
int lane_resources(ccnet_plan* plan, ccnet_plan::DevRes** out) {
    int dev = 0;
    CC_HIP(hipGetDevice(&dev));
    for (auto& r : plan->res)
        if (r.device == dev) { *out = &r; return CCNET_OK; }
    ccnet_plan::DevRes r{};
    r.device = dev;
    for (int i = 0; i + 1 < plan->nlanes; ++i) CC_HIP(hipStreamCreateWithFlags(&r.side[i], hipStreamNonBlocking));
    size_t nwait = 0;
    for (const auto& op : plan->ops) nwait += op.wait_src >= 0 ? 1 : 0;
    r.events.resize(nwait);
    for (auto& e : r.events) CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    plan->res.push_back(r);
    *out = &plan->res.back();
    return CCNET_OK;
}
// Launches every op.  With `lanes`, ops run on their lane's stream and waits become event record / stream wait pairs
// (this also forks and joins a stream capture); without, everything runs in recorded order on `st`.
int run_eager(ccnet_plan* plan, const float* x_dev, hipStream_t st, bool lanes) {
    ccnet_plan::DevRes* r = nullptr;
    if (lanes && plan->nlanes > 1) {
        int rc = lane_resources(plan, &r);
        if (rc != CCNET_OK) return rc;
    }
    size_t wi = 0;
    int rc = CCNET_OK;  // after a failed launch the remaining ops are skipped but the recorded forks / joins are still
                        // emitted, so the side lanes are joined back into `st` (and a capture is closed) whatever happened
    for (const auto& op : plan->ops) {
        if (op.wait_src >= 0) {
            if (r) {
                hipStream_t src = op.wait_src == 0 ? st : r->side[op.wait_src - 1];
                hipStream_t dst = op.lane == 0 ? st : r->side[op.lane - 1];
                if (hipEventRecord(r->events[wi], src) != hipSuccess || hipStreamWaitEvent(dst, r->events[wi], 0) != hipSuccess) {
                    if (rc == CCNET_OK) rc = ccnet_set_error(CCNET_EHIP, "plan_run: event record / stream wait failed");
                }
            }
            ++wi;
            continue;
        }
        if (rc != CCNET_OK) continue;
        hipStream_t s = (r && op.lane > 0) ? r->side[op.lane - 1] : st;
        rc = op.fn(s, x_dev);
    }
    return rc;
}
bool graphs_enabled() {
    static int v = -1;
    if (v < 0) {
        // opt-in (CCNET_GRAPH=1): on ROCm 7.2 a replayed graph runs its parallel branches one after the other and is no
        // faster than eager launches on one stream (6.09 vs 6.06 ms), while eager launches on side lanes do overlap
        const char* e = getenv("CCNET_GRAPH");
        v = (e && atoi(e) != 0) ? 1 : 0;
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
    if (!graphs_enabled()) {
        std::lock_guard<std::mutex> lock(plan->mu);
        return run_eager(plan, x_dev, st, lanes_enabled());
    }
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
        return run_eager(plan, x_dev, st, lanes_enabled());
    }
    // capture on a private stream (the caller's may be the legacy default stream, which cannot be captured)
    hipStream_t cap = nullptr;
    CC_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    int rc = CCNET_OK;
    if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipStreamDestroy(cap);
        (void)hipGetLastError();
        return run_eager(plan, x_dev, st, lanes_enabled());
    }
    rc = run_eager(plan, x_dev, cap, lanes_enabled());
    hipError_t e = hipStreamEndCapture(cap, &graph);
    (void)hipStreamDestroy(cap);
    if (rc != CCNET_OK || e != hipSuccess || graph == nullptr) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return rc != CCNET_OK ? rc : run_eager(plan, x_dev, st, lanes_enabled());
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess || exec == nullptr) {
        (void)hipGetLastError();
        return run_eager(plan, x_dev, st, lanes_enabled());
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
    std::vector<hipEvent_t> ev(n + 1, nullptr);
    hipStream_t st = (hipStream_t)stream;
    int rc = CCNET_OK;
    for (auto& e : ev)
        if (rc == CCNET_OK && hipEventCreate(&e) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipEventCreate failed");
    if (rc == CCNET_OK && hipEventRecord(ev[0], st) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipEventRecord failed");
    for (size_t i = 0; i < n && rc == CCNET_OK; ++i) {
        if (plan->ops[i].wait_src < 0) rc = plan->ops[i].fn(st, x_dev);  // one stream, recorded order: waits are no-ops
        if (hipEventRecord(ev[i + 1], st) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipEventRecord failed");
    }
    if (rc == CCNET_OK && hipStreamSynchronize(st) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipStreamSynchronize failed");
    for (size_t i = 0; i < n && rc == CCNET_OK; ++i)
        if (hipEventElapsedTime(&ms_per_op[i], ev[i], ev[i + 1]) != hipSuccess) rc = ccnet_set_error(CCNET_EHIP, "hipEventElapsedTime failed");
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    return rc;
}

extern "C" void ccnet_plan_destroy(ccnet_plan* plan) {
    if (plan == g_recording) g_recording = nullptr;
    delete plan;
}
