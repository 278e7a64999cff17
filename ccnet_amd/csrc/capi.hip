// Error state, the plan recorder and small queries of libccnet_hip.so.
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "common.h"

struct PlanOp {
    ccnet_launch_fn fn;
    int kind;
    double flops, bytes;
};
struct ccnet_plan {
    std::vector<PlanOp> ops;
    bool open = true;
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

extern "C" int ccnet_plan_run(const ccnet_plan* plan, const float* x_dev, ccnet_stream stream) {
    CC_REQUIRE(plan != nullptr, "plan_run: null plan");
    CC_REQUIRE(!plan->open, "plan_run: plan is still being recorded");
    for (const auto& op : plan->ops) {
        int rc = op.fn((hipStream_t)stream, x_dev);
        if (rc != CCNET_OK) return rc;
    }
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
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

extern "C" void ccnet_plan_destroy(ccnet_plan* plan) {
    if (plan == g_recording) g_recording = nullptr;
    delete plan;
}
