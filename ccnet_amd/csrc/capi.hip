// Error state, the plan recorder and small queries of libccnet_hip.so.
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "common.h"

struct ccnet_plan {
    std::vector<ccnet_launch_fn> ops;
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

int ccnet_submit(ccnet_launch_fn fn, ccnet_stream stream) {
    if (g_recording) {
        g_recording->ops.push_back(std::move(fn));
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
        int rc = op((hipStream_t)stream, x_dev);
        if (rc != CCNET_OK) return rc;
    }
    return CCNET_OK;
}

extern "C" void ccnet_plan_destroy(ccnet_plan* plan) {
    if (plan == g_recording) g_recording = nullptr;
    delete plan;
}
