// Error channel of the C-ABI: every entry point returns 0 on success; on failure the message is
// kept per host thread and read back with mmvid_last_error().
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mmvid_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mmvid_last_error() { return g_err; }

// 2: the front-end state buffer grew from one float (the step counter) to four 32-bit words {step, seed lo, seed hi, reserved}
extern "C" int mmvid_abi_version() { return 3; }

// Number of visible HIP devices (0 when none) -- lets the host side fail loudly and early.
extern "C" int mmvid_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------- tuning knobs
namespace {
struct Opt {
    const char* name;
    const char* env;
    int dflt, value;
    bool set;
};
Opt g_opts[MMVID_OPT_COUNT] = {{"graphs", "MMVID_GRAPHS", 0, 0, false}};
}  // namespace

int mmvid_option(int which) {
    Opt& o = g_opts[which];
    if (!o.set) {
        const char* e = getenv(o.env);
        o.value = e ? atoi(e) : o.dflt;
        o.set = true;
    }
    return o.value;
}

extern "C" int mmvid_set_option(const char* name, int value) {
    MMVID_REQUIRE(name, "set_option: null name");
    for (int i = 0; i < MMVID_OPT_COUNT; ++i)
        if (strcmp(name, g_opts[i].name) == 0) {
            g_opts[i].value = value, g_opts[i].set = true;
            return MMVID_OK;
        }
    mmvid_set_error("set_option: unknown option '%s' (graphs)", name);
    return MMVID_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------- profiler
#include <mutex>
#include <vector>

#include "graphs.h"
#include "prof.h"

namespace {
struct Rec {
    hipEvent_t a, b;
    int cls;
    double flops;
};
bool g_prof_on = false;
int g_prof_stride = 1;
int64_t g_prof_seen[PROF_NCLASS];
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
std::mutex g_prof_mu;

hipEvent_t pool_get() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}
}  // namespace

MmvidProfScope::MmvidProfScope(int cls, double flops, hipStream_t stream) : slot(-1), s(stream) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (cls < 0 || cls >= PROF_NCLASS || (g_prof_seen[cls]++ % g_prof_stride) != 0) return;  // sampled launches only
    Rec r;
    r.a = pool_get(), r.b = pool_get(), r.cls = cls, r.flops = flops;
    (void)hipEventRecord(r.a, s);
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}
MmvidProfScope::~MmvidProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_recs[slot].b, s);
}

// Every `stride`-th launch of each class is bracketed by a pair of HIP events (stride 1 = all of them; an event
// pair costs a few microseconds of stream time, so the benchmark samples).
bool mmvid_prof_recording() { return g_prof_on; }

extern "C" int mmvid_prof_begin(int stride) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.clear();
    g_pool_next = 0;
    g_prof_stride = stride < 1 ? 1 : stride;
    for (int c = 0; c < PROF_NCLASS; ++c) g_prof_seen[c] = 0;
    g_prof_on = true;
    return 0;
}

// Pause / resume recording without resetting what was collected (bench.py times a sample of the steps: while
// recording is on, the long launch sequences run directly instead of as graph replays).
extern "C" int mmvid_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

// Stops recording, waits for the device, and returns per class: ms and algorithmic flops summed over the SAMPLED
// launches, how many were sampled, and how many launches the class had in total.
extern "C" int mmvid_prof_end(double* ms, int64_t* sampled, double* flops, int64_t* launches_total, int nclass) {
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    int64_t* launches = sampled;
    for (int c = 0; c < nclass; ++c) {
        ms[c] = 0, launches[c] = 0, flops[c] = 0;
        if (launches_total) launches_total[c] = c < PROF_NCLASS ? g_prof_seen[c] : 0;
    }
    for (const Rec& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        if (r.cls < nclass) ms[r.cls] += t, launches[r.cls] += 1, flops[r.cls] += r.flops;
    }
    g_recs.clear();
    g_pool_next = 0;
    return 0;
}
