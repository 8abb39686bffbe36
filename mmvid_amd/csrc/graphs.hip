// See graphs.h.
#include "graphs.h"

#include <mutex>
#include <unordered_map>

#include "../../include/mmvid_hip.h"

namespace {
struct Entry {
    int seen = 0;        // direct runs so far; < 0: capture failed once, never try again
    hipGraphExec_t exec = nullptr;
};
std::mutex g_mu;
std::unordered_map<uint64_t, Entry> g_cache;
hipStream_t g_stream = nullptr;
hipEvent_t g_in = nullptr, g_out = nullptr;
int64_t g_stat[3] = {0, 0, 0};  // direct, captured, replayed
constexpr size_t kMaxEntries = 64;

bool enabled() { return mmvid_option(MMVID_OPT_GRAPHS) == 1; }
bool ensure_stream() {
    if (g_stream) return true;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&g_in, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&g_out, hipEventDisableTiming) != hipSuccess) return false;
    return true;
}
void drop_all() {
    for (auto& kv : g_cache)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    g_cache.clear();
}
}  // namespace

uint64_t mmvid_hash_bytes(const void* p, size_t n, uint64_t h) {  // FNV-1a
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int mmvid_run_cached(uint64_t key, hipStream_t user, const std::function<int(hipStream_t)>& enqueue) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (!enabled() || mmvid_prof_recording() ||
        (user != nullptr && hipStreamIsCapturing(user, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)) {
        ++g_stat[0];
        return enqueue(user);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_cache.size() > kMaxEntries) drop_all();  // addresses keep changing: start over rather than grow
    Entry& e = g_cache[key];
    if (!e.exec && (e.seen < 1 || !ensure_stream())) {
        if (e.seen >= 0) ++e.seen;
        ++g_stat[0];
        return enqueue(user);
    }
    // order the internal stream after everything already queued on the user's stream
    if (hipEventRecord(g_in, user) != hipSuccess || hipStreamWaitEvent(g_stream, g_in, 0) != hipSuccess) {
        mmvid_set_error("graph replay: stream ordering failed");
        return MMVID_ERR_HIP;
    }
    if (!e.exec) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            e.seen = -1;
            ++g_stat[0];
            return enqueue(user);
        }
        const int rc = enqueue(g_stream);
        const hipError_t ce = hipStreamEndCapture(g_stream, &graph);
        if (rc != 0 || ce != hipSuccess || !graph ||
            hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            e.exec = nullptr;
            e.seen = -1;
            if (rc != 0) return rc;  // the sequence itself is invalid: report it (nothing was executed)
            ++g_stat[0];
            return enqueue(user);    // capture is unavailable for this sequence: run it directly from now on
        }
        (void)hipGraphDestroy(graph);
        ++g_stat[1];
    } else {
        ++g_stat[2];
    }
    if (hipGraphLaunch(e.exec, g_stream) != hipSuccess) {
        mmvid_set_error("graph replay: hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError()));
        return MMVID_ERR_HIP;
    }
    if (hipEventRecord(g_out, g_stream) != hipSuccess || hipStreamWaitEvent(user, g_out, 0) != hipSuccess) {
        mmvid_set_error("graph replay: stream ordering failed");
        return MMVID_ERR_HIP;
    }
    return MMVID_OK;
}

// counts[0..2] = sequences run directly / captured / replayed since load (diagnostics, tests)
extern "C" int mmvid_graph_stats(int64_t* counts) {
    MMVID_REQUIRE(counts, "graph_stats: null pointer");
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < 3; ++i) counts[i] = g_stat[i];
    return MMVID_OK;
}
