// hipGraph replay of the long fixed launch sequences of the path (VQGAN encode/decode op list, tower forward,
// tower backward).  A sequence is identified by a 64-bit key over everything its launches bake in (shapes, every
// device pointer, the user's stream).  First sighting of a key: the launches go straight to the user's stream.
// Second sighting: the same enqueue code runs under stream capture on an internal stream and the instantiated
// graph is cached.  From then on a call is one hipGraphLaunch, bracketed by events so that it is ordered on the
// user's stream exactly like the direct launches would be.  For callers of the C-ABI that cannot capture a graph
// themselves (the Python training loop captures the WHOLE step instead: engine.GraphedStep).  Opt-in:
// option "graphs" = 1 (MMVID_GRAPHS=1, mmvid_set_option); measured on a healthy host it neither gains nor loses (25.6 vs 24.8
// ms/step), it only takes ~3 ms of launch work per step off the host.  Bypassed while the HIP-event profiler is
// recording (events cannot be read from replays) and while the calling stream is itself being captured.
#pragma once
#include <functional>

#include "common.h"

uint64_t mmvid_hash_bytes(const void* p, size_t n, uint64_t h);
inline uint64_t mmvid_hash_ptr(const void* p, uint64_t h) { return mmvid_hash_bytes(&p, sizeof(p), h); }
// enqueue(stream) must only enqueue work on `stream` (kernel launches, async D2D copies) and return 0 on success.
int mmvid_run_cached(uint64_t key, hipStream_t user_stream, const std::function<int(hipStream_t)>& enqueue);
bool mmvid_prof_recording();
