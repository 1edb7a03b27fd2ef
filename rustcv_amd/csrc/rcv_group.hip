// rcv_group.hip -- device group: one context per GPU of a node and the frame partition rule.
//
// north_star: "Frame batches shard embarrassingly across the 8 GPUs of one node (independent per-GPU HIP streams, no RCCL
// collective required)".  The reference has nothing to replace here (one process, one device: rustcv/src/internal/runtime.rs:13);
// its host would hold one `rcv_group` next to its `Mat`s.  Every batch entry point of this library only enqueues work on its
// context's stream, so ONE host thread drives all devices: for rank i, call the op on rcv_group_ctx(g, i) with the frames
// rcv_shard_range(n, i, G) of the batch that live on device i, then rcv_group_sync(g) once.  Nothing is exchanged between
// devices; results stay where they were produced.  (Contexts are independent: a host may just as well give each one its own
// thread -- rustcv_amd/multigpu.py does -- or its own process -- bench.py under torch.distributed.run.)
#include "rcv_internal.h"
#include <new>
#include <vector>

struct rcv_group {
    std::vector<rcv_ctx*> ctxs;
};

// frames [floor(rank * n / world), floor((rank + 1) * n / world)) of a batch of n: contiguous, sizes differ by at most one,
// every frame in exactly one range (the rule of rustcv_amd/shard.py and of bench.py's ranks)
extern "C" int rcv_shard_range(int64_t n_frames, int rank, int world, int64_t* first, int64_t* last)
{
    if (!first || !last || world < 1 || rank < 0 || rank >= world || n_frames < 0) return RCV_ERR_ARG;
    *first = (int64_t)((__int128)n_frames * rank / world);
    *last = (int64_t)((__int128)n_frames * (rank + 1) / world);
    return RCV_OK;
}

// devices == nullptr: GPUs 0 .. n_devices - 1.  An ordinal may appear more than once (two contexts = two streams on one GPU).
extern "C" int rcv_group_create(const int* devices, int n_devices, rcv_group** out)
{
    if (!out) return RCV_ERR_ARG;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 1024) return RCV_ERR_ARG;
    rcv_group* g = new (std::nothrow) rcv_group();
    if (!g) return RCV_ERR_OOM;
    for (int i = 0; i < n_devices; ++i) {
        rcv_ctx* c = nullptr;
        const int rc = rcv_ctx_create(devices ? devices[i] : i, &c);
        if (rc != RCV_OK) {   // (a node with fewer GPUs than asked for: RCV_ERR_DEVICE, nothing left behind)
            rcv_group_destroy(g);
            return rc;
        }
        g->ctxs.push_back(c);
    }
    *out = g;
    return RCV_OK;
}

extern "C" void rcv_group_destroy(rcv_group* g)
{
    if (!g) return;
    for (rcv_ctx* c : g->ctxs) rcv_ctx_destroy(c);
    delete g;
}

extern "C" int rcv_group_size(const rcv_group* g) { return g ? (int)g->ctxs.size() : RCV_ERR_ARG; }

extern "C" rcv_ctx* rcv_group_ctx(rcv_group* g, int rank) { return g && rank >= 0 && rank < (int)g->ctxs.size() ? g->ctxs[rank] : nullptr; }

// waits for the stream of every context (all of them, also after a failure); returns the first error
extern "C" int rcv_group_sync(rcv_group* g)
{
    if (!g) return RCV_ERR_ARG;
    int first = RCV_OK;
    for (rcv_ctx* c : g->ctxs) {
        const int rc = rcv_sync(c);
        if (first == RCV_OK && rc != RCV_OK) first = rc;
    }
    return first;
}

// Stream timing across the group (bench.py: two batches in flight on two contexts of ONE device overlap, so no single stream's
// events bracket the work).  start: one event on every context's stream, in rank order; stop: one event on every stream, then
// wait for all of them.  Elapsed = the latest stop against the earliest-completed start event of its device (the maximum over all
// start / stop pairs of the device) -- events of one device share a clock, events of different devices do not, so a multi-device
// group reports the maximum over its devices of each device's own span.  The timer uses each context's ev0 / ev1: a per-context
// rcv_timer_start / rcv_timer_stop between the group's start and stop would overwrite them (do not mix the two).
extern "C" int rcv_group_timer_start(rcv_group* g)
{
    if (!g) return RCV_ERR_ARG;
    for (rcv_ctx* c : g->ctxs) RCV_TRY(rcv_timer_start(c));
    return RCV_OK;
}

extern "C" int rcv_group_timer_stop(rcv_group* g, float* elapsed_ms)
{
    if (!g || !elapsed_ms) return RCV_ERR_ARG;
    for (rcv_ctx* c : g->ctxs) {
        RCV_TRY(rcv_bind(c));
        RCV_HIP(hipEventRecord(c->ev1, c->stream));
    }
    // first wait for EVERY stop event (a stream's stop event implies its start event): hipEventElapsedTime on an event that has not
    // completed returns hipErrorNotReady -- a stream that was still busy at timer_start completes its start event late
    for (rcv_ctx* c : g->ctxs) {
        RCV_TRY(rcv_bind(c));
        RCV_HIP(hipEventSynchronize(c->ev1));
    }
    float best = 0.0f;
    for (rcv_ctx* c : g->ctxs) {
        RCV_TRY(rcv_bind(c));
        // against EVERY start event of the device: the earliest-COMPLETED one gives the longest span (the rank-order first context
        // need not be the earliest)
        for (rcv_ctx* o : g->ctxs) {
            if (o->device != c->device) continue;
            float ms = 0.0f;
            RCV_HIP(hipEventElapsedTime(&ms, o->ev0, c->ev1));
            if (ms > best) best = ms;
        }
    }
    *elapsed_ms = best;
    return RCV_OK;
}
