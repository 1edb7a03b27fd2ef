// rcv_graph.hip -- record a sequence of rcv_*_batch calls once, replay it as ONE hipGraph launch.
//
// The small configurations of the path are launch-bound, not bandwidth-bound: the reference's own loop (config 0,
// examples/camera_demo.rs:50-76: YUYV->BGR on a 640x480 frame, then imgproc::rectangle) is two ~5 us launches for
// 1.5 MB of traffic, and a single 1080p GaussianBlur is ~8 us.  Between rcv_graph_begin and rcv_graph_end every entry
// point that only enqueues work on the context stream (all *_batch calls, and the single-Mat calls on RCV_DEVICE
// mats) is captured instead of executed; rcv_graph_launch replays the whole chain with one submission.
//
// What a graph owns: per-call constant tables (the banded weight operand of the MFMA filter) are copied into device
// buffers of the graph at record time, so a replay never depends on what later calls leave in the context's shared
// constant cache, and every workspace an op carves while it is being recorded (the intermediates of the unfused
// fallbacks) is a fresh allocation of the graph -- never the context's grow-only workspace, which a later, larger call
// would free under the graph's feet.  What it does not own: the image buffers -- replays read and write the same device
// pointers, the caller refreshes their contents between launches.  Entry points that must synchronise (host-Mat staging,
// rcv_sync, rcv_free, rcv_upload/rcv_download, timers, the staging ring) return RCV_ERR_UNSUPPORTED while recording.
#include "rcv_internal.h"
#include <string.h>
#include <new>

struct rcv_graph {
    rcv_ctx* ctx;
    hipGraph_t graph;
    hipGraphExec_t exec;
    void* allocs[64];
    int nallocs;
};

extern "C" int rcv_graph_begin(rcv_ctx* ctx)
{
    RCV_TRY(rcv_bind(ctx));
    if (ctx->capturing) return RCV_ERR_ARG;
    ctx->cap_nallocs = 0;
    // relaxed mode: allocation and the side-stream upload of graph-owned constants stay legal while recording
    RCV_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    return RCV_OK;
}

extern "C" int rcv_graph_end(rcv_ctx* ctx, rcv_graph** out)
{
    if (!ctx || !out) return RCV_ERR_ARG;
    *out = nullptr;
    if (!ctx->capturing) return RCV_ERR_ARG;
    RCV_TRY(rcv_bind(ctx));
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    rcv_graph* r = (e == hipSuccess && g) ? new (std::nothrow) rcv_graph() : nullptr;
    if (r) {
        memset(r, 0, sizeof(*r));
        r->ctx = ctx;
        r->graph = g;
        e = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
    }
    if (!r || e != hipSuccess) {
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        for (int i = 0; i < ctx->cap_nallocs; ++i) (void)hipFree(ctx->cap_allocs[i]);
        ctx->cap_nallocs = 0;
        delete r;
        return RCV_ERR_DEVICE;
    }
    r->nallocs = ctx->cap_nallocs;
    memcpy(r->allocs, ctx->cap_allocs, sizeof(void*) * r->nallocs);
    ctx->cap_nallocs = 0;
    ctx->children++;   // the graph keeps using ctx's device and stream: rcv_ctx_destroy defers while it is alive
    *out = r;
    return RCV_OK;
}

extern "C" int rcv_graph_launch(rcv_ctx* ctx, rcv_graph* g)
{
    if (!ctx || !g || g->ctx != ctx) return RCV_ERR_ARG;
    if (ctx->capturing) return RCV_ERR_UNSUPPORTED;
    RCV_TRY(rcv_bind(ctx));
    RCV_HIP(hipGraphLaunch(g->exec, ctx->stream));
    return RCV_OK;
}

extern "C" void rcv_graph_destroy(rcv_graph* g)
{
    if (!g) return;
    if (g->ctx) {
        (void)hipSetDevice(g->ctx->device);
        (void)hipStreamSynchronize(g->ctx->stream);
    }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    for (int i = 0; i < g->nallocs; ++i) (void)hipFree(g->allocs[i]);
    rcv_ctx* owner = g->ctx;
    delete g;
    if (owner) rcv_ctx_child_released(owner);
}
