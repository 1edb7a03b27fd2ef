// rcv_ring.hip -- pinned-host staging ring with asynchronous H2D / D2H ("next" row f3).
//
// The reference's capture loop (rustcv/src/videoio/mod.rs:83-112, VideoCapture::read) hands the caller one host frame
// at a time; the single-Mat entry points of this library accept such frames but run upload -> kernels -> download
// back to back on one stream, so PCIe and the GPU take turns.  A ring keeps `depth` frames in flight instead:
//
//     slot i :  pinned host input  --H2D (copy-in stream)-->  device input
//                                       | event
//               caller's op on the context stream:  device input -> device output        (any rcv_*_batch / rcv_* call)
//                                       | event
//               device output --D2H (copy-out stream)--> pinned host output  --event--> rcv_ring_retire
//
// Three streams, three events per slot; frame k+1 uploads and frame k-1 downloads while frame k computes.  The ring
// owns only buffers, streams and events -- what runs per frame is the caller's callback, which must enqueue on
// rcv_ctx_stream(ctx) and must not synchronise.  Capture backends that can fill a buffer of their choice take the
// pinned input directly (rcv_ring_input) and skip the host-side copy (the role the reference leaves to the declared-
// but-unimplemented AsDmaBuf, rustcv-core/src/frame.rs:58-65).
#include "rcv_internal.h"
#include <string.h>
#include <new>

struct rcv_ring {
    rcv_ctx* ctx;
    int depth;
    rcv_mat in_desc, out_desc;     // shape templates (device layout: 256-B aligned rows)
    size_t in_bytes, out_bytes;
    hipStream_t s_in, s_out;
    struct Slot {
        uint8_t *pin_in, *pin_out, *dev_in, *dev_out;
        hipEvent_t ev_in, ev_k, ev_out;
    }* slots;
    unsigned long long head, tail;  // submitted / retired frame counts
    bool counted;                   // registered as a child of ctx (rcv_ctx_destroy defers while children are alive)
};

namespace {

size_t aligned_step(int cols, int ch, int depth) { return ((size_t)cols * ch * rcv_elem_size(depth) + 255) & ~(size_t)255; }

int fill_desc(rcv_mat* m, int rows, int cols, int ch, int depth)
{
    if (rows <= 0 || cols <= 0 || (ch != 1 && ch != 2 && ch != 3 && ch != 4) || rcv_elem_size(depth) == 0) return RCV_ERR_ARG;
    memset(m, 0, sizeof(*m));
    m->rows = rows;
    m->cols = cols;
    m->channels = ch;
    m->depth = depth;
    m->step = aligned_step(cols, ch, depth);
    m->cap = m->step * rows;
    return RCV_OK;
}

// copy rows between two host layouts of the same shape
void copy_rows(uint8_t* dst, size_t dstep, const uint8_t* src, size_t sstep, int rows, size_t rowbytes)
{
    if (dstep == sstep && dstep == rowbytes) {
        memcpy(dst, src, rowbytes * rows);
        return;
    }
    for (int r = 0; r < rows; ++r) memcpy(dst + (size_t)r * dstep, src + (size_t)r * sstep, rowbytes);
}

bool same_shape(const rcv_mat* a, const rcv_mat* d) { return a->rows == d->rows && a->cols == d->cols && a->channels == d->channels && a->depth == d->depth; }

} // namespace

extern "C" int rcv_ring_create(rcv_ctx* ctx, int depth, int in_rows, int in_cols, int in_ch, int in_depth, int out_rows, int out_cols,
                               int out_ch, int out_depth, rcv_ring** out)
{
    if (!ctx || !out || depth < 1 || depth > 64) return RCV_ERR_ARG;
    *out = nullptr;
    RCV_TRY(rcv_bind(ctx));
    rcv_ring* r = new (std::nothrow) rcv_ring();
    if (!r) return RCV_ERR_OOM;
    memset(r, 0, sizeof(*r));
    r->ctx = ctx;
    r->depth = depth;
    int rc = fill_desc(&r->in_desc, in_rows, in_cols, in_ch, in_depth);
    if (rc == RCV_OK) rc = fill_desc(&r->out_desc, out_rows, out_cols, out_ch, out_depth);
    if (rc != RCV_OK) {
        delete r;
        return rc;
    }
    r->in_bytes = r->in_desc.cap;
    r->out_bytes = r->out_desc.cap;
    r->slots = new (std::nothrow) rcv_ring::Slot[depth];
    if (!r->slots) {
        delete r;
        return RCV_ERR_OOM;
    }
    memset(r->slots, 0, sizeof(rcv_ring::Slot) * depth);
    bool ok = hipStreamCreateWithFlags(&r->s_in, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&r->s_out, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < depth; ++i) {
        rcv_ring::Slot& s = r->slots[i];
        ok = hipHostMalloc((void**)&s.pin_in, r->in_bytes, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&s.pin_out, r->out_bytes, hipHostMallocDefault) == hipSuccess &&
             hipMalloc((void**)&s.dev_in, r->in_bytes) == hipSuccess && hipMalloc((void**)&s.dev_out, r->out_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s.ev_k, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        rcv_ring_destroy(r);
        return RCV_ERR_OOM;
    }
    ctx->children++;
    r->counted = true;
    *out = r;
    return RCV_OK;
}

extern "C" void rcv_ring_destroy(rcv_ring* r)
{
    if (!r) return;
    if (r->ctx) (void)hipSetDevice(r->ctx->device);
    if (r->s_in) (void)hipStreamSynchronize(r->s_in);
    if (r->ctx) (void)hipStreamSynchronize(r->ctx->stream);
    if (r->s_out) (void)hipStreamSynchronize(r->s_out);
    for (int i = 0; r->slots && i < r->depth; ++i) {
        rcv_ring::Slot& s = r->slots[i];
        if (s.pin_in) (void)hipHostFree(s.pin_in);
        if (s.pin_out) (void)hipHostFree(s.pin_out);
        if (s.dev_in) (void)hipFree(s.dev_in);
        if (s.dev_out) (void)hipFree(s.dev_out);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_k) (void)hipEventDestroy(s.ev_k);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    }
    delete[] r->slots;
    if (r->s_in) (void)hipStreamDestroy(r->s_in);
    if (r->s_out) (void)hipStreamDestroy(r->s_out);
    rcv_ctx* owner = r->counted ? r->ctx : nullptr;
    delete r;
    if (owner) rcv_ctx_child_released(owner);
}

extern "C" int rcv_ring_in_flight(const rcv_ring* r) { return r ? (int)(r->head - r->tail) : RCV_ERR_ARG; }

// The pinned input buffer the NEXT submit will use, described as a host mat (256-B aligned rows).  A producer may
// fill it in place and then call rcv_ring_submit(ring, NULL, ...).  RCV_ERR_BUSY when the ring is full.
extern "C" int rcv_ring_input(rcv_ring* r, rcv_mat* host_in)
{
    if (!r || !host_in) return RCV_ERR_ARG;
    if (r->head - r->tail >= (unsigned long long)r->depth) return RCV_ERR_BUSY;
    *host_in = r->in_desc;
    host_in->data = r->slots[r->head % r->depth].pin_in;
    host_in->device = RCV_HOST;
    return RCV_OK;
}

extern "C" int rcv_ring_submit(rcv_ring* r, const rcv_mat* host_in, rcv_ring_op op, void* user)
{
    if (!r || !op) return RCV_ERR_ARG;
    if (r->head - r->tail >= (unsigned long long)r->depth) return RCV_ERR_BUSY;
    rcv_ctx* ctx = r->ctx;
    RCV_TRY(rcv_bind(ctx));
    rcv_ring::Slot& s = r->slots[r->head % r->depth];
    if (host_in) {
        if (host_in->device != RCV_HOST || !host_in->data || !same_shape(host_in, &r->in_desc)) return RCV_ERR_ARG;
        const size_t rowbytes = (size_t)r->in_desc.cols * r->in_desc.channels * rcv_elem_size(r->in_desc.depth);
        if (host_in->step < rowbytes || host_in->cap < (size_t)(host_in->rows - 1) * host_in->step + rowbytes) return RCV_ERR_ARG;
        copy_rows(s.pin_in, r->in_desc.step, (const uint8_t*)host_in->data, host_in->step, host_in->rows, rowbytes);
    }
    // (the slot's previous occupant was retired, i.e. its D2H event was waited for on the host: all three of the slot's
    //  buffers are idle)
    RCV_HIP(hipMemcpyAsync(s.dev_in, s.pin_in, r->in_bytes, hipMemcpyHostToDevice, r->s_in));
    RCV_HIP(hipEventRecord(s.ev_in, r->s_in));
    RCV_HIP(hipStreamWaitEvent(ctx->stream, s.ev_in, 0));
    rcv_mat din = r->in_desc, dout = r->out_desc;
    din.data = s.dev_in;
    din.device = RCV_DEVICE;
    dout.data = s.dev_out;
    dout.device = RCV_DEVICE;
    const int rc = op(ctx, &din, &dout, user);
    // Once the op has been invoked the frame is in flight: the slot's events are recorded and `head` advances even when the
    // op or one of the calls below failed, so that the slot can be retired and the ring stays consistent.
    hipError_t e = hipEventRecord(s.ev_k, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(r->s_out, s.ev_k, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(s.pin_out, s.dev_out, r->out_bytes, hipMemcpyDeviceToHost, r->s_out);
    const hipError_t e2 = hipEventRecord(s.ev_out, r->s_out);   // (always: retire waits on this event)
    r->head++;
    if (e != hipSuccess || e2 != hipSuccess) {
        (void)hipGetLastError();
        return rc < 0 ? rc : RCV_ERR_DEVICE;
    }
    return rc < 0 ? rc : RCV_OK;
}

// Wait for the oldest frame in flight.  host_out may be NULL; *pinned_out (optional) receives a view of the ring's own
// pinned output buffer.  That buffer belongs to the slot just retired, which is the slot the NEXT rcv_ring_submit may use
// (always, when the ring was full): the view is valid only until the next rcv_ring_submit -- consume or copy it before
// submitting again.  RCV_NOOP when nothing is in flight.
extern "C" int rcv_ring_retire(rcv_ring* r, rcv_mat* host_out, rcv_mat* pinned_out)
{
    if (!r) return RCV_ERR_ARG;
    if (r->head == r->tail) return RCV_NOOP;
    RCV_TRY(rcv_bind(r->ctx));
    rcv_ring::Slot& s = r->slots[r->tail % r->depth];
    if (host_out) {
        if (host_out->device != RCV_HOST || !host_out->data || !same_shape(host_out, &r->out_desc)) return RCV_ERR_ARG;
        const size_t rowbytes = (size_t)r->out_desc.cols * r->out_desc.channels * rcv_elem_size(r->out_desc.depth);
        if (host_out->step < rowbytes || host_out->cap < (size_t)(host_out->rows - 1) * host_out->step + rowbytes) return RCV_ERR_ARG;
    }
    RCV_HIP(hipEventSynchronize(s.ev_out));
    if (host_out) {
        const size_t rowbytes = (size_t)r->out_desc.cols * r->out_desc.channels * rcv_elem_size(r->out_desc.depth);
        copy_rows((uint8_t*)host_out->data, host_out->step, s.pin_out, r->out_desc.step, host_out->rows, rowbytes);
    }
    if (pinned_out) {
        *pinned_out = r->out_desc;
        pinned_out->data = s.pin_out;
        pinned_out->device = RCV_HOST;
    }
    r->tail++;
    return RCV_OK;
}
