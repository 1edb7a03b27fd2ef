// rcv_import.hip -- zero-copy import of a capture buffer that lives in a Linux DMA-BUF (second half).
// The reference declares the hand-over but nothing implements it: `AsDmaBuf::as_dmabuf_fd(&self) -> Option<RawFd>`
// (rustcv-core/src/frame.rs:58-65: "get the underlying DMA-BUF fd to feed CUDA/Vulkan; do not close it, ownership belongs to
// the backend").  This is the consuming side for the HIP backend: the fd is duplicated (the caller keeps its own), imported
// as external memory and mapped into the device address space; the returned pointer is ordinary RCV_DEVICE memory for every
// entry point of the ABI -- a V4L2 / DRM capture buffer then feeds rcv_cvt_color without touching the host.
#include "rcv_internal.h"
#include <new>
#include <string.h>
#include <unistd.h>

struct rcv_import {
    rcv_ctx* ctx;
    hipExternalMemory_t ext;
    void* dev;
    size_t bytes;
    int fd;   // OUR duplicate of the caller's descriptor: this object owns it from dup() until rcv_import_release closes it (the HIP
              // runtime imports through it and does not take it over; the caller's own fd is never closed here)
};

extern "C" int rcv_import_dmabuf(rcv_ctx* ctx, int dmabuf_fd, size_t offset, size_t bytes, rcv_import** out, void** dev_ptr)
{
    if (!out || !dev_ptr) return RCV_ERR_ARG;
    *out = nullptr;
    *dev_ptr = nullptr;
    if (dmabuf_fd < 0 || bytes == 0) return RCV_ERR_ARG;
    if (offset + bytes < offset) return RCV_ERR_SIZE;   // offset + bytes wraps size_t
    RCV_TRY(rcv_bind(ctx));
    rcv_import* im = new (std::nothrow) rcv_import();
    if (!im) return RCV_ERR_OOM;
    im->ctx = ctx;
    im->ext = nullptr;
    im->dev = nullptr;
    im->bytes = bytes;
    im->fd = dup(dmabuf_fd);   // "do not close it": the backend keeps its descriptor, the import works on a duplicate
    if (im->fd < 0) {
        delete im;
        return RCV_ERR_ARG;
    }
    hipExternalMemoryHandleDesc hd;
    memset(&hd, 0, sizeof(hd));
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;
    hd.handle.fd = im->fd;
    // the external-memory object is the WHOLE DMA-BUF (its size from lseek, as for any dma-buf fd); [offset, offset + bytes) of it
    // is mapped below (a capture plane's data_offset; an exporter that sub-allocates)
    // A DMA-BUF that reports a size SMALLER than offset + bytes cannot hold the described mapping: refuse it (mapping `bytes`
    // anyway would hand out a device pointer that runs past the real buffer).  The caller's size is only trusted when the
    // descriptor cannot report one (lseek fails or returns 0: some exporters do not implement it).
    const off_t total = lseek(im->fd, 0, SEEK_END);
    if (total > 0 && (size_t)total < offset + bytes) {
        close(im->fd);
        delete im;
        return RCV_ERR_SIZE;
    }
    hd.size = total > 0 ? (size_t)total : offset + bytes;
    hipError_t e = hipImportExternalMemory(&im->ext, &hd);
    if (e == hipSuccess) {
        hipExternalMemoryBufferDesc bd;
        memset(&bd, 0, sizeof(bd));
        bd.offset = offset;
        bd.size = bytes;
        e = hipExternalMemoryGetMappedBuffer(&im->dev, im->ext, &bd);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (im->ext) (void)hipDestroyExternalMemory(im->ext);
        close(im->fd);
        delete im;
        return e == hipErrorOutOfMemory ? RCV_ERR_OOM : RCV_ERR_DEVICE;
    }
    ctx->children++;   // the mapping belongs to this context's device: rcv_ctx_destroy defers while it is alive
    *out = im;
    *dev_ptr = im->dev;
    return RCV_OK;
}

extern "C" void rcv_import_release(rcv_import* im)
{
    if (!im) return;
    rcv_ctx* ctx = im->ctx;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        if (ctx->half) (void)hipStreamSynchronize(ctx->half);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);   // kernels may still read the mapping
    }
    if (im->ext) (void)hipDestroyExternalMemory(im->ext);
    if (im->fd >= 0) close(im->fd);
    (void)hipGetLastError();
    delete im;
    if (ctx) rcv_ctx_child_released(ctx);
}
