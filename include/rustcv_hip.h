/*
 * rustcv_hip.h -- C ABI of librustcv_hip.so, the MI355X (gfx950) backend for the
 * per-pixel hot path of rustcv::imgproc / rustcv::videoio.
 *
 * This is the drop-in boundary.  The reference has no plugin interface for
 * imgproc (its ops are plain `pub fn`s on `&mut Mat`), so the ABI follows the
 * one FFI precedent in the reference, rustcv-camera/src/backend/macos/bridge.h:16-65
 * as consumed by rustcv-camera/src/backend/macos/mod.rs:42-80 :
 *   - opaque handle with explicit create/free          (bridge.h:17,36-39,65)
 *   - int return, 0 = OK, negative = error, no unwinding (bridge.h:20-24)
 *   - caller owns every pixel buffer, capacity is explicit (bridge.h:47-62)
 *   - one thread per handle, handles independent        (bridge.h:4-7)
 * All citations are file:line under /root/reference.  Plain pointers and sizes
 * only; nothing here depends on torch, C++ or HIP types.
 *
 * There is NO CPU fallback behind these symbols.  Every compute entry point
 * returns RCV_ERR_DEVICE when no gfx950 device/context is available.
 */
#ifndef RUSTCV_HIP_H
#define RUSTCV_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCV_ABI_VERSION 1

/* ---- status codes (Rust side maps them to its error enum, cf. macos/mod.rs:82-85) */
#define RCV_OK               0
#define RCV_NOOP             1   /* success; a reference length guard made the call a silent no-op */
#define RCV_ERR_ARG         (-1) /* null pointer, bad enum, bad ksize/shift/block */
#define RCV_ERR_UNSUPPORTED (-2) /* valid but not implemented (channels/depth combination) */
#define RCV_ERR_SIZE        (-3) /* a buffer is smaller than rows*step or than the op needs */
#define RCV_ERR_DEVICE      (-4) /* no device, HIP error, kernel launch failure */
#define RCV_ERR_OOM         (-5) /* device or host allocation failed */
#define RCV_ERR_BUSY        (-6) /* staging ring full: retire a frame first */

/* ---- element depth of an rcv_mat */
#define RCV_8U  0
#define RCV_16S 1
#define RCV_32F 2

/* ---- where rcv_mat.data lives */
#define RCV_HOST   0  /* pageable/pinned host memory: the op stages H2D, runs, stages D2H, and syncs */
#define RCV_DEVICE 1  /* device memory from rcv_malloc (or any hipMalloc on the ctx's device): async on the ctx stream */

/* ---- colour conversion codes for rcv_cvt_color.
 * The first three reproduce the reference's FourCC dispatch
 * (rustcv/src/videoio/mod.rs:201-258; rustcv-camera/src/decode.rs:36-86).      */
#define RCV_YUYV2BGR       0  /* rustcv/src/videoio/mod.rs:344-371 guard: src only            */
#define RCV_BGRA2BGR       1  /* rustcv/src/videoio/mod.rs:385-399 guard: src and dst          */
#define RCV_RGB2BGR        2  /* rustcv-camera/src/decode.rs:213-219 (min whole pixels)        */
#define RCV_YUYV2BGR_TWIN  3  /* rustcv-camera/src/decode.rs:160-191 guard: pairs*4 / pairs*6  */
#define RCV_BGRA2BGR_TWIN  4  /* rustcv-camera/src/decode.rs:200-207 (min whole pixels)        */
#define RCV_BGR2GRAY       5  /* build-defined; stride-aware                  */
/* "next" rows of the scope table: display / codec swizzles and stride-aware capture formats */
#define RCV_BGR2BGRX       6  /* rustcv/src/highgui/mod.rs:125-141 mat_to_u32_buffer: flat BGR -> u32 0x00RRGGBB;
                                 dst: 4 channels, rows*cols pixels, pixels past src.cap/3 are zero            */
#define RCV_BGR2RGB        7  /* rustcv/src/imgcodecs/mod.rs:51-63 (imwrite): src honours step, dst packed    */
#define RCV_YUYV2BGR_STRIDED 8  /* src 2 channels [Y0 U Y1 V], both sides honour step; odd last column untouched */
#define RCV_UYVY2BGR_STRIDED 9  /* src 2 channels [U Y0 V Y1]                                                    */
#define RCV_NV12_2BGR      10 /* src 1 channel, rows x cols luma then ceil(rows/2) rows of interleaved UV at the same
                                 step (rustcv-backend-msmf/examples/camera_view/convert.rs:46-86); RCV_NOOP if short */
#define RCV_BGRA2BGR_STRIDED 11 /* src 4 channels, both sides honour step: bgra_to_bgr per pixel (videoio/mod.rs:385-399)
                                   for backends that report a real row stride (rustcv-backend-avf/src/stream.rs:250-254) */

/* ---- synthetic frame families (replaces the empty rustcv-simulation crate) */
#define RCV_SYNTH_NOISE 0
#define RCV_SYNTH_SCENE 1
#define RCV_SYNTH_YUYV  2

/* Mirror of rustcv::core::mat::Mat (rustcv/src/core/mat.rs:6-15):
 *   data: Vec<u8> -> (data, cap) ; rows, cols: i32 ; step: usize ; channels: u8.
 * `cap` is Vec::len(): the reference's guards (`src.len() < ...`,
 * `idx + 2 < data.len()`) are evaluated against it.  depth/device are additions. */
typedef struct rcv_mat {
    void*    data;
    size_t   cap;      /* addressable bytes at data */
    size_t   step;     /* bytes per row, >= cols*channels*elem_size */
    int32_t  rows;
    int32_t  cols;
    uint8_t  channels;
    uint8_t  depth;    /* RCV_8U / RCV_16S / RCV_32F */
    uint8_t  device;   /* RCV_HOST / RCV_DEVICE */
    uint8_t  reserved;
} rcv_mat;

/* A batch of n equally shaped frames, frame i at frame0.data + i*frame_stride.
 * frame0.cap is the capacity of ONE frame.  Batches must be device-resident.   */
typedef struct rcv_batch {
    rcv_mat  frame0;
    size_t   frame_stride;
    int32_t  n;
    int32_t  reserved;
} rcv_batch;

typedef struct rcv_ctx rcv_ctx; /* opaque: one device + one HIP stream + staging workspace */

/* ---- library / context (replaces nothing; precedent bridge.h:36-39,65) ------- */
int         rcv_abi_version(void);
const char* rcv_strerror(int code);
int         rcv_device_count(int* n);
int         rcv_ctx_create(int device, rcv_ctx** out);
void        rcv_ctx_destroy(rcv_ctx* ctx);
int         rcv_sync(rcv_ctx* ctx);                       /* block until the ctx stream is idle */
/* Every call that waits for the stream (rcv_sync, rcv_download, rcv_upload, the host-Mat entry points, rcv_timer_stop) also reports work the
 * device left undone: RCV_ERR_DEVICE if a chained filter launch did not finish its work lists (an XCD that received no waves under a CU mask /
 * partition mode).  Everything enqueued since the context's last successful wait must then be treated as failed; the error is reported once and
 * the context keeps working (on the kernel that does not depend on wave placement). */
int         rcv_ctx_device(const rcv_ctx* ctx);
void*       rcv_ctx_stream(const rcv_ctx* ctx);           /* the hipStream_t, for event timing by the harness */
/* One context = one stream: everything the library enqueues for a context is ordered on this stream as far as any caller can observe.  Inside, a
 * rcv_filter2d_i8_batch / rcv_harris_pipeline_batch call of 16+ frames runs its second half on a second stream of the context and is not joined
 * per call (the halves of consecutive calls hide each other's launch tails); every other entry point makes the stream wait for that half first.
 * Calling rcv_ctx_stream() does the same and switches this behaviour off for the context for good, because the caller may now enqueue work of its
 * own on the stream.  (Also off: RCV_FR_SPLIT=0; while another context of the device has work in flight.) */

/* ---- device group: one context per GPU of the node, frame-sharded batches ------
 * SURVEY.md 8(e) / north_star "independent per-GPU HIP streams, no RCCL collective": nothing in the reference to
 * replace (one process, one device: rustcv/src/internal/runtime.rs:13).  Batch entry points only enqueue work, so one
 * host thread can drive every device: op on rcv_group_ctx(g, i) over frames rcv_shard_range(n, i, G), then one
 * rcv_group_sync.  devices == NULL: GPUs 0 .. n_devices-1 (RCV_ERR_DEVICE if the node has fewer).                  */
typedef struct rcv_group rcv_group;
int         rcv_group_create(const int* devices, int n_devices, rcv_group** out);
void        rcv_group_destroy(rcv_group* g);
int         rcv_group_size(const rcv_group* g);
rcv_ctx*    rcv_group_ctx(rcv_group* g, int rank);       /* owned by the group; NULL if rank is out of range */
int         rcv_group_sync(rcv_group* g);                /* every context's stream idle; the first error */
/* An ordinal may repeat: {d, d} = two contexts = two HIP streams on GPU d.  That is how a frame STREAM keeps two batches in
 * flight on one GPU (the reference's caller is one: rustcv/src/videoio/mod.rs:168-265 feeds examples/camera_demo.rs:50-76):
 * batch k on context k % 2, each context with its own src/dst buffers; the launches of consecutive batches overlap and fill
 * each other's ramp-up and tail (measured on 64 x 4K 7x7 filter2D: one batch finishes every 0.55 ms instead of every 0.61 ms,
 * DESIGN_HISTORY.md 4.1 round 4).  Ordering holds per context only.
 * Group timer: events on every context's stream; elapsed = latest stop against the first start (per device).                */
int         rcv_group_timer_start(rcv_group* g);
int         rcv_group_timer_stop(rcv_group* g, float* elapsed_ms);   /* records, waits for every stream, returns ms */
/* frames [floor(rank*n/world), floor((rank+1)*n/world)) -- pure host arithmetic, no device needed */
int         rcv_shard_range(int64_t n_frames, int rank, int world, int64_t* first, int64_t* last);

/* ---- device memory for resident batches ------------------------------------- */
int rcv_malloc(rcv_ctx* ctx, size_t bytes, void** out);
int rcv_free(rcv_ctx* ctx, void* p);
int rcv_upload(rcv_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);   /* synchronous */
int rcv_download(rcv_ctx* ctx, void* dst_host, const void* src_device, size_t bytes); /* synchronous */
int rcv_memset(rcv_ctx* ctx, void* dst_device, int value, size_t bytes);

/* ---- stream timing (hipEvent on the ctx stream; used by bench.py) ------------ */
int rcv_timer_start(rcv_ctx* ctx);
int rcv_timer_stop(rcv_ctx* ctx, float* elapsed_ms);      /* records, synchronises, returns ms */

/* ---- host-side helpers (pure CPU, no device needed) -------------------------- */
/* FourCC (little-endian ASCII, rustcv-core/src/pixel_format.rs:10-12) -> cvt code.
 * 'YUYV' -> RCV_YUYV2BGR, 'BGRA' and 'BGR4' -> RCV_BGRA2BGR, 'RGB3' -> RCV_RGB2BGR;
 * anything else RCV_ERR_UNSUPPORTED (twin: DecodeError, decode.rs:77-82).       */
int rcv_fourcc_to_code(uint32_t fourcc, int* code);
/* f32 taps of GaussianBlur(sigma>0): exp(-x^2/2s^2), normalised in f64, cast once */
int rcv_gaussian_taps_f32(int ksize, double sigma, float* taps);

/* ---- a1/a2/a4/a5: colour conversion ------------------------------------------ *
 * replaces yuyv_to_bgr / bgra_to_bgr (rustcv/src/videoio/mod.rs:203,205 call
 * sites) and rustcv_camera::decode::{yuyv_to_bgr,bgra32_to_bgr,rgb_to_bgr}.
 * For codes 0-4 src is a FLAT byte buffer of src->cap bytes (row stride ignored,
 * exactly like the reference) and width/height come from dst->cols/rows;
 * dst is written packed from dst->data.  Returns RCV_NOOP where the reference
 * returns silently.  RCV_BGR2GRAY honours step on both sides.                   */
int rcv_cvt_color(rcv_ctx* ctx, int code, const rcv_mat* src, rcv_mat* dst);
int rcv_cvt_color_batch(rcv_ctx* ctx, int code, const rcv_batch* src, rcv_batch* dst);

/* ---- a3: rectangle ------------------------------------------------------------ *
 * replaces rustcv::imgproc::rectangle (rustcv/src/imgproc/drawing.rs:67-106):
 * in-place outline grown inward, clip to the Mat, 3 channels hard-coded,
 * `idx+2 < data.len()` guard against mat->cap.                                  */
int rcv_rectangle(rcv_ctx* ctx, rcv_mat* mat, int32_t x, int32_t y, int32_t w, int32_t h,
                  uint8_t b, uint8_t g, uint8_t r, int32_t thickness);
int rcv_rectangle_batch(rcv_ctx* ctx, rcv_batch* mats, int32_t x, int32_t y, int32_t w, int32_t h,
                        uint8_t b, uint8_t g, uint8_t r, int32_t thickness);

/* ---- f4: the per-pixel half of put_text --------------------------------------- *
 * replaces the blend closure of rustcv::imgproc::put_text (rustcv/src/imgproc/drawing.rs:137-160).
 * Layout and rasterisation stay on the host (rusttype + a font blob: third-party): the caller hands
 * over, per positioned glyph, the pixel bounding box (`glyph.pixel_bounding_box()`, :134) and the w*h coverage values
 * `glyph.draw` produces (:136), row-major at coverage[offset ..].  The library blends them into the 3-channel u8 Mat in
 * glyph order with the reference's arithmetic (separately rounded f32 multiply / subtract / add, `as u8` saturating
 * truncation, a store after every glyph -- overlapping boxes compose in order), clipped per pixel to the Mat.
 * `glyphs` and `coverage` are HOST pointers, snapshotted before the call returns.  Mats with channels != 3:
 * RCV_ERR_UNSUPPORTED (the reference hard-codes 3, :132); boxes whose coverage range leaves n_coverage: RCV_ERR_SIZE. */
typedef struct rcv_glyph {
    int32_t  x, y;     /* bounding_box.min: top-left pixel of the box in the Mat (any sign; clipped per pixel) */
    int32_t  w, h;     /* box size in pixels (>= 0) */
    uint64_t offset;   /* index of the box's first coverage value */
} rcv_glyph;
int rcv_blend_glyphs(rcv_ctx* ctx, rcv_mat* mat, const rcv_glyph* glyphs, int32_t n_glyphs, const float* coverage,
                     uint64_t n_coverage, uint8_t b, uint8_t g, uint8_t r);
/* the same text on every frame of a device-resident batch */
int rcv_blend_glyphs_batch(rcv_ctx* ctx, rcv_batch* mats, const rcv_glyph* glyphs, int32_t n_glyphs, const float* coverage,
                           uint64_t n_coverage, uint8_t b, uint8_t g, uint8_t r);

/* ---- build-defined ops (not in the reference; spec SURVEY.md 8-A) -------------- *
 * u8, channels in {1,3} unless stated, BORDER_REFLECT_101, arbitrary step.        */
int rcv_gaussian_blur(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, int ksize, double sigma);
int rcv_gaussian_blur_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, int ksize, double sigma);

/* k: ksize*ksize weights row-major (correlation).  ksize odd, 1..7.  0<=shift<=24 */
int rcv_filter2d_i8(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const int8_t* k, int ksize, int shift);
int rcv_filter2d_i8_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const int8_t* k, int ksize, int shift);
int rcv_filter2d_f32(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const float* k, int ksize, float delta);
int rcv_filter2d_f32_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* k, int ksize, float delta);

/* "next" row f1: the capture-side pipeline YUYV -> BGR -> integer filter2D in ONE launch.  src: 2-channel
 * YUYV rows (step honoured, cols even), dst: BGR.  Result == rcv_cvt_color(RCV_YUYV2BGR_STRIDED) followed by
 * rcv_filter2d_i8, bit for bit; the intermediate BGR image never reaches HBM on the fused path.                     */
int rcv_filter2d_i8_yuyv(rcv_ctx* ctx, const rcv_mat* src_yuyv, rcv_mat* dst_bgr, const int8_t* k, int ksize, int shift);
int rcv_filter2d_i8_yuyv_batch(rcv_ctx* ctx, const rcv_batch* src_yuyv, rcv_batch* dst_bgr, const int8_t* k, int ksize, int shift);

/* src u8 1-ch; dx, dy i16 1-ch.  A 3-channel BGR src gives the gradient of its gray conversion (RCV_BGR2GRAY), fused into
 * one launch where the shape allows ("next" row f1: cvtColor -> Sobel chain) */
int rcv_sobel(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dx, rcv_mat* dy);
int rcv_sobel_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dx, rcv_batch* dy);

/* "next" row f1 ("fused filter -> gray -> Sobel"): integer filter2D of a BGR image, RCV_BGR2GRAY of the
 * result and its Sobel gradients in ONE launch -- 3 B read + 4 B written per pixel instead of 13 for the two calls.  src: 3-channel
 * u8; dx, dy: i16 1-ch.  Result == rcv_filter2d_i8 followed by rcv_sobel on its output, bit for bit; the filtered image never
 * reaches HBM on the fused path (other shapes run the two kernels through the context workspace).                       */
int rcv_filter2d_i8_sobel(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dx, rcv_mat* dy, const int8_t* k, int ksize, int shift);
int rcv_filter2d_i8_sobel_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dx, rcv_batch* dy, const int8_t* k, int ksize, int shift);

/* bilinear, output size = dst->rows x dst->cols.  u8 (1 / 3 / 4 channels; result rounded half up) or RCV_32F (1 / 3 / 4 channels:
 * the unrounded interpolated value, same f32 operations in the same order -- SURVEY.md 8-A; src and dst of the same depth) */
int rcv_resize(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst);
int rcv_resize_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst);

/* bilinear, M[6] row-major 2x3 maps dst->src, constant border 0.  u8 or RCV_32F images as for rcv_resize (f32: e.g. the
 * cornerHarris response map; bit-identical to the CPU oracle, inside north_star's 1-ULP allowance) */
int rcv_warp_affine(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const float* M);
int rcv_warp_affine_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* M);

/* "next" row f1: resize(warp_affine(src -> mid_rows x mid_cols), dst) in one call.  When mid is exactly
 * 2x or 4x dst (BGR) the intermediate image is never materialised; other shapes run the two kernels through the context
 * workspace.  Results are identical to calling rcv_warp_affine then rcv_resize. */
int rcv_warp_affine_resize(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const float* M, int mid_rows, int mid_cols);
int rcv_warp_affine_resize_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* M, int mid_rows, int mid_cols);

/* gray u8 1-ch -> f32 response; aperture fixed at 3; block in 1..7 */
int rcv_corner_harris(rcv_ctx* ctx, const rcv_mat* gray, rcv_mat* resp, int block, float k);
int rcv_corner_harris_batch(rcv_ctx* ctx, const rcv_batch* gray, rcv_batch* resp, int block, float k);

/* f32 response -> u8 mask (255 = local max above thr) */
int rcv_nms3x3(rcv_ctx* ctx, const rcv_mat* resp, rcv_mat* mask, float thr);
int rcv_nms3x3_batch(rcv_ctx* ctx, const rcv_batch* resp, rcv_batch* mask, float thr);

/* fused BGR -> gray -> Sobel -> Harris response -> 3x3 NMS in one launch.
 * resp may be NULL (mask only).  A 2-channel src is packed YUYV (SURVEY.md 8(d) config 5 "[or YUYV]"): each macropixel
 * goes through yuyv_to_bgr (rustcv/src/videoio/mod.rs:356-363) first, inside the same launch; a 1-channel src is taken as
 * the gray image itself.                                                                         */
int rcv_harris_pipeline(rcv_ctx* ctx, const rcv_mat* bgr, rcv_mat* mask, rcv_mat* resp,
                        int block, float k, float thr);
int rcv_harris_pipeline_batch(rcv_ctx* ctx, const rcv_batch* bgr, rcv_batch* mask, rcv_batch* resp,
                              int block, float k, float thr);

/* ---- synthetic frames generated on device --------------------- *
 * family NOISE/SCENE: u8, channels 1/3/4.  family YUYV: 2 B/px packed, the mat is
 * described as channels=2.  Frame index of batch element i is frame_base+i.      */
int rcv_synth_batch(rcv_ctx* ctx, rcv_batch* dst, int family, uint64_t seed, uint64_t frame_base);

/* ---- pinned-host staging ring ("next" row f3) ------------------------------------------------------
 * Replaces the upload -> compute -> download serialisation of the host-Mat entry points for streaming callers such as
 * the reference's capture loop (rustcv/src/videoio/mod.rs:83-112): `depth` frames are in flight, H2D, kernels and D2H of
 * different frames overlap on three streams.  The per-frame work is the caller's callback: it receives DEVICE mats and
 * must enqueue on the context's stream without synchronising (every rcv_* entry point does exactly that for
 * RCV_DEVICE mats).  rcv_ring_input exposes the next pinned input buffer so that a capture backend can fill it in
 * place (the zero-copy hand-over the reference declares as AsDmaBuf, rustcv-core/src/frame.rs:58-65, and never
 * implements). */
typedef struct rcv_ring rcv_ring;
typedef int (*rcv_ring_op)(rcv_ctx* ctx, const rcv_mat* dev_in, rcv_mat* dev_out, void* user);

int  rcv_ring_create(rcv_ctx* ctx, int depth, int in_rows, int in_cols, int in_channels, int in_depth,
                     int out_rows, int out_cols, int out_channels, int out_depth, rcv_ring** out);
void rcv_ring_destroy(rcv_ring* ring);
/* frames submitted and not yet retired */
int  rcv_ring_in_flight(const rcv_ring* ring);
/* the pinned host buffer the next submit will upload (fill it, then submit with host_in = NULL); RCV_ERR_BUSY if full */
int  rcv_ring_input(rcv_ring* ring, rcv_mat* host_in);
/* copy host_in (any step) into the slot, then enqueue H2D -> op -> D2H.  Returns the op's error code if it failed
 * (the slot still has to be retired), RCV_ERR_BUSY if `depth` frames are already in flight */
int  rcv_ring_submit(rcv_ring* ring, const rcv_mat* host_in, rcv_ring_op op, void* user);
/* wait for the OLDEST frame; copy it to host_out (may be NULL) and/or describe the ring's own pinned output buffer in
 * *pinned_out (may be NULL).  The pinned view belongs to the slot just retired, which the NEXT rcv_ring_submit may reuse
 * (always, when the ring was full): it is valid only until the next rcv_ring_submit.  RCV_NOOP if nothing is in flight */
int  rcv_ring_retire(rcv_ring* ring, rcv_mat* host_out, rcv_mat* pinned_out);

/* ---- zero-copy import of a DMA-BUF capture buffer ----------------------------------------------------------------------
 * The consuming side of the reference's declared-but-unimplemented hand-over `AsDmaBuf::as_dmabuf_fd(&self) -> Option<RawFd>`
 * (rustcv-core/src/frame.rs:58-65: "feed the fd to CUDA/Vulkan; do not close it, ownership belongs to the backend").  The fd is
 * duplicated -- the caller keeps and later closes its own -- imported as external memory, and bytes [offset, offset + bytes) of
 * it (a plane's data offset) are mapped on the context's device; *dev_ptr is then ordinary RCV_DEVICE memory of `bytes` bytes for every entry point (rcv_mat.data, device = RCV_DEVICE), valid
 * until rcv_import_release.  RCV_ERR_DEVICE when the runtime cannot import the buffer. */
typedef struct rcv_import rcv_import;
int  rcv_import_dmabuf(rcv_ctx* ctx, int dmabuf_fd, size_t offset, size_t bytes, rcv_import** out, void** dev_ptr);
void rcv_import_release(rcv_import* imp);

#ifdef __cplusplus
}
#endif
#endif /* RUSTCV_HIP_H */
