//! FFI binding of `librustcv_hip.so` (include/rustcv_hip.h) plus shims shaped exactly like the reference
//! functions they replace.  Style follows the one FFI precedent in RustCV,
//! `rustcv-camera/src/backend/macos/mod.rs:42-80` (zero-sized `#[repr(C)]` opaque, `extern "C"` block,
//! `unsafe impl Send` on the owning wrapper, `Drop` calls the C `free`).
//!
//! SOURCE ONLY -- never compiled in the build image (no rustc).  `ffi.rs` is generated from the C header and checked against
//! it by tests/test_abi.py; this file adds the owning handle and the shims with the reference's exact signatures, `imgproc.rs` the
//! safe facade for everything else (tests/test_abi.py checks that the two files together wrap every entry point).  See INTEGRATION.md.
#![allow(non_camel_case_types)]
use std::os::raw::c_void;

/// The raw FFI surface: GENERATED from include/rustcv_hip.h by tools/gen_rust_ffi.py (every constant, struct and function of
/// the C ABI, the way `rustcv-camera/src/backend/macos/mod.rs:52-79` declares the whole of `bridge.h`).
pub mod ffi;
pub use ffi::*;

/// Safe facade for the image operations and the device-resident batch path: `Mat`-shaped borrows, `DeviceBatch`, `StagingRing`,
/// one wrapper per compute entry point (the Rust twin of include/rustcv.hpp).
pub mod imgproc;

/// Owning handle: one GPU + one HIP stream.  Not `Sync`; one thread per context (bridge.h:4-7).
pub struct HipContext {
    raw: *mut rcv_ctx,
}
unsafe impl Send for HipContext {}

impl HipContext {
    pub fn new(device: i32) -> Result<Self, i32> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { rcv_ctx_create(device, &mut raw) };
        if rc != RCV_OK { Err(rc) } else { Ok(Self { raw }) }
    }
    pub fn raw(&self) -> *mut rcv_ctx { self.raw }
}
impl Drop for HipContext {
    fn drop(&mut self) { unsafe { rcv_ctx_destroy(self.raw) } }
}

/// A flat byte buffer (`&[u8]` of the reference's private converters) as an `rcv_mat`: the C side reads its LENGTH from `cap`
/// (= `Vec::len()`, the number the reference's guards test); `cols` only mirrors it and saturates for buffers >= 2 GiB.
fn flat_view(src: &[u8]) -> rcv_mat {
    let cols = i32::try_from(src.len()).unwrap_or(i32::MAX);
    rcv_mat { data: src.as_ptr() as *mut c_void, cap: src.len(), step: src.len(), rows: 1, cols, channels: 1, depth: RCV_8U as u8, device: RCV_HOST as u8, reserved: 0 }
}

/// Contiguous frame range of GPU `rank` of `world` for a batch of `n_frames`: one `HipContext` and one host
/// thread per GPU, no collective; `HipContext` is `Send`, so `std::thread::scope` over the contexts is the whole dispatcher.
/// (The library's own `rcv_shard_range`: one rule for every host language.)
pub fn frame_range(n_frames: usize, rank: usize, world: usize) -> (usize, usize) {
    let (mut first, mut last) = (0i64, 0i64);
    let rc = unsafe { rcv_shard_range(n_frames as i64, rank as i32, world as i32, &mut first, &mut last) };
    assert_eq!(rc, RCV_OK, "frame_range({n_frames}, {rank}, {world})");
    (first as usize, last as usize)
}

/// One context per GPU of the node, owned together (north_star: "independent per-GPU HIP streams, no RCCL
/// collective").  Batch entry points only enqueue work on their context's stream, so ONE host thread can drive every device:
/// `for rank in 0..g.len() { op(g.ctx(rank), frames g.frames(n, rank)) }` and one `g.sync()`.  `ctx(rank)` borrows from the group.
pub struct DeviceGroup {
    raw: *mut rcv_group,
    ctxs: Vec<std::mem::ManuallyDrop<HipContext>>,   // views of the group's contexts: destroyed by rcv_group_destroy, not by Drop
}
unsafe impl Send for DeviceGroup {}

impl DeviceGroup {
    /// GPUs `0..n_devices`; `Err(RCV_ERR_DEVICE)` when the node has fewer.
    pub fn new(n_devices: usize) -> Result<Self, i32> {
        Self::create(std::ptr::null(), n_devices)
    }
    /// An explicit list of device ordinals (an ordinal may repeat: two streams on one GPU).
    pub fn with_devices(devices: &[i32]) -> Result<Self, i32> {
        Self::create(devices.as_ptr(), devices.len())
    }
    /// `depth` contexts (= HIP streams) on ONE GPU.  A frame stream -- the reference's caller, `VideoCapture::read` in a loop
    /// (rustcv/src/videoio/mod.rs:168-265) -- keeps `depth` batches in flight: batch k on `ctx(k % depth)`, every context with
    /// its own source / destination buffers.  Consecutive launches then overlap (64 x 4K 7x7 filter2D at depth 2: one batch per
    /// 0.55 ms instead of 0.61 ms).  Ordering holds per context only.
    pub fn in_flight(device: i32, depth: usize) -> Result<Self, i32> {
        Self::with_devices(&vec![device; depth.max(1)])
    }
    fn create(devices: *const i32, n: usize) -> Result<Self, i32> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { rcv_group_create(devices, n as i32, &mut raw) };
        if rc != RCV_OK {
            return Err(rc);
        }
        let len = unsafe { rcv_group_size(raw) }.max(0) as usize;
        let ctxs = (0..len).map(|r| std::mem::ManuallyDrop::new(HipContext { raw: unsafe { rcv_group_ctx(raw, r as i32) } })).collect();
        Ok(Self { raw, ctxs })
    }
    pub fn len(&self) -> usize { self.ctxs.len() }
    pub fn is_empty(&self) -> bool { self.ctxs.is_empty() }
    pub fn ctx(&self, rank: usize) -> &HipContext { &self.ctxs[rank] }
    /// The half-open frame range `first..last` of a batch of `n_frames` that belongs to `rank`.
    pub fn frames(&self, n_frames: usize, rank: usize) -> (usize, usize) { frame_range(n_frames, rank, self.len()) }
    /// Every context's stream idle; the first error.
    pub fn sync(&self) -> Result<(), i32> {
        let rc = unsafe { rcv_group_sync(self.raw) };
        if rc != RCV_OK { Err(rc) } else { Ok(()) }
    }
}
impl DeviceGroup {
    /// Records an event on every context's stream (rank order).
    pub fn timer_start(&self) -> Result<(), i32> {
        let rc = unsafe { rcv_group_timer_start(self.raw) };
        if rc != RCV_OK { Err(rc) } else { Ok(()) }
    }
    /// Records a second event on every stream, waits for all of them; milliseconds from the first start to the latest stop.
    pub fn timer_stop(&self) -> Result<f32, i32> {
        let mut ms = 0f32;
        let rc = unsafe { rcv_group_timer_stop(self.raw, &mut ms) };
        if rc != RCV_OK { Err(rc) } else { Ok(ms) }
    }
}
impl Drop for DeviceGroup {
    fn drop(&mut self) { unsafe { rcv_group_destroy(self.raw) } }
}

/// A capture buffer that lives in a DMA-BUF, mapped on the GPU without a copy: the consuming side of
/// `rustcv_core::frame::AsDmaBuf::as_dmabuf_fd` (rustcv-core/src/frame.rs:58-65).  The backend keeps its fd (the import works on a
/// duplicate); `ptr` is device memory for `rcv_mat { data: ptr, device: RCV_DEVICE, .. }` until the value is dropped.
pub struct DmaBufImport {
    raw: *mut rcv_import,
    pub ptr: *mut c_void,
    pub len: usize,
}
unsafe impl Send for DmaBufImport {}
impl DmaBufImport {
    pub fn new(ctx: &HipContext, fd: std::os::unix::io::RawFd, offset: usize, len: usize) -> Result<Self, i32> {
        let (mut raw, mut ptr) = (std::ptr::null_mut(), std::ptr::null_mut());
        let rc = unsafe { rcv_import_dmabuf(ctx.raw, fd, offset, len, &mut raw, &mut ptr) };
        if rc != RCV_OK { Err(rc) } else { Ok(Self { raw, ptr, len }) }
    }
}
impl Drop for DmaBufImport {
    fn drop(&mut self) { unsafe { rcv_import_release(self.raw) } }
}

/// View of a `rustcv::core::mat::Mat { data, rows, cols, step, channels }` (host memory, u8).
pub fn mat_view(data: &mut [u8], rows: i32, cols: i32, step: usize, channels: u8) -> rcv_mat {
    rcv_mat { data: data.as_mut_ptr() as *mut c_void, cap: data.len(), step, rows, cols, channels, depth: RCV_8U as u8, device: RCV_HOST as u8, reserved: 0 }
}

// ---- shims with the reference's exact signatures -------------------------------------------------------

/// Replaces `fn yuyv_to_bgr(src: &[u8], dest: &mut [u8], width: usize, height: usize)`
/// (rustcv/src/videoio/mod.rs:344).  Silent no-op on short `src`, like the reference.
pub fn yuyv_to_bgr(ctx: &HipContext, src: &[u8], dest: &mut [u8], width: usize, height: usize) {
    let s = flat_view(src);
    let mut d = mat_view(dest, height as i32, width as i32, width * 3, 3);
    let rc = unsafe { rcv_cvt_color(ctx.raw, RCV_YUYV2BGR, &s, &mut d) };
    assert!(rc >= 0, "rustcv_hip: yuyv_to_bgr failed ({rc})"); // the reference panics on a short dest
}

/// Replaces `fn bgra_to_bgr(src, dest, width, height)` (rustcv/src/videoio/mod.rs:385).
pub fn bgra_to_bgr(ctx: &HipContext, src: &[u8], dest: &mut [u8], width: usize, height: usize) {
    let s = flat_view(src);
    let mut d = mat_view(dest, height as i32, width as i32, width * 3, 3);
    let rc = unsafe { rcv_cvt_color(ctx.raw, RCV_BGRA2BGR, &s, &mut d) };
    assert!(rc >= 0, "rustcv_hip: bgra_to_bgr failed ({rc})");
}

/// The per-pixel half of `put_text` (rustcv/src/imgproc/drawing.rs:123-163).  The caller keeps the layout loop
/// (`font.layout(text, scale, start)`, :128) and, per glyph, pushes `bounding_box.min`, the box size and the values
/// `glyph.draw` yields into `glyphs` / `coverage` instead of blending on the CPU (:137-160); this call then blends all
/// of them on the GPU in the same order with the same f32 arithmetic.
#[allow(clippy::too_many_arguments)]
pub fn blend_glyphs(ctx: &HipContext, data: &mut [u8], rows: i32, cols: i32, step: usize,
                    glyphs: &[rcv_glyph], coverage: &[f32], color: (u8, u8, u8)) {
    let mut m = mat_view(data, rows, cols, step, 3);
    let rc = unsafe {
        rcv_blend_glyphs(ctx.raw, &mut m, glyphs.as_ptr(), glyphs.len() as i32, coverage.as_ptr(), coverage.len() as u64,
                         color.0, color.1, color.2)
    };
    assert!(rc >= 0, "rustcv_hip: blend_glyphs failed ({rc})");
}

/// Replaces `pub fn rectangle(mat: &mut Mat, rect: Rect, color: Scalar, thickness: i32)`
/// (rustcv/src/imgproc/drawing.rs:67).  Takes the Mat's fields so this crate does not depend on `rustcv`.
#[allow(clippy::too_many_arguments)]
pub fn rectangle(ctx: &HipContext, data: &mut [u8], rows: i32, cols: i32, step: usize,
                 rect: (i32, i32, i32, i32), color: (u8, u8, u8), thickness: i32) {
    let mut m = mat_view(data, rows, cols, step, 3);
    let rc = unsafe { rcv_rectangle(ctx.raw, &mut m, rect.0, rect.1, rect.2, rect.3, color.0, color.1, color.2, thickness) };
    assert!(rc >= 0, "rustcv_hip: rectangle failed ({rc})");
}
