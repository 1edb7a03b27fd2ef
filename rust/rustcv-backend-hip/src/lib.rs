//! FFI binding of `librustcv_hip.so` (include/rustcv_hip.h) plus shims shaped exactly like the reference
//! functions they replace.  Style follows the one FFI precedent in RustCV,
//! `rustcv-camera/src/backend/macos/mod.rs:42-80` (zero-sized `#[repr(C)]` opaque, `extern "C"` block,
//! `unsafe impl Send` on the owning wrapper, `Drop` calls the C `free`).
//!
//! SOURCE ONLY -- never compiled in the build image (no rustc).  See INTEGRATION.md.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct rcv_ctx {
    _private: [u8; 0],
}

/// Mirror of `rcv_mat` == `rustcv::core::mat::Mat` (rustcv/src/core/mat.rs:6-15) + depth/device.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rcv_mat {
    pub data: *mut c_void,
    pub cap: usize,
    pub step: usize,
    pub rows: i32,
    pub cols: i32,
    pub channels: u8,
    pub depth: u8,
    pub device: u8,
    pub reserved: u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rcv_batch {
    pub frame0: rcv_mat,
    pub frame_stride: usize,
    pub n: i32,
    pub reserved: i32,
}

/// One rasterised glyph box: `bounding_box.min`, size, and where its w*h coverage values start.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rcv_glyph {
    pub x: i32,
    pub y: i32,
    pub w: i32,
    pub h: i32,
    pub offset: u64,
}

#[repr(C)]
pub struct rcv_ring { _private: [u8; 0] }
pub type rcv_ring_op = extern "C" fn(ctx: *mut rcv_ctx, dev_in: *const rcv_mat, dev_out: *mut rcv_mat, user: *mut c_void) -> c_int;

pub const RCV_ERR_BUSY: c_int = -6;
pub const RCV_BGRA2BGR_STRIDED: c_int = 11;
pub const RCV_OK: c_int = 0;
pub const RCV_NOOP: c_int = 1;
pub const RCV_YUYV2BGR: c_int = 0;
pub const RCV_BGRA2BGR: c_int = 1;
pub const RCV_RGB2BGR: c_int = 2;
pub const RCV_BGR2GRAY: c_int = 5;

extern "C" {
    pub fn rcv_strerror(code: c_int) -> *const c_char;
    pub fn rcv_device_count(n: *mut c_int) -> c_int;
    pub fn rcv_ctx_create(device: c_int, out: *mut *mut rcv_ctx) -> c_int;
    pub fn rcv_ctx_destroy(ctx: *mut rcv_ctx);
    pub fn rcv_sync(ctx: *mut rcv_ctx) -> c_int;
    pub fn rcv_malloc(ctx: *mut rcv_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn rcv_free(ctx: *mut rcv_ctx, p: *mut c_void) -> c_int;
    pub fn rcv_upload(ctx: *mut rcv_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn rcv_download(ctx: *mut rcv_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn rcv_cvt_color(ctx: *mut rcv_ctx, code: c_int, src: *const rcv_mat, dst: *mut rcv_mat) -> c_int;
    pub fn rcv_cvt_color_batch(ctx: *mut rcv_ctx, code: c_int, src: *const rcv_batch, dst: *mut rcv_batch) -> c_int;
    pub fn rcv_rectangle(ctx: *mut rcv_ctx, mat: *mut rcv_mat, x: i32, y: i32, w: i32, h: i32, b: u8, g: u8, r: u8, thickness: i32) -> c_int;
    pub fn rcv_blend_glyphs(ctx: *mut rcv_ctx, mat: *mut rcv_mat, glyphs: *const rcv_glyph, n_glyphs: i32, coverage: *const f32,
                            n_coverage: u64, b: u8, g: u8, r: u8) -> c_int;
    pub fn rcv_gaussian_blur(ctx: *mut rcv_ctx, src: *const rcv_mat, dst: *mut rcv_mat, ksize: c_int, sigma: f64) -> c_int;
    pub fn rcv_filter2d_i8(ctx: *mut rcv_ctx, src: *const rcv_mat, dst: *mut rcv_mat, k: *const i8, ksize: c_int, shift: c_int) -> c_int;
    pub fn rcv_filter2d_i8_batch(ctx: *mut rcv_ctx, src: *const rcv_batch, dst: *mut rcv_batch, k: *const i8, ksize: c_int, shift: c_int) -> c_int;
    pub fn rcv_filter2d_f32(ctx: *mut rcv_ctx, src: *const rcv_mat, dst: *mut rcv_mat, k: *const f32, ksize: c_int, delta: f32) -> c_int;
    pub fn rcv_sobel(ctx: *mut rcv_ctx, src: *const rcv_mat, dx: *mut rcv_mat, dy: *mut rcv_mat) -> c_int;
    pub fn rcv_resize(ctx: *mut rcv_ctx, src: *const rcv_mat, dst: *mut rcv_mat) -> c_int;
    pub fn rcv_warp_affine(ctx: *mut rcv_ctx, src: *const rcv_mat, dst: *mut rcv_mat, m: *const f32) -> c_int;
    pub fn rcv_warp_affine_resize(ctx: *mut rcv_ctx, src: *const rcv_mat, dst: *mut rcv_mat, m: *const f32, mid_rows: c_int, mid_cols: c_int) -> c_int;
    pub fn rcv_filter2d_i8_yuyv(ctx: *mut rcv_ctx, src_yuyv: *const rcv_mat, dst_bgr: *mut rcv_mat, k: *const i8, ksize: c_int, shift: c_int) -> c_int;
    // pinned-host staging ring: `depth` frames in flight, H2D / kernels / D2H overlap
    pub fn rcv_ring_create(ctx: *mut rcv_ctx, depth: c_int, in_rows: c_int, in_cols: c_int, in_channels: c_int, in_depth: c_int,
                           out_rows: c_int, out_cols: c_int, out_channels: c_int, out_depth: c_int, out: *mut *mut rcv_ring) -> c_int;
    pub fn rcv_ring_destroy(ring: *mut rcv_ring);
    pub fn rcv_ring_in_flight(ring: *const rcv_ring) -> c_int;
    pub fn rcv_ring_input(ring: *mut rcv_ring, host_in: *mut rcv_mat) -> c_int;
    pub fn rcv_ring_submit(ring: *mut rcv_ring, host_in: *const rcv_mat, op: rcv_ring_op, user: *mut c_void) -> c_int;
    pub fn rcv_ring_retire(ring: *mut rcv_ring, host_out: *mut rcv_mat, pinned_out: *mut rcv_mat) -> c_int;
    pub fn rcv_corner_harris(ctx: *mut rcv_ctx, gray: *const rcv_mat, resp: *mut rcv_mat, block: c_int, k: f32) -> c_int;
    pub fn rcv_nms3x3(ctx: *mut rcv_ctx, resp: *const rcv_mat, mask: *mut rcv_mat, thr: f32) -> c_int;
    pub fn rcv_harris_pipeline(ctx: *mut rcv_ctx, bgr: *const rcv_mat, mask: *mut rcv_mat, resp: *mut rcv_mat, block: c_int, k: f32, thr: f32) -> c_int;
}

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

/// View of a `rustcv::core::mat::Mat { data, rows, cols, step, channels }` (host memory, u8).
pub fn mat_view(data: &mut [u8], rows: i32, cols: i32, step: usize, channels: u8) -> rcv_mat {
    rcv_mat { data: data.as_mut_ptr() as *mut c_void, cap: data.len(), step, rows, cols, channels, depth: 0, device: 0, reserved: 0 }
}

// ---- shims with the reference's exact signatures -------------------------------------------------------

/// Replaces `fn yuyv_to_bgr(src: &[u8], dest: &mut [u8], width: usize, height: usize)`
/// (rustcv/src/videoio/mod.rs:344).  Silent no-op on short `src`, like the reference.
pub fn yuyv_to_bgr(ctx: &HipContext, src: &[u8], dest: &mut [u8], width: usize, height: usize) {
    let s = rcv_mat { data: src.as_ptr() as *mut c_void, cap: src.len(), step: src.len(), rows: 1, cols: src.len() as i32, channels: 1, depth: 0, device: 0, reserved: 0 };
    let mut d = mat_view(dest, height as i32, width as i32, width * 3, 3);
    let rc = unsafe { rcv_cvt_color(ctx.raw, RCV_YUYV2BGR, &s, &mut d) };
    assert!(rc >= 0, "rustcv_hip: yuyv_to_bgr failed ({rc})"); // the reference panics on a short dest
}

/// Replaces `fn bgra_to_bgr(src, dest, width, height)` (rustcv/src/videoio/mod.rs:385).
pub fn bgra_to_bgr(ctx: &HipContext, src: &[u8], dest: &mut [u8], width: usize, height: usize) {
    let s = rcv_mat { data: src.as_ptr() as *mut c_void, cap: src.len(), step: src.len(), rows: 1, cols: src.len() as i32, channels: 1, depth: 0, device: 0, reserved: 0 };
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
