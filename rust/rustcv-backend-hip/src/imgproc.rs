//! Safe Rust facade over the C ABI of `librustcv_hip.so` for the build-defined image operations and the
//! device-resident batch path -- the Rust twin of `include/rustcv.hpp`, one wrapper per compute entry point of
//! `include/rustcv_hip.h`.  `tests/test_abi.py::test_rust_facade_wraps_every_entry_point` parses this file and `lib.rs` and checks
//! that every non-debug C function is called by a wrapper with the header's arity.
//!
//! Shapes follow the reference:
//!   * `MatRef` / `MatMut` borrow the fields of `rustcv::core::mat::Mat { data: Vec<u8>, rows, cols, step, channels }`
//!     (rustcv/src/core/mat.rs:6-15) -- this crate does not depend on `rustcv`, so the facade crate builds them from its `Mat`
//!     (`MatMut::new(&mut m.data, m.rows, m.cols, m.step, m.channels)`); `depth` extends the reference's u8-only Mat to the i16
//!     gradients and the f32 Harris response.
//!   * errors: a negative `rcv` status becomes `Err(HipError)`, the way `rustcv-camera/src/backend/macos/mod.rs:145-164,230-241`
//!     maps the bridge's codes to `CameraError`; the reference's silent length-guard returns (RCV_NOOP) stay `Ok(false)`.
//!   * handles own their C object and free it in `Drop` (macos/mod.rs:264-272).  `HipContext` is `Send`, not `Sync`: one thread per
//!     context (bridge.h:4-7).  `DeviceBatch` and `StagingRing` are neither (since round 4: they hold a raw pointer into their
//!     context's device and must be used and dropped on the thread that owns the context).
//!
//! SOURCE ONLY -- never compiled in the build image (no rustc / cargo); see INTEGRATION.md.
use crate::ffi::*;
use crate::HipContext;
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};

/// `rcv` status codes as a Rust error (include/rustcv_hip.h: RCV_ERR_*).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum HipError {
    Arg,
    Unsupported,
    Size,
    Device,
    OutOfMemory,
    Busy,
    Other(i32),
}

impl HipError {
    pub fn from_code(rc: c_int) -> Self {
        match rc {
            RCV_ERR_ARG => HipError::Arg,
            RCV_ERR_UNSUPPORTED => HipError::Unsupported,
            RCV_ERR_SIZE => HipError::Size,
            RCV_ERR_DEVICE => HipError::Device,
            RCV_ERR_OOM => HipError::OutOfMemory,
            RCV_ERR_BUSY => HipError::Busy,
            other => HipError::Other(other),
        }
    }
    /// The library's own text for the code.
    pub fn message(code: c_int) -> String {
        let p = unsafe { rcv_strerror(code) };
        if p.is_null() {
            return String::new();
        }
        unsafe { std::ffi::CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rustcv_hip: {:?}", self)
    }
}
impl std::error::Error for HipError {}

pub type Result<T> = std::result::Result<T, HipError>;

/// negative -> Err; RCV_OK -> Ok(true); RCV_NOOP (the reference's silent length guard fired) -> Ok(false)
fn status(rc: c_int) -> Result<bool> {
    if rc < 0 {
        Err(HipError::from_code(rc))
    } else {
        Ok(rc == RCV_OK)
    }
}

/// Sample type of a Mat (the reference's Mat is u8 only; i16 = Sobel gradients, f32 = Harris response).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Depth {
    U8,
    I16,
    F32,
}
impl Depth {
    fn code(self) -> u8 {
        (match self {
            Depth::U8 => RCV_8U,
            Depth::I16 => RCV_16S,
            Depth::F32 => RCV_32F,
        }) as u8
    }
    pub fn bytes(self) -> usize {
        match self {
            Depth::U8 => 1,
            Depth::I16 => 2,
            Depth::F32 => 4,
        }
    }
}

/// Read-only borrow of a host Mat (`rustcv::core::mat::Mat`, mat.rs:6-15).
pub struct MatRef<'a> {
    raw: rcv_mat,
    _data: PhantomData<&'a [u8]>,
}
/// Mutable borrow of a host Mat.
pub struct MatMut<'a> {
    raw: rcv_mat,
    _data: PhantomData<&'a mut [u8]>,
}

fn host_mat(data: *mut c_void, len: usize, rows: i32, cols: i32, step: usize, channels: u8, depth: Depth) -> rcv_mat {
    // `cap` = Vec::len(): the number the reference's guards test (videoio/mod.rs:345-348, drawing.rs:82)
    rcv_mat { data, cap: len, step, rows, cols, channels, depth: depth.code(), device: RCV_HOST as u8, reserved: 0 }
}

impl<'a> MatRef<'a> {
    /// `MatRef::new(&m.data, m.rows, m.cols, m.step, m.channels)` for a `rustcv::core::mat::Mat` `m`.
    pub fn new(data: &'a [u8], rows: i32, cols: i32, step: usize, channels: u8) -> Self {
        Self::with_depth(data, rows, cols, step, channels, Depth::U8)
    }
    pub fn with_depth(data: &'a [u8], rows: i32, cols: i32, step: usize, channels: u8, depth: Depth) -> Self {
        Self { raw: host_mat(data.as_ptr() as *mut c_void, data.len(), rows, cols, step, channels, depth), _data: PhantomData }
    }
    /// A flat byte buffer (the `&[u8]` of the reference's private converters, videoio/mod.rs:344,385): one row, one channel; the
    /// C side reads its length from `cap`.
    pub fn flat(data: &'a [u8]) -> Self {
        let cols = i32::try_from(data.len()).unwrap_or(i32::MAX);
        Self::new(data, 1, cols, data.len(), 1)
    }
    pub fn raw(&self) -> &rcv_mat {
        &self.raw
    }
}

impl<'a> MatMut<'a> {
    pub fn new(data: &'a mut [u8], rows: i32, cols: i32, step: usize, channels: u8) -> Self {
        Self::with_depth(data, rows, cols, step, channels, Depth::U8)
    }
    pub fn with_depth(data: &'a mut [u8], rows: i32, cols: i32, step: usize, channels: u8, depth: Depth) -> Self {
        Self { raw: host_mat(data.as_mut_ptr() as *mut c_void, data.len(), rows, cols, step, channels, depth), _data: PhantomData }
    }
    pub fn raw_mut(&mut self) -> &mut rcv_mat {
        &mut self.raw
    }
}

// ---- library-level helpers -------------------------------------------------------------------------------------------------

pub fn abi_version() -> i32 {
    unsafe { rcv_abi_version() }
}

pub fn device_count() -> Result<i32> {
    let mut n: c_int = 0;
    status(unsafe { rcv_device_count(&mut n) })?;
    Ok(n)
}

/// FourCC of a capture format -> conversion code (`rustcv/src/videoio/mod.rs:201-258`, `rustcv-camera/src/decode.rs:36-86`);
/// unknown formats are `Err(Unsupported)` where the twin returns `DecodeError` (decode.rs:77-82).
pub fn fourcc_to_code(fourcc: u32) -> Result<i32> {
    let mut code: c_int = 0;
    status(unsafe { rcv_fourcc_to_code(fourcc, &mut code) })?;
    Ok(code)
}

/// The f32 taps `gaussian_blur` uses for sigma > 0 (f64 normalisation, cast once).
pub fn gaussian_taps_f32(ksize: i32, sigma: f64) -> Result<Vec<f32>> {
    let mut taps = vec![0.0f32; ksize.max(0) as usize];
    status(unsafe { rcv_gaussian_taps_f32(ksize, sigma, taps.as_mut_ptr()) })?;
    Ok(taps)
}

impl HipContext {
    /// Block until everything enqueued on this context's stream has finished.
    pub fn sync(&self) -> Result<()> {
        status(unsafe { rcv_sync(self.raw()) }).map(|_| ())
    }
    pub fn device(&self) -> i32 {
        unsafe { rcv_ctx_device(self.raw()) }
    }
    /// The context's `hipStream_t` (for callers that enqueue their own HIP work in order with the library's).
    pub fn stream(&self) -> *mut c_void {
        unsafe { rcv_ctx_stream(self.raw()) }
    }
    /// HIP-event timing on the context's own stream.
    pub fn timer_start(&self) -> Result<()> {
        status(unsafe { rcv_timer_start(self.raw()) }).map(|_| ())
    }
    pub fn timer_stop(&self) -> Result<f32> {
        let mut ms = 0.0f32;
        status(unsafe { rcv_timer_stop(self.raw(), &mut ms) })?;
        Ok(ms)
    }
}

// ---- host-Mat operations (upload -> kernel -> download; the drop-in forms) ------------------------------------------------

/// `cvt_color(code, src, dst)`: YUYV / BGRA / RGB -> BGR, BGR -> GRAY and the strided / planar capture formats.  `Ok(false)`:
/// the reference's silent length guard fired and `dst` is untouched (videoio/mod.rs:345-348,388-390).
pub fn cvt_color(ctx: &HipContext, code: i32, src: &MatRef, dst: &mut MatMut) -> Result<bool> {
    status(unsafe { rcv_cvt_color(ctx.raw(), code, src.raw(), dst.raw_mut()) })
}

/// Replaces `fn rgb_to_bgr(src: &[u8], dst: &mut [u8])` (rustcv-camera/src/decode.rs:213-219): zips to the shorter buffer.
pub fn rgb_to_bgr(ctx: &HipContext, src: &[u8], dst: &mut [u8]) -> Result<bool> {
    let s = MatRef::flat(src);
    let cols = i32::try_from(dst.len() / 3).unwrap_or(i32::MAX);
    let len = dst.len();
    let mut d = MatMut::new(dst, 1, cols, len, 3);
    cvt_color(ctx, RCV_RGB2BGR, &s, &mut d)
}

/// Replaces `pub fn rectangle(mat: &mut Mat, rect: Rect, color: Scalar, thickness: i32)` (rustcv/src/imgproc/drawing.rs:67).
pub fn rectangle(ctx: &HipContext, mat: &mut MatMut, rect: (i32, i32, i32, i32), color: (u8, u8, u8), thickness: i32) -> Result<()> {
    status(unsafe { rcv_rectangle(ctx.raw(), mat.raw_mut(), rect.0, rect.1, rect.2, rect.3, color.0, color.1, color.2, thickness) }).map(|_| ())
}

/// The per-pixel half of `put_text` (drawing.rs:137-160): ordered alpha blend of rasterised glyphs.
pub fn blend_glyphs(ctx: &HipContext, mat: &mut MatMut, glyphs: &[rcv_glyph], coverage: &[f32], color: (u8, u8, u8)) -> Result<()> {
    status(unsafe {
        rcv_blend_glyphs(ctx.raw(), mat.raw_mut(), glyphs.as_ptr(), glyphs.len() as i32, coverage.as_ptr(), coverage.len() as u64, color.0, color.1, color.2)
    })
    .map(|_| ())
}

/// ksize 3 / 5 / 7 with sigma <= 0: the integer binomial-style taps; sigma > 0: f32 taps, odd ksize up to 31.
pub fn gaussian_blur(ctx: &HipContext, src: &MatRef, dst: &mut MatMut, ksize: i32, sigma: f64) -> Result<()> {
    status(unsafe { rcv_gaussian_blur(ctx.raw(), src.raw(), dst.raw_mut(), ksize, sigma) }).map(|_| ())
}

/// Correlation with i8 weights (row-major ksize x ksize), `(sum + (1 << (shift - 1))) >> shift`, saturated.
pub fn filter2d_i8(ctx: &HipContext, src: &MatRef, dst: &mut MatMut, kernel: &[i8], ksize: i32, shift: i32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    status(unsafe { rcv_filter2d_i8(ctx.raw(), src.raw(), dst.raw_mut(), kernel.as_ptr(), ksize, shift) }).map(|_| ())
}

/// Correlation with f32 weights: `acc = delta; acc = fmaf(k, p, acc)` in tap order, `rintf`, saturated.
pub fn filter2d_f32(ctx: &HipContext, src: &MatRef, dst: &mut MatMut, kernel: &[f32], ksize: i32, delta: f32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    status(unsafe { rcv_filter2d_f32(ctx.raw(), src.raw(), dst.raw_mut(), kernel.as_ptr(), ksize, delta) }).map(|_| ())
}

/// Fused capture chain: packed / strided YUYV (2 channels) -> BGR -> integer filter2D in one launch.
pub fn filter2d_i8_yuyv(ctx: &HipContext, src_yuyv: &MatRef, dst_bgr: &mut MatMut, kernel: &[i8], ksize: i32, shift: i32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    status(unsafe { rcv_filter2d_i8_yuyv(ctx.raw(), src_yuyv.raw(), dst_bgr.raw_mut(), kernel.as_ptr(), ksize, shift) }).map(|_| ())
}

/// 3x3 Sobel of a gray (or BGR: gray conversion fused) u8 Mat into two i16 one-channel Mats.
pub fn sobel(ctx: &HipContext, src: &MatRef, dx: &mut MatMut, dy: &mut MatMut) -> Result<()> {
    status(unsafe { rcv_sobel(ctx.raw(), src.raw(), dx.raw_mut(), dy.raw_mut()) }).map(|_| ())
}

/// BASELINE config 3 in one launch: integer filter2D -> BGR2GRAY -> Sobel; the filtered image never reaches memory.
pub fn filter2d_i8_sobel(ctx: &HipContext, src_bgr: &MatRef, dx: &mut MatMut, dy: &mut MatMut, kernel: &[i8], ksize: i32, shift: i32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    status(unsafe { rcv_filter2d_i8_sobel(ctx.raw(), src_bgr.raw(), dx.raw_mut(), dy.raw_mut(), kernel.as_ptr(), ksize, shift) }).map(|_| ())
}

/// Bilinear, half-pixel centres; the output size is `dst`'s.  u8 or f32 Mats (same depth on both sides).
pub fn resize(ctx: &HipContext, src: &MatRef, dst: &mut MatMut) -> Result<()> {
    status(unsafe { rcv_resize(ctx.raw(), src.raw(), dst.raw_mut()) }).map(|_| ())
}

/// Bilinear, `m` (row-major 2 x 3) maps dst -> src, constant border 0.  u8 or f32 Mats.
pub fn warp_affine(ctx: &HipContext, src: &MatRef, dst: &mut MatMut, m: &[f32; 6]) -> Result<()> {
    status(unsafe { rcv_warp_affine(ctx.raw(), src.raw(), dst.raw_mut(), m.as_ptr()) }).map(|_| ())
}

/// `resize(warp_affine(src -> mid_rows x mid_cols), dst)`; fused (no intermediate image) for exact 2x / 4x BGR down-scales.
pub fn warp_affine_resize(ctx: &HipContext, src: &MatRef, dst: &mut MatMut, m: &[f32; 6], mid_rows: i32, mid_cols: i32) -> Result<()> {
    status(unsafe { rcv_warp_affine_resize(ctx.raw(), src.raw(), dst.raw_mut(), m.as_ptr(), mid_rows, mid_cols) }).map(|_| ())
}

/// gray u8 -> f32 Harris response (aperture 3, block 1..=7).
pub fn corner_harris(ctx: &HipContext, gray: &MatRef, resp: &mut MatMut, block: i32, k: f32) -> Result<()> {
    status(unsafe { rcv_corner_harris(ctx.raw(), gray.raw(), resp.raw_mut(), block, k) }).map(|_| ())
}

/// f32 response -> u8 mask: 255 where `r > thr` and `r >=` each of its 8 neighbours.
pub fn nms3x3(ctx: &HipContext, resp: &MatRef, mask: &mut MatMut, thr: f32) -> Result<()> {
    status(unsafe { rcv_nms3x3(ctx.raw(), resp.raw(), mask.raw_mut(), thr) }).map(|_| ())
}

/// BASELINE config 5 in one launch: BGR (or YUYV / gray) -> gray -> Sobel -> response -> 3x3 NMS -> mask; `resp` is optional.
pub fn harris_pipeline(ctx: &HipContext, src: &MatRef, mask: &mut MatMut, resp: Option<&mut MatMut>, block: i32, k: f32, thr: f32) -> Result<()> {
    let rp = match resp {
        Some(r) => r.raw_mut() as *mut rcv_mat,
        None => std::ptr::null_mut(),
    };
    status(unsafe { rcv_harris_pipeline(ctx.raw(), src.raw(), mask.raw_mut(), rp, block, k, thr) }).map(|_| ())
}

// ---- device-resident batches (the measured path: SURVEY.md 8(b), (e)) ------------------------------------------------------

/// `n` equally shaped frames in ONE device allocation of one GPU; frees it in `Drop`.  Borrows its context: a batch cannot
/// outlive the `HipContext` it was allocated on.  Frames are independent for every op, so multi-GPU is one `HipContext` +
/// `DeviceBatch` per GPU over `crate::frame_range(..)`, one host thread each (`std::thread::scope`), no collective.
pub struct DeviceBatch<'c> {
    ctx: &'c HipContext,
    ptr: *mut c_void,
    pub n: i32,
    pub rows: i32,
    pub cols: i32,
    pub channels: u8,
    pub depth: Depth,
    pub step: usize,
    pub frame_stride: usize,
}
// (NOT `Send`: a batch borrows its `HipContext`, which is deliberately `!Sync` -- one thread per context, bridge.h:4-7; moving a batch to another
//  thread would let two threads drive one rcv_ctx.  Create batches on the thread that owns the context.)

impl<'c> DeviceBatch<'c> {
    /// Packed rows (`step = cols * channels * sample bytes`), frames 256-byte aligned.
    pub fn new(ctx: &'c HipContext, n: i32, rows: i32, cols: i32, channels: u8, depth: Depth) -> Result<Self> {
        let step = cols.max(0) as usize * channels as usize * depth.bytes();
        let frame_stride = (rows.max(0) as usize * step + 255) / 256 * 256;
        let mut ptr = std::ptr::null_mut();
        status(unsafe { rcv_malloc(ctx.raw(), frame_stride * n.max(1) as usize, &mut ptr) })?;
        Ok(Self { ctx, ptr, n, rows, cols, channels, depth, step, frame_stride })
    }
    pub fn bytes(&self) -> usize {
        self.frame_stride * self.n.max(1) as usize
    }
    pub fn as_raw(&self) -> rcv_batch {
        let frame0 = rcv_mat {
            data: self.ptr,
            cap: self.rows.max(0) as usize * self.step,
            step: self.step,
            rows: self.rows,
            cols: self.cols,
            channels: self.channels,
            depth: self.depth.code(),
            device: RCV_DEVICE as u8,
            reserved: 0,
        };
        rcv_batch { frame0, frame_stride: self.frame_stride, n: self.n, reserved: 0 }
    }
    /// Host -> device copy of the whole allocation (frames at `frame_stride`); synchronous.
    pub fn upload(&mut self, host: &[u8]) -> Result<()> {
        if host.len() > self.bytes() {
            return Err(HipError::Size);
        }
        status(unsafe { rcv_upload(self.ctx.raw(), self.ptr, host.as_ptr() as *const c_void, host.len()) }).map(|_| ())
    }
    pub fn download(&self, host: &mut [u8]) -> Result<()> {
        if host.len() > self.bytes() {
            return Err(HipError::Size);
        }
        status(unsafe { rcv_download(self.ctx.raw(), host.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, host.len()) }).map(|_| ())
    }
    pub fn memset(&mut self, value: u8) -> Result<()> {
        status(unsafe { rcv_memset(self.ctx.raw(), self.ptr, value as c_int, self.bytes()) }).map(|_| ())
    }
    /// Deterministic synthetic frames generated on the device: family RCV_SYNTH_NOISE / _SCENE / _YUYV.
    pub fn synth(&mut self, family: i32, seed: u64, frame_base: u64) -> Result<()> {
        let mut b = self.as_raw();
        status(unsafe { rcv_synth_batch(self.ctx.raw(), &mut b, family, seed, frame_base) }).map(|_| ())
    }
}

impl<'c> Drop for DeviceBatch<'c> {
    fn drop(&mut self) {
        unsafe { rcv_free(self.ctx.raw(), self.ptr) };
    }
}

/// Asynchronous on the context's stream (call `ctx.sync()` before reading results on the host), like every `_batch` entry point.
pub fn cvt_color_batch(ctx: &HipContext, code: i32, src: &DeviceBatch, dst: &mut DeviceBatch) -> Result<bool> {
    let (s, mut d) = (src.as_raw(), dst.as_raw());
    status(unsafe { rcv_cvt_color_batch(ctx.raw(), code, &s, &mut d) })
}

pub fn rectangle_batch(ctx: &HipContext, mats: &mut DeviceBatch, rect: (i32, i32, i32, i32), color: (u8, u8, u8), thickness: i32) -> Result<()> {
    let mut b = mats.as_raw();
    status(unsafe { rcv_rectangle_batch(ctx.raw(), &mut b, rect.0, rect.1, rect.2, rect.3, color.0, color.1, color.2, thickness) }).map(|_| ())
}

pub fn blend_glyphs_batch(ctx: &HipContext, mats: &mut DeviceBatch, glyphs: &[rcv_glyph], coverage: &[f32], color: (u8, u8, u8)) -> Result<()> {
    let mut b = mats.as_raw();
    status(unsafe {
        rcv_blend_glyphs_batch(ctx.raw(), &mut b, glyphs.as_ptr(), glyphs.len() as i32, coverage.as_ptr(), coverage.len() as u64, color.0, color.1, color.2)
    })
    .map(|_| ())
}

pub fn gaussian_blur_batch(ctx: &HipContext, src: &DeviceBatch, dst: &mut DeviceBatch, ksize: i32, sigma: f64) -> Result<()> {
    let (s, mut d) = (src.as_raw(), dst.as_raw());
    status(unsafe { rcv_gaussian_blur_batch(ctx.raw(), &s, &mut d, ksize, sigma) }).map(|_| ())
}

/// The north-star entry point: 4K BGR 7x7 integer filter2D over a device-resident batch.
pub fn filter2d_i8_batch(ctx: &HipContext, src: &DeviceBatch, dst: &mut DeviceBatch, kernel: &[i8], ksize: i32, shift: i32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    let (s, mut d) = (src.as_raw(), dst.as_raw());
    status(unsafe { rcv_filter2d_i8_batch(ctx.raw(), &s, &mut d, kernel.as_ptr(), ksize, shift) }).map(|_| ())
}

pub fn filter2d_f32_batch(ctx: &HipContext, src: &DeviceBatch, dst: &mut DeviceBatch, kernel: &[f32], ksize: i32, delta: f32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    let (s, mut d) = (src.as_raw(), dst.as_raw());
    status(unsafe { rcv_filter2d_f32_batch(ctx.raw(), &s, &mut d, kernel.as_ptr(), ksize, delta) }).map(|_| ())
}

pub fn filter2d_i8_yuyv_batch(ctx: &HipContext, src_yuyv: &DeviceBatch, dst_bgr: &mut DeviceBatch, kernel: &[i8], ksize: i32, shift: i32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    let (s, mut d) = (src_yuyv.as_raw(), dst_bgr.as_raw());
    status(unsafe { rcv_filter2d_i8_yuyv_batch(ctx.raw(), &s, &mut d, kernel.as_ptr(), ksize, shift) }).map(|_| ())
}

pub fn sobel_batch(ctx: &HipContext, src: &DeviceBatch, dx: &mut DeviceBatch, dy: &mut DeviceBatch) -> Result<()> {
    let (s, mut a, mut b) = (src.as_raw(), dx.as_raw(), dy.as_raw());
    status(unsafe { rcv_sobel_batch(ctx.raw(), &s, &mut a, &mut b) }).map(|_| ())
}

pub fn filter2d_i8_sobel_batch(ctx: &HipContext, src_bgr: &DeviceBatch, dx: &mut DeviceBatch, dy: &mut DeviceBatch, kernel: &[i8], ksize: i32, shift: i32) -> Result<()> {
    if kernel.len() < (ksize.max(0) as usize).pow(2) {
        return Err(HipError::Arg);
    }
    let (s, mut a, mut b) = (src_bgr.as_raw(), dx.as_raw(), dy.as_raw());
    status(unsafe { rcv_filter2d_i8_sobel_batch(ctx.raw(), &s, &mut a, &mut b, kernel.as_ptr(), ksize, shift) }).map(|_| ())
}

pub fn resize_batch(ctx: &HipContext, src: &DeviceBatch, dst: &mut DeviceBatch) -> Result<()> {
    let (s, mut d) = (src.as_raw(), dst.as_raw());
    status(unsafe { rcv_resize_batch(ctx.raw(), &s, &mut d) }).map(|_| ())
}

pub fn warp_affine_batch(ctx: &HipContext, src: &DeviceBatch, dst: &mut DeviceBatch, m: &[f32; 6]) -> Result<()> {
    let (s, mut d) = (src.as_raw(), dst.as_raw());
    status(unsafe { rcv_warp_affine_batch(ctx.raw(), &s, &mut d, m.as_ptr()) }).map(|_| ())
}

/// BASELINE config 4 in one launch (8K warpAffine + resize -> 1080p: `mid` = 4 x `dst`).
pub fn warp_affine_resize_batch(ctx: &HipContext, src: &DeviceBatch, dst: &mut DeviceBatch, m: &[f32; 6], mid_rows: i32, mid_cols: i32) -> Result<()> {
    let (s, mut d) = (src.as_raw(), dst.as_raw());
    status(unsafe { rcv_warp_affine_resize_batch(ctx.raw(), &s, &mut d, m.as_ptr(), mid_rows, mid_cols) }).map(|_| ())
}

pub fn corner_harris_batch(ctx: &HipContext, gray: &DeviceBatch, resp: &mut DeviceBatch, block: i32, k: f32) -> Result<()> {
    let (s, mut d) = (gray.as_raw(), resp.as_raw());
    status(unsafe { rcv_corner_harris_batch(ctx.raw(), &s, &mut d, block, k) }).map(|_| ())
}

pub fn nms3x3_batch(ctx: &HipContext, resp: &DeviceBatch, mask: &mut DeviceBatch, thr: f32) -> Result<()> {
    let (s, mut d) = (resp.as_raw(), mask.as_raw());
    status(unsafe { rcv_nms3x3_batch(ctx.raw(), &s, &mut d, thr) }).map(|_| ())
}

pub fn harris_pipeline_batch(ctx: &HipContext, src: &DeviceBatch, mask: &mut DeviceBatch, resp: Option<&mut DeviceBatch>, block: i32, k: f32, thr: f32) -> Result<()> {
    let (s, mut m) = (src.as_raw(), mask.as_raw());
    let mut r = resp.map(|b| b.as_raw());
    let rp = match r.as_mut() {
        Some(b) => b as *mut rcv_batch,
        None => std::ptr::null_mut(),
    };
    status(unsafe { rcv_harris_pipeline_batch(ctx.raw(), &s, &mut m, rp, block, k, thr) }).map(|_| ())
}

// ---- pinned staging ring: the streaming replacement of the read() loop (rustcv/src/videoio/mod.rs:83-112) ----------------

/// `depth` frames in flight between a pinned host ring and the GPU: upload, the caller's per-frame op and download overlap.
pub struct StagingRing<'c> {
    raw: *mut rcv_ring,
    _ctx: PhantomData<&'c HipContext>,
}
// (NOT `Send`, for the same reason as DeviceBatch: the ring borrows a `!Sync` context.)

/// Shape of the ring's input or output frames.
#[derive(Debug, Clone, Copy)]
pub struct FrameShape {
    pub rows: i32,
    pub cols: i32,
    pub channels: i32,
    pub depth: Depth,
}

impl<'c> StagingRing<'c> {
    pub fn new(ctx: &'c HipContext, depth: i32, input: FrameShape, output: FrameShape) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        status(unsafe {
            rcv_ring_create(ctx.raw(), depth, input.rows, input.cols, input.channels, input.depth.code() as c_int, output.rows, output.cols, output.channels,
                            output.depth.code() as c_int, &mut raw)
        })?;
        Ok(Self { raw, _ctx: PhantomData })
    }
    pub fn in_flight(&self) -> i32 {
        unsafe { rcv_ring_in_flight(self.raw) }
    }
    /// The next pinned input slot as a host Mat: a capture backend can dequeue straight into it (no extra copy), then `submit(None, ..)`.
    pub fn input(&mut self) -> Result<rcv_mat> {
        let mut m = rcv_mat { data: std::ptr::null_mut(), cap: 0, step: 0, rows: 0, cols: 0, channels: 0, depth: 0, device: RCV_HOST as u8, reserved: 0 };
        status(unsafe { rcv_ring_input(self.raw, &mut m) })?;
        Ok(m)
    }
    /// Enqueue one frame: `host_in` (copied into the pinned slot; `None` = the slot `input()` handed out was filled in place),
    /// H2D, `op` on the context stream, D2H.  `Err(Busy)`: the ring is full -- `retire` a frame first.
    pub fn submit(&mut self, host_in: Option<&MatRef>, op: rcv_ring_op, user: *mut c_void) -> Result<()> {
        let hp = match host_in {
            Some(m) => m.raw() as *const rcv_mat,
            None => std::ptr::null(),
        };
        status(unsafe { rcv_ring_submit(self.raw, hp, op, user) }).map(|_| ())
    }
    /// Wait for the oldest frame in flight and copy it to `host_out` (or `None`: leave it in the pinned slot returned as a Mat).
    pub fn retire(&mut self, host_out: Option<&mut MatMut>) -> Result<rcv_mat> {
        let mut pinned = rcv_mat { data: std::ptr::null_mut(), cap: 0, step: 0, rows: 0, cols: 0, channels: 0, depth: 0, device: RCV_HOST as u8, reserved: 0 };
        let hp = match host_out {
            Some(m) => m.raw_mut() as *mut rcv_mat,
            None => std::ptr::null_mut(),
        };
        status(unsafe { rcv_ring_retire(self.raw, hp, &mut pinned) })?;
        Ok(pinned)
    }
}
impl<'c> Drop for StagingRing<'c> {
    fn drop(&mut self) {
        unsafe { rcv_ring_destroy(self.raw) }
    }
}

