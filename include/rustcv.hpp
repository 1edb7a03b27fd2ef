// rustcv.hpp -- C++ host facade over the C ABI (include/rustcv_hip.h).
//
// The reference's host side is Rust (rustcv::core::Mat, rustcv::imgproc::*, the private colour helpers of
// rustcv::videoio).  This image has no Rust toolchain, so the host side above the C ABI is written in C++
// with the SAME names, argument order and behaviour, so that code and tests read like the reference's:
//   rustcv::Mat                      <- rustcv/src/core/mat.rs:6-53
//   rustcv::imgproc::{Point,Rect,Scalar,rectangle}  <- rustcv/src/imgproc/drawing.rs:8-106
//   rustcv::imgproc::blendGlyphs                     <- the blend closure of put_text, drawing.rs:137-160
//   rustcv::videoio::{yuyv_to_bgr,bgra_to_bgr}      <- rustcv/src/videoio/mod.rs:344-399
//   rustcv::decode::rgb_to_bgr                       <- rustcv-camera/src/decode.rs:213-219
// The ops the reference does not have follow OpenCV's names.  Header-only; link with
// -lrustcv_hip.  Errors: a negative rcv status becomes std::runtime_error (Rust: Err / panic); the
// reference's silent length-guard returns stay silent (functions return false).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "rustcv_hip.h"

namespace rustcv {

struct Mat {
    std::vector<uint8_t> data;
    int32_t rows = 0, cols = 0;
    size_t step = 0;
    uint8_t channels = 0;
    uint8_t depth = RCV_8U;   // not in the reference (u8 only): i16 / f32 outputs of Sobel and the Harris response

    static Mat create(int32_t rows, int32_t cols, uint8_t channels, uint8_t depth = RCV_8U)  // Mat::new (mat.rs:18-29)
    {
        Mat m;
        m.rows = rows;
        m.cols = cols;
        m.channels = channels;
        m.depth = depth;
        m.step = (size_t)cols * channels * (depth == RCV_8U ? 1 : (depth == RCV_16S ? 2 : 4));
        m.data.assign((size_t)rows * m.step, 0);
        return m;
    }
    static Mat empty() { return Mat(); }                                   // mat.rs:31-39
    bool is_empty() const { return data.empty() || rows == 0 || cols == 0; }  // mat.rs:42-44
    const uint8_t* row_bytes(int32_t row) const { return data.data() + (size_t)row * step; }  // mat.rs:47-51 (cols*channels bytes)

    rcv_mat view() { return view(depth); }
    rcv_mat view(uint8_t depth)
    {
        rcv_mat v{};
        v.data = data.empty() ? nullptr : data.data();
        v.cap = data.size();
        v.step = step;
        v.rows = rows;
        v.cols = cols;
        v.channels = channels;
        v.depth = depth;
        v.device = RCV_HOST;
        return v;
    }
};

class Backend {  // one rcv_ctx per thread of use; Drop -> rcv_ctx_destroy (precedent macos/mod.rs:264-272)
public:
    explicit Backend(int device = 0)
    {
        int rc = rcv_ctx_create(device, &ctx_);
        if (rc != RCV_OK) throw std::runtime_error(std::string("rcv_ctx_create: ") + rcv_strerror(rc));
    }
    ~Backend() { rcv_ctx_destroy(ctx_); }
    Backend(const Backend&) = delete;
    Backend& operator=(const Backend&) = delete;
    rcv_ctx* ctx() const { return ctx_; }
    static Backend& instance()
    {
        static thread_local Backend b(0);
        return b;
    }

private:
    rcv_ctx* ctx_ = nullptr;
};

inline int check(int rc, const char* what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + rcv_strerror(rc));
    return rc;
}

// One context per GPU of the node and the frame partition rule: batch entry points only enqueue work, so one host
// thread drives every device -- for (rank) { op(group.ctx(rank), frames group.frames(n, rank)); } group.sync();  no collective.
class DeviceGroup {
public:
    explicit DeviceGroup(int n_devices) { check(rcv_group_create(nullptr, n_devices, &g_), "rcv_group_create"); }
    explicit DeviceGroup(const std::vector<int>& devices) { check(rcv_group_create(devices.data(), (int)devices.size(), &g_), "rcv_group_create"); }
    ~DeviceGroup() { rcv_group_destroy(g_); }
    DeviceGroup(const DeviceGroup&) = delete;
    DeviceGroup& operator=(const DeviceGroup&) = delete;
    int size() const { return rcv_group_size(g_); }
    rcv_ctx* ctx(int rank) const
    {
        rcv_ctx* c = rcv_group_ctx(g_, rank);
        if (!c) throw std::out_of_range("DeviceGroup::ctx: rank");
        return c;
    }
    // frames [first, last) of a batch of n that belong to `rank`
    std::pair<int64_t, int64_t> frames(int64_t n, int rank) const
    {
        int64_t a = 0, b = 0;
        check(rcv_shard_range(n, rank, size(), &a, &b), "rcv_shard_range");
        return {a, b};
    }
    void sync() { check(rcv_group_sync(g_), "rcv_group_sync"); }
    // `depth` contexts (= HIP streams) on ONE GPU: a frame stream keeps `depth` batches in flight, batch k on ctx(k % depth),
    // each context with its own buffers (rustcv_hip.h, "An ordinal may repeat"; 64 x 4K filter2D: -10 % per batch at depth 2)
    static std::vector<int> in_flight(int device, int depth) { return std::vector<int>((size_t)(depth < 1 ? 1 : depth), device); }
    // events on every context's stream: elapsed ms from the first start to the latest stop
    void timer_start() { check(rcv_group_timer_start(g_), "rcv_group_timer_start"); }
    float timer_stop()
    {
        float ms = 0.0f;
        check(rcv_group_timer_stop(g_, &ms), "rcv_group_timer_stop");
        return ms;
    }

private:
    rcv_group* g_ = nullptr;
};

namespace imgproc {

struct Point { int32_t x, y; };
struct Rect { int32_t x, y, width, height; };
struct Scalar {
    uint8_t v0, v1, v2;  // blue, green, red
    static Scalar all(uint8_t v) { return Scalar{v, v, v}; }
};

inline void rectangle(Mat& mat, Rect rect, Scalar color, int32_t thickness)  // drawing.rs:67
{
    rcv_mat m = mat.view();
    check(rcv_rectangle(Backend::instance().ctx(), &m, rect.x, rect.y, rect.width, rect.height, color.v0, color.v1, color.v2, thickness),
          "rectangle");
}

// One rasterised glyph as put_text's loop has it in hand (drawing.rs:134-136): the pixel bounding box's min corner
// and the coverage values `glyph.draw` yields, row-major h x w.
struct Glyph {
    int32_t x, y, w, h;
    std::vector<float> coverage;
};
// The per-pixel half of put_text (drawing.rs:137-160): ordered alpha blend of the glyphs into the Mat.  Layout and
// rasterisation (rusttype + a font blob) stay with the caller.
inline void blendGlyphs(Mat& mat, const std::vector<Glyph>& glyphs, Scalar color)
{
    std::vector<rcv_glyph> tbl;
    std::vector<float> cov;
    for (const Glyph& g : glyphs) {
        if (g.w < 0 || g.h < 0 || g.coverage.size() < (size_t)g.w * (size_t)g.h) throw std::runtime_error("blendGlyphs: coverage shorter than w*h");
        tbl.push_back(rcv_glyph{g.x, g.y, g.w, g.h, (uint64_t)cov.size()});
        cov.insert(cov.end(), g.coverage.begin(), g.coverage.begin() + (size_t)g.w * (size_t)g.h);
    }
    rcv_mat m = mat.view();
    check(rcv_blend_glyphs(Backend::instance().ctx(), &m, tbl.data(), (int32_t)tbl.size(), cov.data(), cov.size(), color.v0, color.v1,
                           color.v2),
          "blendGlyphs");
}

inline void GaussianBlur(Mat& src, Mat& dst, int ksize, double sigma = 0.0)
{
    rcv_mat s = src.view(), d = dst.view();
    check(rcv_gaussian_blur(Backend::instance().ctx(), &s, &d, ksize, sigma), "GaussianBlur");
}
inline void filter2D(Mat& src, Mat& dst, const int8_t* kernel, int ksize, int shift)
{
    rcv_mat s = src.view(), d = dst.view();
    check(rcv_filter2d_i8(Backend::instance().ctx(), &s, &d, kernel, ksize, shift), "filter2D");
}
inline void filter2D(Mat& src, Mat& dst, const float* kernel, int ksize, float delta = 0.0f)
{
    rcv_mat s = src.view(), d = dst.view();
    check(rcv_filter2d_f32(Backend::instance().ctx(), &s, &d, kernel, ksize, delta), "filter2D");
}
inline void resize(Mat& src, Mat& dst)
{
    rcv_mat s = src.view(), d = dst.view();
    check(rcv_resize(Backend::instance().ctx(), &s, &d), "resize");
}
inline void warpAffine(Mat& src, Mat& dst, const float M[6])
{
    rcv_mat s = src.view(), d = dst.view();
    check(rcv_warp_affine(Backend::instance().ctx(), &s, &d, M), "warpAffine");
}
inline void cvtColor(Mat& src, Mat& dst, int code)
{
    rcv_mat s = src.view(), d = dst.view();
    check(rcv_cvt_color(Backend::instance().ctx(), code, &s, &d), "cvtColor");
}
// gray u8 -> dx, dy (i16 Mats: Mat::create(rows, cols, 1, RCV_16S))
inline void Sobel(Mat& gray, Mat& dx, Mat& dy)
{
    rcv_mat s = gray.view(), a = dx.view(), b = dy.view();
    check(rcv_sobel(Backend::instance().ctx(), &s, &a, &b), "Sobel");
}
// BGR -> corner mask (255 = 3x3 local maximum of the Harris response above thr); resp (f32) is optional
inline void harrisCorners(Mat& bgr, Mat& mask, Mat* resp, int blockSize, float k, float thr)
{
    rcv_mat s = bgr.view(), m = mask.view(), r;
    if (resp) r = resp->view();
    check(rcv_harris_pipeline(Backend::instance().ctx(), &s, &m, resp ? &r : nullptr, blockSize, k, thr), "harrisCorners");
}
// resize(warpAffine(src -> mid_rows x mid_cols), dst) in one call ("next" row f1; fused for exact 2x / 4x BGR down-scales)
inline void warpAffineResize(Mat& src, Mat& dst, const float M[6], int mid_rows, int mid_cols)
{
    rcv_mat s = src.view(), d = dst.view();
    check(rcv_warp_affine_resize(Backend::instance().ctx(), &s, &d, M, mid_rows, mid_cols), "warpAffineResize");
}
// packed / strided YUYV -> BGR -> integer filter2D in one launch ("next" row f1)
inline void filter2D_yuyv(Mat& src_yuyv, Mat& dst_bgr, const int8_t* kernel, int ksize, int shift)
{
    rcv_mat s = src_yuyv.view(), d = dst_bgr.view();
    check(rcv_filter2d_i8_yuyv(Backend::instance().ctx(), &s, &d, kernel, ksize, shift), "filter2D_yuyv");
}

// integer filter2D -> BGR2GRAY -> Sobel of a BGR Mat in one launch ("next" row f1): dx, dy are i16 one-channel Mats
inline void filter2D_sobel(Mat& src_bgr, Mat& dx, Mat& dy, const int8_t* kernel, int ksize, int shift)
{
    rcv_mat s = src_bgr.view(), a = dx.view(), b = dy.view();
    check(rcv_filter2d_i8_sobel(Backend::instance().ctx(), &s, &a, &b, kernel, ksize, shift), "filter2D_sobel");
}

}  // namespace imgproc

// Pinned-host staging ring ("next" row f3): the streaming replacement of the read() loop (rustcv/src/videoio/mod.rs:83-112).
// `op` is a plain function that enqueues rcv_* calls on the context stream for one frame.
class StagingRing {
public:
    StagingRing(int depth, int in_rows, int in_cols, int in_ch, int out_rows, int out_cols, int out_ch, int in_depth = RCV_8U, int out_depth = RCV_8U)
    {
        check(rcv_ring_create(Backend::instance().ctx(), depth, in_rows, in_cols, in_ch, in_depth, out_rows, out_cols, out_ch, out_depth, &ring_),
              "rcv_ring_create");
    }
    ~StagingRing() { rcv_ring_destroy(ring_); }
    StagingRing(const StagingRing&) = delete;
    StagingRing& operator=(const StagingRing&) = delete;
    int in_flight() const { return rcv_ring_in_flight(ring_); }
    // the pinned buffer the next submit(nullptr, ...) uploads; fill it in place
    rcv_mat input()
    {
        rcv_mat m;
        check(rcv_ring_input(ring_, &m), "rcv_ring_input");
        return m;
    }
    void submit(Mat* host_in, rcv_ring_op op, void* user = nullptr)
    {
        rcv_mat m;
        if (host_in) m = host_in->view();
        check(rcv_ring_submit(ring_, host_in ? &m : nullptr, op, user), "rcv_ring_submit");
    }
    // oldest frame -> out (copied); false when nothing was in flight
    bool retire(Mat& out)
    {
        rcv_mat m = out.view();
        return check(rcv_ring_retire(ring_, &m, nullptr), "rcv_ring_retire") == RCV_OK;
    }

private:
    rcv_ring* ring_ = nullptr;
};

namespace videoio {

inline bool convert_flat(int code, const uint8_t* src, size_t src_len, std::vector<uint8_t>& dest, size_t width, size_t height)
{
    rcv_mat s{}, d{};
    s.data = const_cast<uint8_t*>(src);
    s.cap = s.step = src_len;
    s.rows = src_len ? 1 : 0;
    s.cols = (int32_t)src_len;
    s.channels = 1;
    d.data = dest.empty() ? nullptr : dest.data();
    d.cap = dest.size();
    d.step = width * 3;
    d.rows = (int32_t)height;
    d.cols = (int32_t)width;
    d.channels = 3;
    return check(rcv_cvt_color(Backend::instance().ctx(), code, &s, &d), "cvt_color") == RCV_OK;
}
// fn yuyv_to_bgr(src: &[u8], dest: &mut [u8], width: usize, height: usize)   (mod.rs:344)
inline bool yuyv_to_bgr(const uint8_t* src, size_t src_len, std::vector<uint8_t>& dest, size_t width, size_t height)
{
    return convert_flat(RCV_YUYV2BGR, src, src_len, dest, width, height);
}
// fn bgra_to_bgr(src: &[u8], dest: &mut [u8], width: usize, height: usize)   (mod.rs:385)
inline bool bgra_to_bgr(const uint8_t* src, size_t src_len, std::vector<uint8_t>& dest, size_t width, size_t height)
{
    return convert_flat(RCV_BGRA2BGR, src, src_len, dest, width, height);
}

}  // namespace videoio

namespace decode {
// fn rgb_to_bgr(src: &[u8], dst: &mut [u8])   (rustcv-camera/src/decode.rs:213)
inline void rgb_to_bgr(const uint8_t* src, size_t src_len, std::vector<uint8_t>& dst) { videoio::convert_flat(RCV_RGB2BGR, src, src_len, dst, 0, 0); }
}  // namespace decode

}  // namespace rustcv
