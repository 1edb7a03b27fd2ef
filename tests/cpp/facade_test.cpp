// The reference's own tests for this path (rustcv-camera/src/decode.rs:234-273) and a rectangle check,
// written against the C++ facade exactly as they read in Rust.  Needs a gfx950 GPU to RUN; the CPU suite
// only compiles and links it.  Exit code 0 = all assertions held.
#include <cstdio>
#include <cstdlib>
#include "rustcv.hpp"

#define ASSERT(c)                                                     \
    do {                                                              \
        if (!(c)) {                                                   \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                             \
        }                                                             \
    } while (0)

using namespace rustcv;

static void yuyv_to_bgr_basic()
{
    const uint8_t yuyv[4] = {235, 128, 235, 128};
    std::vector<uint8_t> bgr(6, 0);
    ASSERT(videoio::yuyv_to_bgr(yuyv, 4, bgr, 2, 1));
    for (int i = 0; i < 6; ++i) ASSERT(bgr[i] > 240);
}
static void yuyv_to_bgr_black()
{
    const uint8_t yuyv[4] = {16, 128, 16, 128};
    std::vector<uint8_t> bgr(6, 0);
    ASSERT(videoio::yuyv_to_bgr(yuyv, 4, bgr, 2, 1));
    for (int i = 0; i < 6; ++i) ASSERT(bgr[i] < 10);
}
static void rgb_to_bgr_swap()
{
    const uint8_t rgb[6] = {255, 0, 0, 0, 255, 0};
    std::vector<uint8_t> bgr(6, 0);
    decode::rgb_to_bgr(rgb, 6, bgr);
    const uint8_t want[6] = {0, 0, 255, 0, 255, 0};
    for (int i = 0; i < 6; ++i) ASSERT(bgr[i] == want[i]);
}
static void short_source_is_a_silent_noop()
{
    const uint8_t yuyv[3] = {1, 2, 3};
    std::vector<uint8_t> bgr(6, 7);
    ASSERT(!videoio::yuyv_to_bgr(yuyv, 3, bgr, 2, 1));
    for (int i = 0; i < 6; ++i) ASSERT(bgr[i] == 7);
}
static void rectangle_grows_inward()
{
    Mat m = Mat::create(12, 16, 3);
    imgproc::rectangle(m, imgproc::Rect{2, 3, 8, 6}, imgproc::Scalar{10, 20, 30}, 2);
    for (int y = 0; y < 12; ++y)
        for (int x = 0; x < 16; ++x) {
            bool inside = y >= 3 && y < 9 && x >= 2 && x < 10, inner = y >= 5 && y < 7 && x >= 4 && x < 8;
            const uint8_t* p = m.row_bytes(y) + 3 * x;
            if (inside && !inner) ASSERT(p[0] == 10 && p[1] == 20 && p[2] == 30);
            else ASSERT(p[0] == 0 && p[1] == 0 && p[2] == 0);
        }
}

// put_text's blend on hand-derived values (tests/golden/kat_reference.json "blend_hand_derived"), two overlapping
// boxes composed in glyph order, one box clipped by the Mat
static void glyph_blend_is_ordered_and_clipped()
{
    Mat m = Mat::create(4, 6, 3);
    for (auto& v : m.data) v = 100;
    std::vector<imgproc::Glyph> glyphs;
    glyphs.push_back({1, 1, 2, 1, {0.5f, 1.0f}});        // (1,1): 177.5 -> 177 ; (2,1): colour
    glyphs.push_back({2, 1, 2, 1, {0.5f, 0.25f}});       // (2,1): 255*.5 + 255*.5 = 255 ; (3,1): 63.75 + 75 = 138
    glyphs.push_back({-1, 3, 2, 2, {1.f, 1.f, 1.f, 1.f}}); // only its (1,0) cell lies on the Mat: pixel (0,3)
    imgproc::blendGlyphs(m, glyphs, imgproc::Scalar{255, 255, 255});
    auto px = [&](int x, int y) { return (int)m.row_bytes(y)[3 * x]; };
    ASSERT(px(1, 1) == 177 && px(2, 1) == 255 && px(3, 1) == 138 && px(0, 3) == 255);
    ASSERT(px(0, 0) == 100 && px(4, 1) == 100 && px(1, 3) == 100 && px(0, 2) == 100);
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 6; ++x) ASSERT(m.row_bytes(y)[3 * x] == m.row_bytes(y)[3 * x + 1] && m.row_bytes(y)[3 * x] == m.row_bytes(y)[3 * x + 2]);
}

static void sobel_of_a_ramp()
{
    Mat g = Mat::create(9, 24, 1);
    for (int y = 0; y < 9; ++y)
        for (int x = 0; x < 24; ++x) g.data[(size_t)y * g.step + x] = (uint8_t)(3 * x);
    Mat dx = Mat::create(9, 24, 1, RCV_16S), dy = Mat::create(9, 24, 1, RCV_16S);
    imgproc::Sobel(g, dx, dy);
    const int16_t* px = (const int16_t*)dx.data.data();
    const int16_t* py = (const int16_t*)dy.data.data();
    for (int y = 0; y < 9; ++y)
        for (int x = 0; x < 24; ++x) {
            ASSERT(py[y * 24 + x] == 0);
            ASSERT(px[y * 24 + x] == ((x == 0 || x == 23) ? 0 : 24));   // (p[x+1]-p[x-1]) * (1+2+1); reflect-101 cancels at the edges
        }
}
static void fused_warp_resize_identity_is_a_box_average()
{
    Mat s = Mat::create(8, 16, 3);
    for (size_t i = 0; i < s.data.size(); ++i) s.data[i] = (uint8_t)((i * 37 + 11) & 0xff);
    Mat d = Mat::create(4, 8, 3);
    const float M[6] = {1, 0, 0, 0, 1, 0};
    imgproc::warpAffineResize(s, d, M, 8, 16);
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 8; ++x)
            for (int c = 0; c < 3; ++c) {
                int sum = 2;
                for (int j = 0; j < 2; ++j)
                    for (int i = 0; i < 2; ++i) sum += s.row_bytes(2 * y + j)[3 * (2 * x + i) + c];
                ASSERT(d.row_bytes(y)[3 * x + c] == (uint8_t)(sum >> 2));
            }
}
static int bgra_to_bgr_op(rcv_ctx* ctx, const rcv_mat* din, rcv_mat* dout, void*) { return rcv_cvt_color(ctx, RCV_BGRA2BGR_STRIDED, din, dout); }
static void staging_ring_streams_frames_in_order()
{
    StagingRing ring(2, 6, 10, 4, 6, 10, 3);
    Mat out = Mat::create(6, 10, 3);
    int retired = 0;
    auto check_frame = [&](int f) {
        for (int y = 0; y < 6; ++y)
            for (int x = 0; x < 10; ++x)
                for (int c = 0; c < 3; ++c) ASSERT(out.row_bytes(y)[3 * x + c] == (uint8_t)(f * 50 + y * 10 + x + c));
    };
    for (int f = 0; f < 5; ++f) {
        if (ring.in_flight() == 2) {
            ASSERT(ring.retire(out));
            check_frame(retired++);
        }
        Mat in = Mat::create(6, 10, 4);
        for (int y = 0; y < 6; ++y)
            for (int x = 0; x < 10; ++x)
                for (int c = 0; c < 4; ++c) in.data[(size_t)y * in.step + 4 * x + c] = (uint8_t)(f * 50 + y * 10 + x + c);
        ring.submit(&in, bgra_to_bgr_op);
    }
    while (ring.in_flight()) {
        ASSERT(ring.retire(out));
        check_frame(retired++);
    }
    ASSERT(retired == 5);
    ASSERT(!ring.retire(out));
}

static void device_group_shards_a_batch_from_one_thread()
{
    // three contexts (= three streams) on GPU 0, 7 frames dealt 2 / 2 / 3; every rank's launches are queued before the one sync
    DeviceGroup g(std::vector<int>{0, 0, 0});
    ASSERT(g.size() == 3);
    int64_t covered = 0;
    const int n = 7, rows = 8, cols = 16;
    std::vector<uint8_t> host((size_t)n * rows * cols * 4), out((size_t)n * rows * cols * 3, 0);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (uint8_t)(i * 7 + 3);
    std::vector<void*> din(3, nullptr), dout(3, nullptr);
    for (int r = 0; r < 3; ++r) {
        auto fr = g.frames(n, r);
        ASSERT(fr.first == covered && fr.second >= fr.first);
        covered = fr.second;
        const size_t nf = (size_t)(fr.second - fr.first), fin = (size_t)rows * cols * 4, fout = (size_t)rows * cols * 3;
        check(rcv_malloc(g.ctx(r), nf * fin, &din[r]), "rcv_malloc");
        check(rcv_malloc(g.ctx(r), nf * fout, &dout[r]), "rcv_malloc");
        check(rcv_upload(g.ctx(r), din[r], host.data() + (size_t)fr.first * fin, nf * fin), "rcv_upload");
        rcv_batch bi{}, bo{};
        bi.frame0.data = din[r]; bi.frame0.cap = fin; bi.frame0.step = (size_t)cols * 4; bi.frame0.rows = rows; bi.frame0.cols = cols;
        bi.frame0.channels = 4; bi.frame0.depth = RCV_8U; bi.frame0.device = RCV_DEVICE; bi.frame_stride = fin; bi.n = (int32_t)nf;
        bo = bi;
        bo.frame0.data = dout[r]; bo.frame0.cap = fout; bo.frame0.step = (size_t)cols * 3; bo.frame0.channels = 3; bo.frame_stride = fout;
        check(rcv_cvt_color_batch(g.ctx(r), RCV_BGRA2BGR_STRIDED, &bi, &bo), "rcv_cvt_color_batch");   // queued, not waited for
    }
    ASSERT(covered == n);
    g.sync();
    for (int r = 0; r < 3; ++r) {
        auto fr = g.frames(n, r);
        const size_t nf = (size_t)(fr.second - fr.first), fout = (size_t)rows * cols * 3;
        check(rcv_download(g.ctx(r), out.data() + (size_t)fr.first * fout, dout[r], nf * fout), "rcv_download");
        check(rcv_free(g.ctx(r), din[r]), "rcv_free");
        check(rcv_free(g.ctx(r), dout[r]), "rcv_free");
    }
    for (size_t px = 0; px < (size_t)n * rows * cols; ++px)
        for (int c = 0; c < 3; ++c) ASSERT(out[3 * px + c] == host[4 * px + c]);
    bool threw = false;
    try { g.ctx(3); } catch (const std::out_of_range&) { threw = true; }
    ASSERT(threw);
}

int main()
{
    device_group_shards_a_batch_from_one_thread();
    sobel_of_a_ramp();
    fused_warp_resize_identity_is_a_box_average();
    staging_ring_streams_frames_in_order();
    yuyv_to_bgr_basic();
    yuyv_to_bgr_black();
    rgb_to_bgr_swap();
    short_source_is_a_silent_noop();
    rectangle_grows_inward();
    glyph_blend_is_ordered_and_clipped();
    std::puts("facade_test: all passed");
    return 0;
}
