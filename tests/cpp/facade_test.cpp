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

int main()
{
    yuyv_to_bgr_basic();
    yuyv_to_bgr_black();
    rgb_to_bgr_swap();
    short_source_is_a_silent_noop();
    rectangle_grows_inward();
    std::puts("facade_test: all passed");
    return 0;
}
