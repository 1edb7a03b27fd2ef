// rcv_gauss_int.hip -- integer GaussianBlur (sigma <= 0) expressed as an integer filter2D on the MFMA strip kernel.
#include "rcv_kernels.h"
// GaussianBlur(sigma<=0) IS an integer filter2D: the 2-D weights are the outer product of the 1-D taps and
// (sum + D/2) / D == (sum + (1 << (s-1))) >> s for D = 2^s and sum >= 0.  ksize 3 / 5 fit i8 (max 4 / 36); ksize 7 has
// weights up to 18*18 = 324 and runs the dual-table variant of the MFMA kernel (K = 4Q + R).
int rcv_gauss_int_tiled(rcv_ctx* ctx, const View& s, const View& d, int ksize)
{
    static const int t3[3] = {1, 2, 1}, t5[5] = {1, 4, 6, 4, 1}, t7[7] = {2, 7, 14, 18, 14, 7, 2};
    const int* t = ksize == 3 ? t3 : (ksize == 5 ? t5 : t7);
    if (ksize != 3 && ksize != 5 && ksize != 7) return RCV_ERR_UNSUPPORTED;
    int16_t k[49];
    for (int y = 0; y < ksize; ++y)
        for (int x = 0; x < ksize; ++x) k[y * ksize + x] = (int16_t)(t[y] * t[x]);
    return rcv_filter_i16_fast(ctx, s, d, k, ksize, ksize == 3 ? 4 : (ksize == 5 ? 8 : 12), 0);
}
