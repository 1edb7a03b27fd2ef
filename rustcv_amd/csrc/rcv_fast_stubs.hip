// temporary: shape-specialised kernels not written yet -> everything takes the generic HIP path
#include "rcv_kernels.h"
// GaussianBlur(sigma<=0) with ksize 3 / 5 IS an integer filter2D: taps [1,2,1] / [1,4,6,4,1] outer products fit in i8
// (max 4 / 36) and (sum + D/2) / D == (sum + (1 << (s-1))) >> s for D = 2^s and sum >= 0.  ksize 7 has taps up to
// 18*18 = 324 > 127 and stays on the generic kernel.
int rcv_gauss_int_tiled(rcv_ctx* ctx, const View& s, const View& d, int ksize)
{
    if (ksize != 3 && ksize != 5) return RCV_ERR_UNSUPPORTED;
    static const int t3[3] = {1, 2, 1}, t5[5] = {1, 4, 6, 4, 1};
    const int* t = ksize == 3 ? t3 : t5;
    int8_t k[25];
    for (int y = 0; y < ksize; ++y)
        for (int x = 0; x < ksize; ++x) k[y * ksize + x] = (int8_t)(t[y] * t[x]);
    return rcv_filter_i8_fast(ctx, s, d, k, ksize, ksize == 3 ? 4 : 8);
}
