// temporary: shape-specialised kernels not written yet -> everything takes the generic HIP path
#include "rcv_kernels.h"
int rcv_gauss_int_tiled(rcv_ctx*, const View&, const View&, int) { return RCV_ERR_UNSUPPORTED; }
int rcv_sobel_tiled(rcv_ctx*, const View&, const View&, const View&) { return RCV_ERR_UNSUPPORTED; }
int rcv_harris_fused(rcv_ctx*, const View&, const View&, const View*, int, float, float) { return RCV_ERR_UNSUPPORTED; }
