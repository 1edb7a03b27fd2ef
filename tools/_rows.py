"""tools/_rows.py -- the row-streaming 7x7 filter through the MEASUREMENT entry (librustcv_hip_bench.so: rcv__filter_rows_bench), every plan
parameter an argument (rustcv_amd._ffi.ROWS_TUNE): chain / chain_rows, dbg (4 = memory-only), wpc, rounds, pp, order, bpf, band_rows, taper,
wpb, edge_pct.  Used by tools/ablate_chain*.py, ablate_bands.py, ablate_walk.py."""
import ctypes as C


class Rows:
    def __init__(self, ctx, src, dst, kernel, ksize=7, shift=6):
        from rustcv_amd import _ffi
        self._ffi, self.BL = _ffi, _ffi.bench_lib()
        self.ctx, self.bs, self.bd = ctx, src.as_rcv(), dst.as_rcv()
        self.k = kernel
        self.kp = kernel.ctypes.data_as(C.POINTER(C.c_int8))
        self.ksize, self.shift = ksize, shift

    def fn(self, **tune):
        """a callable that enqueues one launch with this plan"""
        t = self._ffi.rows_tune(**tune)

        def launch():
            rc = self.BL.rcv__filter_rows_bench(self.ctx.handle, C.byref(self.bs), C.byref(self.bd), self.kp, self.ksize, self.shift, t, None)
            assert rc == 0, (rc, tune)
        return launch
