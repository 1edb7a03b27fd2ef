#!/usr/bin/env python3
"""tools/cu_mask_probe.py -- (round 6) the chained filter under a REAL CU mask: run as  HSA_CU_MASK=0:0-223 python tools/cu_mask_probe.py  (or with
ROC_GLOBAL_CU_MASK=0x...).  Prints the CU count the runtime reports, the kernel the 16-frame launch took, and whether the wait reported an error / the bytes
equal the oracle's.  Measured on MI355X / ROCm 7.2 (profiles/r06_cu_mask_probe.txt): HSA_CU_MASK leaves 256 CUs reported and every XCD still receives waves
(correct bytes, no error -- late workgroups find their queues drawn empty and leave); ROC_GLOBAL_CU_MASK lowers the reported count: the one-band-per-wave kernel."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from rustcv_amd._ffi import RcvError
from oracle import pyoracle as orc
os.environ["RCV_FR_CHAIN"] = "1"; os.environ["RCV_F7_ROWS"] = "1"
L = _ffi.lib(); L.rcv__debug_reload_knobs()
c = rcv.Context(0)
import torch
print("mask", os.environ.get("HSA_CU_MASK"), os.environ.get("ROC_GLOBAL_CU_MASK"), "torch CUs", torch.cuda.get_device_properties(0).multi_processor_count)
r = np.random.default_rng(5)
n, rows, cols = 16, 80, 1040
frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
k = r.integers(-9, 10, size=(7, 7)).astype(np.int8)
src = device.DeviceBatch(c, n, rows, cols, 3); src.upload(frames)
dst = device.DeviceBatch(c, n, rows, cols, 3); dst.memset(0)
L.rcv__debug_kernels_reset()
try:
    device.filter2d(src, dst, k, shift=5)
    print("kernels:", L.rcv__debug_kernels().decode()[:60])
    c.sync()
    got = dst.download()
    ok = all(np.array_equal(got[i], orc.filter2d_i8(frames[i], k, 5)) for i in range(n))
    print("no error; bytes correct:", ok)
except RcvError as e:
    print("error reported:", e)
    dst.memset(0); L.rcv__debug_kernels_reset()
    device.filter2d(src, dst, k, shift=5); c.sync()
    got = dst.download()
    print("after the error:", L.rcv__debug_kernels().decode()[:50], "bytes correct:", all(np.array_equal(got[i], orc.filter2d_i8(frames[i], k, 5)) for i in range(n)))
