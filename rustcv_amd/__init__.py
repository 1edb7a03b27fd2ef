"""rustcv_amd -- MI355X (gfx950) backend for the per-pixel hot path of rustcv::imgproc / videoio.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/rustcv_hip.h, built into
librustcv_hip.so), and this thin host mirror of the reference's interface for the path:
`Mat` (core), `imgproc`, `videoio`, plus device-resident batches (`device`) and frame sharding
(`shard`, `multigpu`: one context + one host thread per GPU).  Importing the package does not load the shared library; the first call does, and
raises if it is missing.  There is no CPU fallback anywhere in this package.
"""
from . import _ffi
from ._ffi import RcvError
from .core import Context, Mat, default_context, device_count
from . import imgproc, videoio, device, shard, ring, multigpu
from .multigpu import DeviceGroup
from .ring import StagingRing

__all__ = ["Mat", "Context", "RcvError", "default_context", "device_count", "imgproc", "videoio", "device", "shard", "ring", "multigpu", "DeviceGroup", "StagingRing", "_ffi"]
