"""Multi-device dispatch on one node: one `Context` (device + stream + workspace) and ONE HOST THREAD per
GPU, frames [floor(i*N/G), floor((i+1)*N/G)) of a batch on device i (`shard.frame_range`), no collective -- results stay on
the device that produced them.  The C ABI is built for this: a context is single-threaded, different contexts are independent
and may be driven from different threads concurrently (include/rustcv_hip.h, threading rule of bridge.h:4-7); ctypes releases
the GIL for the duration of every library call.

    with DeviceGroup(4) as g:
        outs = g.run(lambda rank, ctx: work_on(ctx, *g.frames(64, rank)))

`bench.py --gpus N` uses this when it is started as ONE process; under `torch.distributed.run` (one process per GPU) the same
partitioning rule is applied per rank instead.
"""
import threading

from . import shard
from .core import Context, device_count


class DeviceGroup:
    def __init__(self, devices):
        """devices: a count (GPUs 0..n-1) or an explicit list of device ordinals.  Raises if the node has fewer GPUs."""
        have = device_count()
        devs = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
        if not devs:
            raise ValueError("DeviceGroup needs at least one device")
        if max(devs) >= have or min(devs) < 0:
            raise RuntimeError(f"DeviceGroup{devs}: this node exposes {have} GPU(s)")
        self.devices = devs
        self.ctxs = [Context(d) for d in devs]
        self.barrier = threading.Barrier(len(devs))

    @property
    def world(self):
        return len(self.ctxs)

    def frames(self, n_frames, rank):
        """the contiguous frame range of `rank`"""
        return shard.frame_range(n_frames, rank, self.world)

    def run(self, fn):
        """fn(rank, ctx) on every device concurrently, one host thread each; returns the results in rank order.  The first
        exception (if any) is re-raised after all threads have finished; a failing rank breaks the group barrier so that the
        others do not wait for it."""
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                out[r] = fn(r, self.ctxs[r])
            except BaseException as e:  # noqa: BLE001  (re-raised below)
                err[r] = e
                self.barrier.abort()

        ts = [threading.Thread(target=body, args=(r,), name=f"rcv-gpu{self.devices[r]}") for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out

    def sync(self):
        for c in self.ctxs:
            c.sync()

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class _GroupContext(Context):
    """a context owned by an rcv_group: same interface, but closing it is the group's business"""

    def __init__(self, handle, device):   # (no rcv_ctx_create)
        self._h, self.device = handle, int(device)

    def close(self):
        self._h = None


class NativeGroup:
    """The library's own device group (`rcv_group_*`, include/rustcv_hip.h): one context per device, created and destroyed
    together, driven from ONE host thread -- batch entry points only enqueue work on their context's stream, so

        with NativeGroup(8) as g:
            for rank in range(g.world):
                op(g.ctxs[rank], *g.frames(n, rank))      # returns as soon as the launch is queued
            g.sync()

    keeps every GPU busy without a thread per device (the dispatcher a Rust / C++ host would write; `DeviceGroup` above is the
    thread-per-device form)."""

    def __init__(self, devices):
        import ctypes as C
        from . import _ffi
        L = _ffi.lib()
        devs = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
        arr = (C.c_int * len(devs))(*devs)
        h = C.c_void_p()
        _ffi.check(L.rcv_group_create(arr, len(devs), C.byref(h)), f"rcv_group_create({devs})")
        self._h, self.devices = h, devs
        self.ctxs = [_GroupContext(C.c_void_p(L.rcv_group_ctx(h, r)), devs[r]) for r in range(L.rcv_group_size(h))]

    @classmethod
    def in_flight(cls, device, depth=2):
        """`depth` contexts (= HIP streams) on ONE GPU: a frame stream keeps `depth` batches in flight, batch k on
        ctxs[k % depth], every context with its own src / dst buffers, so that consecutive launches overlap (64 x 4K 7x7
        filter2D at depth 2: one batch per 0.55 ms instead of 0.61 ms -- DESIGN_HISTORY.md 4.1 round 4).  Ordering holds per context."""
        return cls([int(device)] * max(1, int(depth)))

    def timer_start(self):
        from . import _ffi
        _ffi.check(_ffi.lib().rcv_group_timer_start(self._h), "rcv_group_timer_start")

    def timer_stop(self):
        """ms from the first context's start event to the latest stop event (waits for every stream)"""
        import ctypes as C
        from . import _ffi
        ms = C.c_float(0.0)
        _ffi.check(_ffi.lib().rcv_group_timer_stop(self._h, C.byref(ms)), "rcv_group_timer_stop")
        return float(ms.value)

    @property
    def world(self):
        return len(self.ctxs)

    def frames(self, n_frames, rank):
        import ctypes as C
        from . import _ffi
        a, b = C.c_int64(), C.c_int64()
        _ffi.check(_ffi.lib().rcv_shard_range(int(n_frames), int(rank), self.world, C.byref(a), C.byref(b)), "rcv_shard_range")
        return a.value, b.value

    def sync(self):
        from . import _ffi
        _ffi.check(_ffi.lib().rcv_group_sync(self._h), "rcv_group_sync")

    def close(self):
        """destroys the group's contexts.  Free the DeviceBatches allocated on them FIRST: a batch freed afterwards cannot call rcv_free
        any more (its context handle is gone) and its device memory stays allocated until the process ends."""
        from . import _ffi
        if self._h is not None:
            for c in self.ctxs:
                c.close()
            _ffi.lib().rcv_group_destroy(self._h)
            self._h, self.ctxs = None, []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

