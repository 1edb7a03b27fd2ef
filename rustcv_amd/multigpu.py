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
