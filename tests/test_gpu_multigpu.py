"""Multi-device dispatch: one context + one host thread per GPU, contiguous frame ranges, no collective.
The sharded result, concatenated, must be byte-identical to the single-GPU result.  On a one-GPU box the two-device cases are
skipped (they need hardware that is not there); the one-device group and the error paths still run."""
import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi, device

pytestmark = pytest.mark.gpu


def _filter_shard(ctx, f0, f1, rows, cols, k, seed):
    n = f1 - f0
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 1, seed, f0)                 # frame numbers are global: the generator is counter-based
    device.filter2d(src, dst, k, shift=6)
    out = dst.download()
    src.free()
    dst.free()
    return out


def test_device_group_one_device_equals_plain_context(ctx):
    k = (np.arange(49, dtype=np.int8).reshape(7, 7) % 17) - 8
    want = _filter_shard(ctx, 0, 5, 96, 256, k, 0xABC)
    with rcv.DeviceGroup(1) as g:
        outs = g.run(lambda r, c: _filter_shard(c, *g.frames(5, r), 96, 256, k, 0xABC))
    assert len(outs) == 1 and np.array_equal(outs[0], want)


def test_device_group_rejects_missing_devices():
    have = rcv.device_count()
    with pytest.raises(RuntimeError):
        rcv.DeviceGroup(have + 1)
    with pytest.raises(RuntimeError):
        rcv.DeviceGroup([0, have])


def test_a_failing_rank_does_not_hang_the_group():
    with rcv.DeviceGroup(1) as g:
        with pytest.raises(ZeroDivisionError):
            g.run(lambda r, c: 1 // 0)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_sharded_equals_single_gpu(ctx, world):
    if rcv.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {rcv.device_count()}")
    k = (np.arange(49, dtype=np.int8).reshape(7, 7) % 17) - 8
    n, rows, cols = 13, 270, 512                   # 13 frames: uneven shards
    want = _filter_shard(ctx, 0, n, rows, cols, k, 0x5EED)
    with rcv.DeviceGroup(world) as g:
        assert [g.frames(n, r) for r in range(world)] == rcv.shard.all_ranges(n, world)
        outs = g.run(lambda r, c: _filter_shard(c, *g.frames(n, r), rows, cols, k, 0x5EED))
    assert np.array_equal(np.concatenate(outs, axis=0), want)


def test_two_contexts_on_one_device_in_two_threads(ctx):
    """the threading rule of the ABI (one thread per context, contexts independent) exercised on the hardware that is here: two
    contexts on GPU 0 driven concurrently by two host threads over disjoint frame ranges"""
    import threading
    k = (np.arange(49, dtype=np.int8).reshape(7, 7) % 17) - 8
    n, rows, cols = 8, 270, 512
    want = _filter_shard(ctx, 0, n, rows, cols, k, 0x77)
    ctxs = [rcv.Context(0), rcv.Context(0)]
    outs, errs = [None, None], []

    def body(r):
        try:
            f0, f1 = rcv.shard.frame_range(n, r, 2)
            for _ in range(3):                    # a few rounds: the two streams really overlap
                outs[r] = _filter_shard(ctxs[r], f0, f1, rows, cols, k, 0x77)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for c in ctxs:
        c.close()
    assert not errs, errs
    assert np.array_equal(np.concatenate(outs, axis=0), want)


def test_native_group_one_thread_drives_every_context(oracle):
    """rcv_group_* (include/rustcv_hip.h): three contexts -- three streams -- of GPU 0 as one group; a batch of 7 frames is dealt
    2 / 2 / 3 by rcv_shard_range, ONE host thread queues the filter on every context and waits once; each shard equals the oracle.
    (The N-device form of the same loop is what a one-GPU box cannot run; the group code does not depend on the ordinals.)"""
    import numpy as np
    from rustcv_amd import device
    from rustcv_amd.multigpu import NativeGroup
    n, rows, cols = 7, 96, 256
    k = np.arange(-24, 25, dtype=np.int8).reshape(7, 7)
    with NativeGroup([0, 0, 0]) as g:
        assert g.world == 3 and [c.device for c in g.ctxs] == [0, 0, 0]
        srcs, dsts, covered = [], [], 0
        for r in range(g.world):
            f0, f1 = g.frames(n, r)
            assert f0 == covered
            covered = f1
            s = device.DeviceBatch(g.ctxs[r], f1 - f0, rows, cols, 3)
            device.synth(s, 1, 0x5EED0E00, f0)
            srcs.append(s)
            dsts.append(device.DeviceBatch(g.ctxs[r], f1 - f0, rows, cols, 3))
        assert covered == n
        for r in range(g.world):
            device.filter2d(srcs[r], dsts[r], k, shift=5)       # queued on context r's stream
        g.sync()
        for r in range(g.world):
            frames, got = srcs[r].download(), dsts[r].download()
            for i in range(len(frames)):
                assert np.array_equal(got[i], oracle.filter2d_i8(frames[i], k, 5)), (r, i)
            srcs[r].free()
            dsts[r].free()
    import pytest as _pt
    from rustcv_amd import RcvError
    with _pt.raises(RcvError):
        NativeGroup([0, 4096])       # an ordinal the node does not have: RCV_ERR_DEVICE, nothing left behind

