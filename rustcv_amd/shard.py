"""Batch sharding across GPUs: contiguous frame ranges, no collective.

Every op on the path is per-frame independent, so rank i of G processes frames
[floor(i*N/G), floor((i+1)*N/G)) on its own GPU/stream; nothing crosses xGMI.  The only
communication is the harness exchanging timings / checksums (torch.distributed, gloo or RCCL).
"""


def frame_range(n_frames: int, rank: int, world: int):
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard arguments")
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def all_ranges(n_frames: int, world: int):
    return [frame_range(n_frames, r, world) for r in range(world)]
