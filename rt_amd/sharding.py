"""Sharding of one frame over several processes (one per GPU) and the reduction that puts it back together.

The samples of a frame, numbered (pixel, sx, sy) in the order Frame::ForEachRayImpl visits them
(ray_tracing/frame.cpp:44-66), are dealt out to the slices in turn: sample (p * S + sx) * S + sy belongs to slice
number (that index) % slice_count.  Every rank traces its slice into planar sums; one sum-reduce to rank 0
(RCCL over xGMI when the tensors live on GPUs, gloo on CPUs) rebuilds the frame's sums, which rank 0 then
normalises.  The reference has no counterpart: its CUDA backend is single-GPU (rt_cuda/rt_cuda.cu:80).
"""


def owner_of(pixel, sx, sy, samples_in_row, slice_count):
    """Slice that owns sample (sx, sy) of a pixel."""
    return ((pixel * samples_in_row + sx) * samples_in_row + sy) % max(slice_count, 1)


def slice_params(make_params, rank, world_size):
    """Trace parameters of `rank`: the caller's parameters restricted to its slice."""
    p = make_params()
    p.slice_index = rank % max(world_size, 1)
    p.slice_count = max(world_size, 1)
    return p


def reduce_to_root(dist, tensor, world_size, root=0):
    """Sums every rank's planar sums into rank `root`'s tensor (in place); a no-op for one rank."""
    if world_size > 1:
        dist.reduce(tensor, dst=root, op=dist.ReduceOp.SUM)
    return tensor
