"""Sharding of one frame over several processes (one per GPU) and the reduction that puts it back together.

The samples of a frame, numbered (pixel, sx, sy) in the order Frame::ForEachRayImpl visits them
(ray_tracing/frame.cpp:44-66), are dealt out to the slices in turn: sample (p * S + sx) * S + sy belongs to slice
number (that index) % slice_count -- or, with pixel_runs, the pixels are: run number (p // 64) belongs to slice
(that number) % slice_count, with all of its samples.  Every rank traces its slice into planar sums; one sum-reduce to rank 0
(RCCL over xGMI when the tensors live on GPUs, gloo on CPUs) rebuilds the frame's sums, which rank 0 then
normalises.  The reference has no counterpart: its CUDA backend is single-GPU (rt_cuda/rt_cuda.cu:80).
"""
import contextlib



def owner_of(pixel, sx, sy, samples_in_row, slice_count):
    """Slice that owns sample (sx, sy) of a pixel."""
    return ((pixel * samples_in_row + sx) * samples_in_row + sy) % max(slice_count, 1)


def slice_params(make_params, rank, world_size, pixel_runs=False):
    """Trace parameters of `rank`: the caller's parameters restricted to its slice.  pixel_runs: the slices divide
    the frame by runs of 64 pixels (RT_HIP_FLAG_PIXEL_RUNS: a rank traces every sample of its pixels, and the
    reduced frame equals the one-GPU frame bit for bit) instead of by samples."""
    p = make_params()
    p.slice_index = rank % max(world_size, 1)
    p.slice_count = max(world_size, 1)
    if pixel_runs and world_size > 1:
        p.flags |= FLAG_PIXEL_RUNS
    return p


FLAG_PIXEL_RUNS = 16  # include/rt_hip.h RT_HIP_FLAG_PIXEL_RUNS


def run_owner(pixel, slice_count, run_pixels=64):
    """Slice that owns `pixel` when the frame is divided by runs of pixels."""
    return (pixel // run_pixels) % max(slice_count, 1)


def reduce_to_root(dist, tensor, world_size, root=0):
    """Sums every rank's planar sums into rank `root`'s tensor (in place); a no-op for one rank."""
    if world_size > 1:
        dist.reduce(tensor, dst=root, op=dist.ReduceOp.SUM)
    return tensor


class ReducePipeline:
    """Sum-reduce of successive frames to rank `root`, one frame behind the producer: while the planes of frame
    k are being reduced, the caller already fills the other buffer with frame k + 1.

        buf = pipe.acquire(k)      # the buffer frame k goes into (its previous reduce has been waited for)
        ... fill buf ...           # on GPUs: enqueue the trace on the current stream
        pipe.submit(k)             # starts the reduce of frame k, then retires frame k - 1
        pipe.drain()               # retires what is still in flight

    Retiring a frame waits for its reduce and, on the root, calls on_done(k, buffer) with the summed planes.
    With one rank there is nothing to reduce and submit() retires the frame at once.  The collectives are
    issued in frame order by every rank, as torch.distributed requires.
    """

    def __init__(self, dist, world_size, buffers, on_done=None, root=0, rank=0, context=None):
        assert len(buffers) >= 2
        self.dist, self.world, self.buffers, self.on_done, self.root, self.rank = (
            dist, max(world_size, 1), buffers, on_done, root, rank)
        # context(k): what frame k is retired under (on GPUs: the stream that owns the frame's buffer, so that
        # the wait and on_done are ordered before the next use of that buffer); submit(k) is the caller's to wrap
        self.context = context if context is not None else (lambda k: contextlib.nullcontext())
        self.pending = {}  # frame -> work handle (None when there was nothing to wait for)

    def acquire(self, k):
        older = k - len(self.buffers)
        if older in self.pending:
            self._retire(older)
        return self.buffers[k % len(self.buffers)]

    def submit(self, k):
        buf = self.buffers[k % len(self.buffers)]
        work = None
        if self.world > 1:
            work = self.dist.reduce(buf, dst=self.root, op=self.dist.ReduceOp.SUM, async_op=True)
        self.pending[k] = work
        if self.world == 1:
            self._retire(k)
        elif (k - 1) in self.pending:
            self._retire(k - 1)

    def drain(self):
        for k in sorted(self.pending):
            self._retire(k)

    def _retire(self, k):
        work = self.pending.pop(k)
        with self.context(k):
            if work is not None:
                work.wait()  # on GPUs: makes the current stream wait for the collective
            if self.on_done is not None and self.rank == self.root:
                self.on_done(k, self.buffers[k % len(self.buffers)])
