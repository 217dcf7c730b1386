"""Sharding of one frame over several processes (one per GPU) and the reduction that puts it back together.

The samples of a frame, numbered (pixel, sx, sy) in the order Frame::ForEachRayImpl visits them
(ray_tracing/frame.cpp:44-66), are dealt out to the slices in turn: sample (p * S + sx) * S + sy belongs to slice
number (that index) % slice_count -- or, with pixel_runs, the pixels are: run number (p // 64) belongs to slice
(that number) % slice_count, with all of its samples.  Every rank traces its slice into planar sums; one collective to
rank 0 (RCCL over xGMI when the tensors live on GPUs, gloo on CPUs) rebuilds the frame's sums, which rank 0 then
normalises: a sum-reduce of the whole planes for slices by samples (every rank holds a part of every pixel), a gather
of each rank's own runs for slices by runs of pixels (every pixel comes from one rank: 1 / slice_count of the bytes,
and no arithmetic).  The reference has no counterpart: its CUDA backend is single-GPU (rt_cuda/rt_cuda.cu:80).
"""
import contextlib

import torch



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


class RunGather:
    """The collective of slices by runs of pixels: every rank sends the sums of its own runs, packed, to `root`, which
    puts them where they belong in its planes.  planes: 3 x n_pixels floats (R, G, B planes); rank r owns the pixels
    p with (p // run_pixels) % world_size == r.  Nothing is added, so the frame on the root holds every pixel exactly
    as the rank that traced it wrote it.

        g = RunGather(n_pixels, world_size, rank, device)
        handle = g.start(dist, planes)        # packs this rank's runs and starts the gather (asynchronous)
        g.finish(handle, planes)              # waits; on the root: the other ranks' runs go into `planes`
    """

    def __init__(self, n_pixels, world_size, rank, device="cpu", root=0, run_pixels=64):
        self.n, self.world, self.rank, self.root = n_pixels, max(world_size, 1), rank, root
        pixel = torch.arange(n_pixels, device=device)
        owner = (pixel // run_pixels) % self.world
        counts = [int((owner == r).sum().item()) for r in range(self.world)]
        self.count = max(counts) if counts else 0  # every rank sends this many pixels (the rest of a short share: padding)
        self.own = pixel[owner == rank]
        self.bytes_per_rank = 3 * 4 * self.count
        # On the root, one pair of index vectors for all the other ranks' pixels: element (r, plane, j) of what was
        # received, flat, goes to element (plane, pixel) of the planes, flat -- the whole unpacking is one indexed
        # read and one indexed write, whatever the number of ranks.
        self.src = self.dst = None
        if rank == root and self.world > 1:
            src, dst = [], []
            for r in range(self.world):
                if r == root:
                    continue
                theirs = pixel[owner == r]
                j = torch.arange(theirs.numel(), device=device)
                for plane in range(3):
                    src.append((r * 3 + plane) * self.count + j)
                    dst.append(plane * n_pixels + theirs)
            self.src, self.dst = torch.cat(src), torch.cat(dst)

    def pack(self, planes):
        """This rank's own pixels, 3 x count (the rest of a short share is padding that nobody reads)."""
        view = planes.view(3, self.n)
        if self.own.numel() == self.count:
            packed = planes.new_empty((3, self.count))
            torch.index_select(view, 1, self.own, out=packed)
        else:
            packed = planes.new_zeros((3, self.count))
            packed[:, : self.own.numel()] = view[:, self.own]
        return packed

    def unpack(self, planes, received):
        """On the root: received[r] (what rank r packed; world x 3 x count, contiguous) goes where rank r's pixels
        belong."""
        if self.dst is not None and self.dst.numel():
            planes.view(-1).index_copy_(0, self.dst, received.view(-1).index_select(0, self.src))
        return planes

    def start(self, dist, planes):
        if self.world == 1:
            return None
        packed = self.pack(planes)
        received = planes.new_empty((self.world, 3, self.count)) if self.rank == self.root else None
        work = dist.gather(packed, [received[r] for r in range(self.world)] if received is not None else None,
                           dst=self.root, async_op=True)
        return work, packed, received

    def finish(self, handle, planes):
        if handle is None:
            return planes
        work, _, received = handle
        work.wait()  # on GPUs: makes the current stream wait for the collective
        if self.rank == self.root:
            self.unpack(planes, received)
        return planes


def gather_runs_to_root(dist, planes, n_pixels, world_size, rank, root=0):
    """One frame's slices by runs of pixels put together on `root` (in place); a no-op for one rank."""
    g = RunGather(n_pixels, world_size, rank, planes.device, root)
    return g.finish(g.start(dist, planes), planes)


class ReducePipeline:
    """The collective of successive frames to rank `root` (a sum-reduce; with `gather`, a RunGather, the gather of
    slices by runs of pixels instead), one frame behind the producer: while the planes of frame
    k are being reduced, the caller already fills the other buffer with frame k + 1.

        buf = pipe.acquire(k)      # the buffer frame k goes into (its previous reduce has been waited for)
        ... fill buf ...           # on GPUs: enqueue the trace on the current stream
        pipe.submit(k)             # starts the reduce of frame k, then retires frame k - 1
        pipe.drain()               # retires what is still in flight

    Retiring a frame waits for its reduce and, on the root, calls on_done(k, buffer) with the summed planes.
    With one rank there is nothing to reduce and submit() retires the frame at once.  The collectives are
    issued in frame order by every rank, as torch.distributed requires.
    """

    def __init__(self, dist, world_size, buffers, on_done=None, root=0, rank=0, context=None, gather=None):
        assert len(buffers) >= 2
        self.dist, self.world, self.buffers, self.on_done, self.root, self.rank = (
            dist, max(world_size, 1), buffers, on_done, root, rank)
        self.gather = gather
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
        if self.world > 1 and self.gather is not None:
            work = self.gather.start(self.dist, buf)
        elif self.world > 1:
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
            if work is not None and self.gather is not None:
                self.gather.finish(work, self.buffers[k % len(self.buffers)])
            elif work is not None:
                work.wait()  # on GPUs: makes the current stream wait for the collective
            if self.on_done is not None and self.rank == self.root:
                self.on_done(k, self.buffers[k % len(self.buffers)])
