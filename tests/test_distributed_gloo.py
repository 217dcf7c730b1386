"""The multi-process path on CPUs: two ranks each render their slice, one gloo sum-reduce rebuilds the frame.

bench.py runs exactly this sequence on GPUs (slice_params -> trace -> reduce_to_root -> resolve) with the HIP
backend and RCCL; here the slice renderer is the host build of the kernel's state machine, so the sharding
rule, the reduction and the normalisation are covered without a GPU.
"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    for path in (ROOT, os.path.join(ROOT, "tests")):
        if path not in sys.path:
            sys.path.insert(0, path)
    import emu_ffi
    from rt_amd import ffi, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ffi.Scene.demo(1, 1)
    p = sharding.slice_params(lambda: ffi.params(48, 36, 3, seed=5), rank, world)
    assert (p.slice_index, p.slice_count) == (rank, world)
    rgb, stats = emu_ffi.render(scene, p)  # this slice's sums, already scaled by 1/S^2 (the scale is linear)
    planes = torch.from_numpy(rgb.astype(np.float32))
    samples = torch.tensor([stats["samples"]], dtype=torch.int64)
    sharding.reduce_to_root(dist, planes, world)
    dist.reduce(samples, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, planes.numpy())
        assert int(samples.item()) == 48 * 36 * 9  # every sample traced by exactly one rank
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_rebuild_the_frame(tmp_path, oracle, ffi, scenes):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    combined = np.load(out_path)
    full, _ = oracle.render(scenes(1), ffi.params(48, 36, 3, seed=5), 4)
    assert np.abs(combined - full).mean() < 1e-5
    rel = np.abs(combined - full) / np.maximum(np.abs(full), 1e-3)
    assert (rel.max(axis=2) > 1e-3).mean() < 5e-3


def _pipeline_worker(rank, world, port, out_path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from rt_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    done = []
    buffers = [torch.zeros(1000), torch.zeros(1000)]
    pipe = sharding.ReducePipeline(dist, world, buffers, on_done=lambda k, b: done.append((k, b.clone())), rank=rank)
    for k in range(5):
        buf = pipe.acquire(k)
        buf.fill_(float((rank + 1) * (k + 1)))  # "the trace" of frame k
        pipe.submit(k)
        assert len(pipe.pending) <= 1  # at most the frame just submitted is in flight
    pipe.drain()
    assert not pipe.pending
    if rank == 0:
        assert [k for k, _ in done] == [0, 1, 2, 3, 4]
        torch.save(torch.stack([b for _, b in done]), out_path)
    else:
        assert done == []
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_pipeline_keeps_frames_apart(tmp_path):
    """bench.py's double-buffered reduce: every frame arrives once, in order, summed over the ranks, although
    frame k + 1 is produced while frame k is still being reduced."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "frames.pt")
    mp.spawn(_pipeline_worker, args=(2, port, out_path), nprocs=2, join=True)
    frames = torch.load(out_path, weights_only=True)
    for k in range(5):
        assert torch.all(frames[k] == 3.0 * (k + 1))


def test_reduce_pipeline_with_one_rank():
    from rt_amd import sharding

    done = []
    pipe = sharding.ReducePipeline(None, 1, [torch.zeros(4), torch.zeros(4)], on_done=lambda k, b: done.append((k, float(b[0]))))
    for k in range(3):
        pipe.acquire(k).fill_(k + 1.0)
        pipe.submit(k)
    pipe.drain()
    assert done == [(0, 1.0), (1, 2.0), (2, 3.0)]


def _gather_worker(rank, world, port, out_path):
    for path in (ROOT, os.path.join(ROOT, "tests")):
        if path not in sys.path:
            sys.path.insert(0, path)
    import emu_ffi
    from rt_amd import ffi, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ffi.Scene.demo(1, 1)
    w, h = 50, 37  # 1850 pixels: 29 runs of 64, the last one short, and an odd number of them for two ranks
    rgb, _ = emu_ffi.render(scene, ffi.params(w, h, 2, seed=5))
    whole = torch.from_numpy(np.ascontiguousarray(rgb.reshape(-1, 3).T.astype(np.float32))).reshape(-1)  # 3 planes
    # what a slice by runs of pixels looks like: this rank's pixels complete, the others' zero
    pixel = torch.arange(w * h)
    mine = torch.tensor([sharding.run_owner(int(p), world) == rank for p in pixel])
    planes = (whole.view(3, -1) * mine).reshape(-1).clone()
    frames = [planes.clone(), planes.clone()]
    # once directly ...
    sharding.gather_runs_to_root(dist, frames[0], w * h, world, rank)
    # ... and once through the double-buffered pipeline bench.py uses
    done = []
    g = sharding.RunGather(w * h, world, rank)
    pipe = sharding.ReducePipeline(dist, world, [frames[1], planes.clone()], on_done=lambda k, b: done.append(b.clone()),
                                   rank=rank, gather=g)
    pipe.acquire(0)
    pipe.submit(0)
    pipe.drain()
    if rank == 0:
        assert g.bytes_per_rank == 12 * (14 * 64 + 58)  # 15 of the 29 runs are rank 0's, the last one 58 pixels long
        assert torch.equal(frames[0], whole) and torch.equal(done[0], whole)  # bit for bit: nothing was added
        np.save(out_path, frames[0].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_their_runs_of_pixels(tmp_path):
    """Slices by runs of 64 pixels are put together by a gather of each rank's own runs, not by a sum: the frame on
    the root is the one-rank frame bit for bit, for a frame whose last run is short and whose runs do not divide
    evenly among the ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "gathered.npy")
    mp.spawn(_gather_worker, args=(2, port, out_path), nprocs=2, join=True)
    assert np.isfinite(np.load(out_path)).all()
