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
