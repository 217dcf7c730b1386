"""Slices by runs of pixels (RT_HIP_FLAG_PIXEL_RUNS), on a real MI355X, through the C ABI.

With the flag a slice traces every sample of its pixels (runs of 64 pixels dealt out to the slices in turn) and
leaves the sums of the others zero, so the slices' planes add up to the one-GPU frame bit for bit whatever the
order of the addition: every pixel's sum comes from one slice.  That is the bar here: integer equality of the
float bit patterns, and equal counters.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # demo, spheres, width, height, samples in a row, flags: ragged sizes, every accumulate kernel (S = 2, 10, 16, 32
    # through LDS, 3 lane per pixel), the tiled kernel (301 spheres) and the grid
    (1, 0, 161, 97, 16, 0),
    (2, 0, 200, 120, 10, 0),
    (1, 0, 333, 77, 3, 0),
    (1, 0, 128, 64, 32, 0),
    (1, 0, 40, 30, 2, 0),
    (3, 300, 96, 64, 4, 0),
    (3, 3000, 64, 48, 2, "grid"),
]


def planes_of(hip, scene, p, w, h):
    import torch

    planes = torch.full((3 * w * h,), float("nan"), dtype=torch.float32, device="cuda")  # whatever was there before
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    hip.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return planes.cpu().numpy(), stats[:4].cpu().numpy()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "demo%d_n%d_%dx%d_s%d_%s" % c)
@pytest.mark.parametrize("slices", [2, 3, 8])
def test_pixel_run_slices_add_up_bit_for_bit(hip, ffi, scenes, case, slices):
    from rt_amd import sharding

    demo, spheres, w, h, s, grid = case
    flags = ffi.FLAG_GRID if grid else 0
    scene = scenes(demo, 1, spheres or 10000)
    whole, counters = planes_of(hip, scene, ffi.params(w, h, s, seed=5, flags=flags), w, h)
    total = np.zeros_like(whole)
    counted = np.zeros(4, dtype=np.int64)
    owner = np.array([sharding.run_owner(p, slices) for p in range(w * h)])
    for rank in range(slices):
        p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=5, flags=flags), rank, slices, pixel_runs=True)
        assert p.flags & ffi.FLAG_PIXEL_RUNS
        mine, c = planes_of(hip, scene, p, w, h)
        assert np.all(mine.reshape(3, -1)[:, owner != rank] == 0.0), "pixels of other slices must be zero sums"
        total += mine
        counted += c
    assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))
    assert tuple(counted) == tuple(counters)


def test_headline_frame_in_eight_pixel_run_slices(hip, ffi, scenes):
    """The 1920x1080 16x16 frame as eight ranks would trace it, through the host path too: the slices' images (already
    divided by the sample count) add up to the frame's image bit for bit."""
    from rt_amd import sharding

    scene = scenes(1)
    w, h, s = 1920, 1080, 16
    whole, sw = hip.render(scene, ffi.params(w, h, s, seed=1))
    total = np.zeros_like(whole)
    samples = 0
    for rank in range(8):
        p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1), rank, 8, pixel_runs=True)
        img, st = hip.render(scene, p)
        total += img
        samples += st.samples
    assert samples == sw.samples == w * h * s * s
    assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))


def test_one_slice_ignores_the_flag(hip, ffi, scenes):
    scene = scenes(1)
    a, sa = hip.render(scene, ffi.params(96, 64, 4, seed=2))
    b, sb = hip.render(scene, ffi.params(96, 64, 4, seed=2, flags=ffi.FLAG_PIXEL_RUNS))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa.samples == sb.samples


def test_more_slices_than_runs(hip, ffi, scenes):
    """A frame of two runs dealt to five slices: three of them have nothing to trace and return zero sums."""
    from rt_amd import sharding

    scene = scenes(1)
    w, h, s = 16, 8, 4  # 128 pixels: two runs
    whole, _ = planes_of(hip, scene, ffi.params(w, h, s, seed=9), w, h)
    total = np.zeros_like(whole)
    for rank in range(5):
        p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=9), rank, 5, pixel_runs=True)
        mine, c = planes_of(hip, scene, p, w, h)
        if rank >= 2:
            assert not mine.any() and c[0] == 0
        total += mine
    assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))


def test_gather_of_each_slices_runs_rebuilds_the_frame(hip, ffi, scenes):
    """What bench.py's ranks exchange: every rank packs the sums of its own runs (rt_amd/sharding.py RunGather), the root
    puts the packed runs where they belong.  The frame on the root is the one-GPU frame bit for bit, for a frame
    whose runs do not divide evenly and whose last run is short."""
    import torch

    from rt_amd import sharding

    scene = scenes(1)
    w, h, s, world = 250, 131, 4, 3
    n = w * h
    stream = torch.cuda.current_stream().cuda_stream
    whole = torch.empty(3 * n, dtype=torch.float32, device="cuda")
    hip.trace_device(scene, ffi.params(w, h, s, seed=7), whole.data_ptr(), None, stream)
    slices, packed = [], []
    for rank in range(world):
        p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=7), rank, world, pixel_runs=True)
        planes = torch.full((3 * n,), float("nan"), dtype=torch.float32, device="cuda")
        hip.trace_device(scene, p, planes.data_ptr(), None, stream)
        torch.cuda.synchronize()
        slices.append(planes)
        g = sharding.RunGather(n, world, rank, "cuda")
        packed.append(g.pack(planes))
        assert g.bytes_per_rank <= 12 * (n // world + 64)
    root = sharding.RunGather(n, world, 0, "cuda")
    root.unpack(slices[0], torch.stack(packed))  # what the gather delivers: world x 3 x count
    assert torch.equal(slices[0], whole)


def test_the_launches_of_the_eight_rank_benchmark_gather_to_the_frame(hip, ffi, scenes):
    """What each rank of `bench.py --gpus 8` launches -- its runs of the 1920x1080 16x16 frame as one of a stream of
    frames, with block slots left free for the collective's kernels (RT_HIP_FLAG_PIPELINED | RT_HIP_FLAG_LEAVE_ROOM) --
    and what the root makes of the ranks' packed runs: the one-GPU frame, bit for bit, counters summing to its
    counters.  The launch is the one the slice timings quote: 960 blocks, pixel sums by the waves."""
    import torch

    from rt_amd import sharding

    scene = scenes(1)
    w, h, s, world = 1920, 1080, 16, 8
    n = w * h
    stream = torch.cuda.current_stream().cuda_stream
    whole = torch.empty(3 * n, dtype=torch.float32, device="cuda")
    whole_stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    hip.trace_device(scene, ffi.params(w, h, s, seed=1), whole.data_ptr(), whole_stats.data_ptr(), stream)
    flags = ffi.FLAG_PIPELINED | ffi.FLAG_LEAVE_ROOM
    first, packed = None, []
    counted = torch.zeros(4, dtype=torch.int64)
    for rank in range(world):
        p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1, flags=flags), rank, world, pixel_runs=True)
        plan = hip.describe(scene, p)
        assert (plan.blocks, plan.wave_sums, plan.kernel_sums) == (960, 1, 0)
        planes = torch.full((3 * n,), float("nan"), dtype=torch.float32, device="cuda")
        stats = torch.zeros(16, dtype=torch.int64, device="cuda")
        hip.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream)
        torch.cuda.synchronize()
        counted += stats[:4].cpu()
        packed.append(sharding.RunGather(n, world, rank, "cuda").pack(planes).clone())
        if rank == 0:
            first = planes
    sharding.RunGather(n, world, 0, "cuda").unpack(first, torch.stack(packed))
    assert torch.equal(first, whole)
    assert counted.tolist() == whole_stats[:4].cpu().tolist()
