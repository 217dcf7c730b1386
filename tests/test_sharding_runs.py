"""Host logic of the slices by runs of pixels (rt_amd/sharding.py); no GPU needed."""
import numpy as np

from rt_amd import ffi, sharding


def test_runs_partition_the_pixels():
    for pixels, slices in [(1, 1), (63, 2), (64, 2), (65, 2), (1000, 3), (2073600, 8)]:
        owner = (np.arange(pixels) // 64) % slices
        assert all(sharding.run_owner(p, slices) == owner[p] for p in range(0, pixels, max(pixels // 997, 1)))
        counts = np.bincount(owner, minlength=slices)
        assert counts.sum() == pixels
        # whole runs differ by at most one between slices
        assert counts.max() - counts.min() <= 64


def test_slice_params_sets_the_flag_only_when_asked_and_sharded():
    make = lambda: ffi.params(48, 36, 3, seed=5)  # noqa: E731
    p = sharding.slice_params(make, 1, 4, pixel_runs=True)
    assert (p.slice_index, p.slice_count) == (1, 4) and p.flags & ffi.FLAG_PIXEL_RUNS
    assert sharding.FLAG_PIXEL_RUNS == ffi.FLAG_PIXEL_RUNS
    assert not sharding.slice_params(make, 1, 4).flags & ffi.FLAG_PIXEL_RUNS
    assert not sharding.slice_params(make, 0, 1, pixel_runs=True).flags & ffi.FLAG_PIXEL_RUNS


def test_header_declares_the_flag():
    import os

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rt_hip.h")).read()
    assert "#define RT_HIP_FLAG_PIXEL_RUNS 16u" in text


def test_every_item_is_in_exactly_one_slice_queue():
    """The kernels' own bookkeeping (rt_prep.h set_pixel_runs, rt_core.h queue_item, compiled for the host): over the
    slices, the queues hold every item of the frame once, each in the slice that owns its run, in increasing order."""
    import ctypes as C
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "tests", "emu", "libcore_emu.so"))
    lib.emu_pixel_run_queue.restype = C.c_uint32
    lib.emu_pixel_run_queue.argtypes = [C.c_uint32] * 5 + [C.c_void_p]
    for pixels, s, slices in [(1, 1, 2), (63, 2, 2), (64, 16, 2), (65, 3, 3), (1000, 10, 8), (128, 4, 5), (4321, 16, 7),
                              (20736, 16, 8)]:
        seen = np.zeros(pixels * s, dtype=np.int32)
        for rank in range(slices):
            items = np.zeros(pixels * s, dtype=np.uint32)
            n = lib.emu_pixel_run_queue(pixels, s, 64, rank, slices, items.ctypes.data)
            mine = items[:n].astype(np.int64)
            assert np.all(np.diff(mine) > 0)
            assert np.all(mine < pixels * s)
            assert np.all(((mine // s) // 64) % slices == rank)
            seen[mine] += 1
        assert np.all(seen == 1), (pixels, s, slices)
