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
