"""RT_HIP_FLAG_KERNEL_SUMS: the trace kernel adds up the pixels itself (column sums of the queue windows in LDS, 12
bytes per pixel to memory) instead of leaving 12 bytes per sub-sample column for the accumulate pass.  The additions
and their order are the same, so planes, images and counters must be identical bit for bit, whatever the scene's
kernel (resident, LDS tiles, grid walk), the slicing and the end-of-queue sample sharing do.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def planes_and_counters(hip, ffi, scene, p, kernel_sums):
    import torch

    q = type(p).from_buffer_copy(p)
    q.flags = (q.flags | ffi.FLAG_KERNEL_SUMS) if kernel_sums else (q.flags & ~ffi.FLAG_KERNEL_SUMS)
    planes = torch.full((3 * q.width * q.height,), float("nan"), dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    hip.trace_device(scene, q, planes.data_ptr(), stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return planes, stats.cpu().tolist()


CASES = [
    # demo, spheres, width, height, samples_in_row, flags, (slice_index, slice_count, by runs)
    (1, 0, 96, 72, 2, 0, None),
    (1, 0, 97, 53, 4, 0, None),                  # ragged: the last window of the queue is short
    (1, 0, 64, 40, 1, 0, None),                  # a pixel per column
    (2, 0, 80, 60, 64, 0, None),                 # a pixel per window
    (2, 0, 200, 120, 8, 0, (1, 3, False)),       # slice by samples: every third sample of every column
    (1, 0, 333, 200, 16, 0, (2, 8, True)),       # slice by runs of 64 pixels
    (1, 0, 31, 17, 32, 0, (5, 7, True)),         # more slices than some of them have runs
    (3, 100, 320, 240, 4, 0, None),              # 101 spheres: resident, primary rays filtered per sample
    (3, 200, 320, 240, 4, "grid", None),         # the same through the uniform grid
    (3, 3000, 160, 120, 2, 0, None),             # LDS tiles: the block's waves move in step
    (1, 0, 1024, 768, 2, 0, None),               # the reference's default window
    (2, 0, 640, 480, 16, 0, None),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "demo%d_n%d_%dx%d_s%d_%s_%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6]))
def test_kernel_sums_change_no_bit(hip, ffi, scenes, case):
    from rt_amd import sharding

    demo, spheres, w, h, s, flags, sl = case
    scene = scenes(demo, 1, spheres) if spheres else scenes(demo)
    fl = ffi.FLAG_GRID if flags == "grid" else 0
    make = lambda: ffi.params(w, h, s, seed=5, flags=fl)  # noqa: E731
    p = sharding.slice_params(make, sl[0], sl[1], pixel_runs=sl[2]) if sl else make()
    with_flag = type(p).from_buffer_copy(p)
    with_flag.flags |= ffi.FLAG_KERNEL_SUMS
    assert hip.describe(scene, with_flag).kernel_sums == 1 and hip.describe(scene, p).kernel_sums == 0
    a, sa = planes_and_counters(hip, ffi, scene, p, True)
    b, sb = planes_and_counters(hip, ffi, scene, p, False)
    assert sa[10] == 0 and sb[10] == 0  # no wave gave up waiting for a window slot
    assert sa[:4] == sb[:4]
    assert bool((a == b).all())


def test_kernel_sums_ignored_where_a_window_holds_no_whole_pixels(hip, ffi, scenes):
    """10 samples per row (the viewer's high-quality mode): the flag is accepted and the frame is traced as without."""
    scene = scenes(2)
    p = ffi.params(120, 90, 10, seed=3)
    assert hip.describe(scene, ffi.params(120, 90, 10, seed=3, flags=ffi.FLAG_KERNEL_SUMS)).kernel_sums == 0
    a, sa = planes_and_counters(hip, ffi, scene, p, True)
    b, sb = planes_and_counters(hip, ffi, scene, p, False)
    assert sa[:4] == sb[:4] and bool((a == b).all())


def test_kernel_sums_through_the_job_api(hip, oracle, ffi, scenes):
    """Whole frames through rt_hip_trace_async / finish, with a snapshot on the way: the image is the one the
    accumulate pass gives, and within the usual bar of the CPU path."""
    scene = scenes(1)
    plain, sp = hip.render(scene, ffi.params(320, 240, 4, seed=8))
    p = ffi.params(320, 240, 4, seed=8, flags=ffi.FLAG_KERNEL_SUMS)
    job = hip.start(scene, p)
    snap = hip.snapshot(job, p)
    out, st = hip.finish(job, p)
    np.testing.assert_array_equal(out, plain)
    assert (st.samples, st.ray_queries, st.hit_queries, st.shaded_bounces) == (sp.samples, sp.ray_queries, sp.hit_queries, sp.shaded_bounces)
    # a snapshot shows finished pixels as they will be and unfinished ones black (a pixel's three sums are written
    # one after the other, so one being written right now may show some of them)
    assert ((snap == 0.0) | (snap == out)).all()
    # deterministic mode against the oracle, per pixel
    pd = ffi.params(160, 120, 2, flags=ffi.FLAG_DETERMINISTIC | ffi.FLAG_KERNEL_SUMS)
    ref, rs = oracle.render(scene, pd)
    out, st = hip.render(scene, pd)
    rel = np.abs(ref - out) / np.maximum(np.abs(ref), 1e-3)
    assert rel.max() < 1e-4
    assert (st.samples, st.ray_queries, st.hit_queries, st.shaded_bounces) == (rs.samples, rs.ray_queries, rs.hit_queries, rs.shaded_bounces)


def test_kernel_sums_cancelled_job_leaves_whole_pixels(hip, ffi, scenes):
    """A cancelled frame: every pixel is either complete (as in the finished frame) or still black."""
    scene = scenes(2)
    p = ffi.params(1920, 1080, 16, seed=2, flags=ffi.FLAG_KERNEL_SUMS)
    whole, _ = hip.render(scene, p)
    job = hip.start(scene, p)
    hip.cancel(job)
    part, _ = hip.finish(job, p, allow_cancelled=True)
    assert ((part == 0.0) | (part == whole)).all()
