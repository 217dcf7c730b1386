"""Pixel sums taken by the waves of the resident trace kernels as they run out of work (the default for launches that
fill the chip, where a few 64-column queue windows hold whole pixels) against the fold and accumulate passes
(RT_HIP_NO_WAVE_SUMS): the wave that wrote a pixel's column sums adds them in the order the passes would, so planes,
images and counters must be identical bit for bit.  Small test frames do not fill the chip; RT_HIP_LOG_ENTRIES makes
them take the path anyway, and with a small number it also fills the waves' reservation logs, after which a wave
takes no more work and leaves the rest of the queue to the others.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def planes_and_counters(hip, scene, p):
    import torch

    planes = torch.full((3 * p.width * p.height,), float("nan"), dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    hip.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return planes, stats.cpu().tolist()


def both_ways(hip, scene, p, monkeypatch, log_entries="384"):
    monkeypatch.delenv("RT_HIP_NO_WAVE_SUMS", raising=False)
    if log_entries is not None:
        monkeypatch.setenv("RT_HIP_LOG_ENTRIES", log_entries)
    assert hip.describe(scene, p).wave_sums == 1  # the path under test is the one that runs
    a, sa = planes_and_counters(hip, scene, p)
    monkeypatch.delenv("RT_HIP_LOG_ENTRIES", raising=False)
    monkeypatch.setenv("RT_HIP_NO_WAVE_SUMS", "1")
    assert hip.describe(scene, p).wave_sums == 0
    b, sb = planes_and_counters(hip, scene, p)
    monkeypatch.delenv("RT_HIP_NO_WAVE_SUMS")
    return a, sa, b, sb


CASES = [
    # demo, spheres, width, height, samples_in_row, (slice_index, slice_count, by runs), log entries
    (1, 0, 96, 72, 2, None, "384"),
    (1, 0, 97, 53, 4, None, "384"),                  # ragged: the last window of the queue is short
    (1, 0, 64, 40, 1, None, "384"),                  # a pixel per column
    (2, 0, 80, 60, 64, None, "384"),                 # a pixel per window, columns long enough to be shared out
    (2, 0, 200, 120, 8, (1, 3, False), "384"),       # slice by samples: every third sample of every column
    (1, 0, 333, 200, 16, (2, 8, True), "384"),       # slice by runs of 64 pixels: the other runs' sums are zero
    (1, 0, 31, 17, 32, (5, 7, True), "384"),         # more slices than some of them have runs
    (3, 100, 320, 240, 4, None, "384"),              # 101 spheres: resident, primary rays filtered per sample
    (2, 0, 120, 90, 10, None, "384"),                # the viewer's high-quality mode: a window holds 6.4 pixels, five windows 32
    (1, 0, 333, 77, 12, (1, 2, True), "384"),        # three windows hold 16 pixels; slices by runs of 64 pixels
    (1, 0, 150, 100, 3, None, "384"),
    (2, 0, 97, 53, 5, (2, 3, False), "384"),
    (2, 0, 100, 60, 20, None, "384"),
    (1, 0, 40, 30, 128, None, "384"),                # a pixel is two windows
    (1, 0, 640, 360, 10, None, "3"),                 # three reservations per wave: the logs fill up
    (1, 0, 640, 360, 16, None, "3"),
    (2, 0, 320, 240, 8, (0, 2, True), "2"),
    (1, 0, 1024, 768, 2, None, "384"),               # the reference's default window
    (2, 0, 1280, 720, 16, None, None),               # as a caller gets it: the launch fills the chip
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "demo%d_n%d_%dx%d_s%d_%s_log%s" % c)
def test_wave_sums_change_no_bit(hip, ffi, scenes, monkeypatch, case):
    from rt_amd import sharding

    demo, spheres, w, h, s, sl, log_entries = case
    scene = scenes(demo, 1, spheres) if spheres else scenes(demo)
    make = lambda: ffi.params(w, h, s, seed=5)  # noqa: E731
    p = sharding.slice_params(make, sl[0], sl[1], pixel_runs=sl[2]) if sl else make()
    a, sa, b, sb = both_ways(hip, scene, p, monkeypatch, log_entries)
    assert sa[:4] == sb[:4]
    assert bool((a == b).all())


def test_wave_sums_of_the_headline_frame_and_its_slices(hip, ffi, scenes, monkeypatch):
    """1920x1080, 16x16 samples per pixel, whole and as slice 3 of 8 by runs of pixels, as callers get them (nothing
    forced): the launches fill the chip, so the waves take the sums."""
    from rt_amd import sharding

    scene = scenes(1)
    make = lambda: ffi.params(1920, 1080, 16, seed=1)  # noqa: E731
    for p in (make(), sharding.slice_params(make, 3, 8, pixel_runs=True)):
        a, sa, b, sb = both_ways(hip, scene, p, monkeypatch, None)
        assert sa[:4] == sb[:4]
        assert bool((a == b).all())


def test_wave_sums_of_the_largest_configuration(hip, ffi, scenes, monkeypatch):
    """glass_spheres 3840x2160, 32x32 samples per pixel: 4 million queue windows, reservations of 14 windows so that
    the waves' logs hold them."""
    scene = scenes(2)
    p = ffi.params(3840, 2160, 32, seed=2)
    assert hip.describe(scene, p).batch_items > 256
    a, sa, b, sb = both_ways(hip, scene, p, monkeypatch, None)
    assert sa[0] == sb[0] == 3840 * 2160 * 32 * 32 and sa[:4] == sb[:4]
    assert bool((a == b).all())


def test_wave_sums_with_path_stacks_several_lights_and_library_math(hip, ffi, monkeypatch):
    """The other resident kernels: the reference's reverse loop over a stack (RT_HIP_FORCE_STACK), any number of
    lights, the compiler's IEEE sequences (RT_HIP_LIBRARY_MATH)."""
    def unit(d):
        d = np.asarray(d, dtype=np.float64)
        return tuple(d / np.linalg.norm(d))

    mats = [(0, (0.7, 0.6, 0.5), 0.7, 0.0), (1, (0.9, 0.8, 0.3), 0.1, 0.15), (2, (0.3, 0.9, 0.8), 0.0, 0.85)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0), (-1.2, -0.5, 2.0, 0.5, 1), (0.0, -0.4, 1.0, 0.6, 2), (1.3, -0.6, 2.5, 0.4, 0)]
    lights = [(unit((0.0, -1.0, 0.0)), (1.0, 0.5, 0.25)), (unit((0.6, -0.6, 0.5)), (0.2, 0.9, 0.2))]
    several = ffi.Scene.build(spheres, mats, lights, dict(position=(0.0, 0.5, -3.0), direction=(0.0, -0.2, 1.0)))
    monkeypatch.setenv("RT_HIP_FORCE_STACK", "1")
    stack = ffi.Scene.demo(1, 1)
    monkeypatch.delenv("RT_HIP_FORCE_STACK")
    p = ffi.params(200, 150, 8, seed=4)
    for scene in (several, stack):
        a, sa, b, sb = both_ways(hip, scene, p, monkeypatch)
        assert sa[:4] == sb[:4] and bool((a == b).all())
    monkeypatch.setenv("RT_HIP_LIBRARY_MATH", "1")
    for scene in (several, ffi.Scene.demo(2, 1)):
        a, sa, b, sb = both_ways(hip, scene, p, monkeypatch)
        assert sa[:4] == sb[:4] and bool((a == b).all())
    monkeypatch.delenv("RT_HIP_LIBRARY_MATH")


def test_wave_sums_through_the_job_api(hip, ffi, scenes, monkeypatch):
    """Whole frames through rt_hip_trace_async / finish with a snapshot on the way, and a job cancelled at once: the
    pixels of windows no wave reserved are zero, the others what the finished frame holds."""
    scene = scenes(1)
    p = ffi.params(320, 240, 4, seed=8)
    monkeypatch.setenv("RT_HIP_NO_WAVE_SUMS", "1")
    plain, sp = hip.render(scene, p)
    monkeypatch.delenv("RT_HIP_NO_WAVE_SUMS")
    monkeypatch.setenv("RT_HIP_LOG_ENTRIES", "384")
    job = hip.start(scene, p)
    snap = hip.snapshot(job, p)
    out, st = hip.finish(job, p)
    np.testing.assert_array_equal(out, plain)
    assert (st.samples, st.ray_queries, st.hit_queries, st.shaded_bounces) == (sp.samples, sp.ray_queries, sp.hit_queries, sp.shaded_bounces)
    assert np.isfinite(snap).all()
    big = ffi.params(1920, 1080, 16, seed=8)
    whole, _ = hip.render(scene, big)
    job = hip.start(scene, big)
    hip.cancel(job)
    part, _ = hip.finish(job, big, allow_cancelled=True)
    assert ((part == 0.0) | (part == whole)).all()


def test_which_frames_take_which_path(hip, ffi, scenes):
    """rt_hip_describe_launch: the waves take the sums for launches that fill the chip and whose windows hold whole
    pixels; small frames, sample counts that do not divide 64, tiled scenes and grid walks keep the passes."""
    pretty = scenes(1)
    headline = hip.describe(pretty, ffi.params(1920, 1080, 16))
    assert (headline.wave_sums, headline.kernel_sums, headline.blocks, headline.forward, headline.one_light,
            headline.column_masks, headline.sample_sharing) == (1, 0, 1024, 1, 1, 1, 1)
    assert headline.batch_items == 256 and headline.batch_min == 64
    stream = hip.describe(pretty, ffi.params(1920, 1080, 16, flags=ffi.FLAG_PIPELINED))
    assert stream.batch_min == 128 and stream.wave_sums == 1
    assert hip.describe(pretty, ffi.params(1024, 768, 2)).wave_sums == 0       # a third of the chip's block slots
    # 10x10: reservations of five windows hold 32 whole pixels; worth it in a stream of frames, not for a frame on its own
    hq = hip.describe(scenes(2), ffi.params(1024, 768, 10, flags=ffi.FLAG_PIPELINED))
    assert hq.wave_sums == 1 and hq.batch_items % 320 == 0 and hq.batch_min % 320 == 0
    assert hip.describe(scenes(2), ffi.params(1024, 768, 10)).wave_sums == 0
    assert hip.describe(pretty, ffi.params(1920, 1080, 9)).wave_sums == 0      # nine windows before a reservation holds whole pixels
    fused = hip.describe(pretty, ffi.params(1920, 1080, 16, flags=ffi.FLAG_KERNEL_SUMS))
    assert (fused.kernel_sums, fused.wave_sums) == (1, 0)
    # 201 spheres with as many materials: resident, but the waves' logs would cost a block per compute unit
    assert hip.describe(scenes(3, 1, 200), ffi.params(1920, 1080, 4)).wave_sums == 0
    big = scenes(3, 1, 3000)
    tiled = hip.describe(big, ffi.params(640, 360, 4))
    assert (tiled.tiled, tiled.grid, tiled.wave_sums) == (1, 0, 0)
    walked = hip.describe(big, ffi.params(640, 360, 4, flags=ffi.FLAG_GRID))
    assert (walked.tiled, walked.grid, walked.wave_sums) == (0, 1, 0)
    four_k = hip.describe(scenes(2), ffi.params(3840, 2160, 32))
    assert four_k.wave_sums == 1 and four_k.batch_items >= 256  # long reservations: the waves' logs hold them
