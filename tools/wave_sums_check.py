"""Pixel sums taken by the waves of the trace kernel as they run out of work (the default where it applies) against the
fold and accumulate passes (RT_HIP_NO_WAVE_SUMS=1): the planes of rt_hip_trace_device must be identical bit for bit,
counters included, on whole frames, slices by samples and by runs of pixels; then the two are timed (trace kernel
alone, the whole call, and a stream of frames on two streams).

python tools/wave_sums_check.py [quick | time | fuzz SECONDS [SEED]]
fuzz: random frames large enough to fill the chip (so that the waves take the sums without being told to), random
sample counts, slices by samples and by runs, both demo scenes and a 100-sphere scene, until the time is up.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rt_amd import ffi, sharding  # noqa: E402

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
timing_only = len(sys.argv) > 1 and sys.argv[1] == "time"
r = ffi.HipRenderer(0)
stream = torch.cuda.current_stream().cuda_stream


def passes(on):
    if on:
        os.environ["RT_HIP_NO_WAVE_SUMS"] = "1"
    else:
        os.environ.pop("RT_HIP_NO_WAVE_SUMS", None)


def planes_of(scene, p, with_passes):
    passes(with_passes)
    n = p.width * p.height
    planes = torch.full((3 * n,), -7.0, dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    r.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream)
    torch.cuda.synchronize()
    return planes, stats[:11].cpu().tolist()


cases = [
    ("pretty 96x72 2x2", 1, 10000, dict(w=96, h=72, s=2), {}),
    ("pretty 97x53 4x4 (ragged)", 1, 10000, dict(w=97, h=53, s=4), {}),
    ("pretty 64x40 1x1", 1, 10000, dict(w=64, h=40, s=1), {}),
    ("glass 80x60 64x64", 2, 10000, dict(w=80, h=60, s=64), {}),
    ("glass 200x120 8x8, slice 1 of 3 by samples", 2, 10000, dict(w=200, h=120, s=8), dict(rank=1, world=3, runs=False)),
    ("pretty 333x200 16x16, slice 2 of 8 by runs", 1, 10000, dict(w=333, h=200, s=16), dict(rank=2, world=8, runs=True)),
    ("pretty 31x17 16x16, slice 7 of 8 by runs (more slices than runs of its own)", 1, 10000, dict(w=31, h=17, s=16), dict(rank=7, world=8, runs=True)),
    ("200 spheres 320x240 4x4", 3, 200, dict(w=320, h=240, s=4), {}),
    ("pretty 1024x768 2x2", 1, 10000, dict(w=1024, h=768, s=2), {}),
    ("pretty 1920x1080 16x16", 1, 10000, dict(w=1920, h=1080, s=16), {}),
    ("pretty 1920x1080 16x16, slice 3 of 8 by runs", 1, 10000, dict(w=1920, h=1080, s=16), dict(rank=3, world=8, runs=True)),
    ("glass 1024x768 32x32", 2, 10000, dict(w=1024, h=768, s=32), {}),
    ("glass 1024x768 10x10", 2, 10000, dict(w=1024, h=768, s=10), {}),
    ("pretty 1920x1080 12x12, slice 1 of 4 by runs", 1, 10000, dict(w=1920, h=1080, s=12), dict(rank=1, world=4, runs=True)),
]
if quick:
    cases = cases[:8]
if timing_only or (len(sys.argv) > 1 and sys.argv[1] == "fuzz"):
    cases = []
bad = 0
for name, demo, spheres, fr, sl in cases:
    scene = ffi.Scene.demo(demo, 1, spheres)
    # (marked as frames of a stream: where a reservation has to be larger than usual to hold whole pixels, only those
    # take the waves' sums)
    mk = lambda: ffi.params(fr["w"], fr["h"], fr["s"], seed=5, flags=fr.get("flags", 0) | ffi.FLAG_PIPELINED)  # noqa: E731
    p = sharding.slice_params(mk, sl["rank"], sl["world"], pixel_runs=sl["runs"]) if sl else mk()
    a, sa = planes_of(scene, p, False)
    b, sb = planes_of(scene, p, True)
    same = torch.equal(a, b)
    counters = sa[:4] == sb[:4]
    print("%-76s planes %s, counters %s, mean %.6f" % (name, "identical" if same else "DIFFER", "equal" if counters else "DIFFER",
                                                    float(a.mean())), flush=True)
    if not same:
        d = (a != b).nonzero().flatten()
        print("   %d of %d values differ, first at %s: %s against %s" % (d.numel(), a.numel(), d[:6].tolist(), a[d[:6]].tolist(), b[d[:6]].tolist()))
    bad += (not same) + (not counters)

if len(sys.argv) > 2 and sys.argv[1] == "fuzz":
    import numpy as np

    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    scenes = [ffi.Scene.demo(1, 1), ffi.Scene.demo(2, 1), ffi.Scene.demo(3, 1, 100)]
    t_end = time.time() + float(sys.argv[2])
    n = taken = 0
    while time.time() < t_end:
        scene = scenes[int(rng.integers(len(scenes)))]
        s = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32, 40, 64]))
        w, h = int(rng.integers(300, 2200)), int(rng.integers(200, 1300))
        if w * h * s * s > 600e6:
            continue
        world = int(rng.choice([1, 1, 2, 3, 8]))
        rank = int(rng.integers(world))
        by_runs = bool(rng.integers(2))
        mk = lambda: ffi.params(w, h, s, seed=int(rng.integers(1, 1000)), flags=ffi.FLAG_PIPELINED)  # noqa: E731
        p = sharding.slice_params(mk, rank, world, pixel_runs=by_runs) if world > 1 else mk()
        passes(False)
        took = r.describe(scene, p).wave_sums
        a, sa = planes_of(scene, p, False)
        b, sb = planes_of(scene, p, True)
        n += 1
        if n % 200 == 0:
            print("%d frames so far" % n, flush=True)  # (a run that prints nothing for minutes is taken for hung)
        taken += took
        if not torch.equal(a, b) or sa[:4] != sb[:4]:
            print("DIFFER: %dx%d s=%d slice %d/%d %s" % (w, h, s, rank, world, "runs" if by_runs else "samples"))
            bad += 1
    print("%d frames, %d of them summed by the waves" % (n, taken))
    r.close()
    print("FAILED: %d" % bad if bad else "all identical")
    sys.exit(1 if bad else 0)

if not quick:
    streams = [torch.cuda.Stream(priority=-(i % 2)) for i in range(2)]
    for name, demo, fr, sl in (("pretty 1920x1080 16x16", 1, dict(w=1920, h=1080, s=16), None),
                               ("pretty 1920x1080 16x16, 1/8 by runs", 1, dict(w=1920, h=1080, s=16), dict(rank=0, world=8, runs=True)),
                               ("pretty 1920x1080 16x16, 1/4 by runs", 1, dict(w=1920, h=1080, s=16), dict(rank=0, world=4, runs=True)),
                               ("pretty 1024x768 2x2", 1, dict(w=1024, h=768, s=2), None),
                               ("glass 1024x768 10x10", 2, dict(w=1024, h=768, s=10), None),
                               ("glass 1024x768 32x32", 2, dict(w=1024, h=768, s=32), None)):
        scene = ffi.Scene.demo(demo, 1)
        mk = lambda: ffi.params(fr["w"], fr["h"], fr["s"], seed=1)  # noqa: E731
        p = sharding.slice_params(mk, sl["rank"], sl["world"], pixel_runs=sl["runs"]) if sl else mk()
        n = p.width * p.height
        planes = [torch.empty(3 * n, dtype=torch.float32, device="cuda") for _ in range(2)]
        for with_passes in (False, True, False, True):
            passes(with_passes)
            kernel_ms, _ = r.time_trace_kernel(scene, p, 5)
            for _ in range(3):
                r.trace_device(scene, p, planes[0].data_ptr(), None, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                r.trace_device(scene, p, planes[0].data_ptr(), None, stream)
            torch.cuda.synchronize()
            call_ms = (time.perf_counter() - t0) / 10 * 1e3
            q = type(p).from_buffer_copy(p)
            q.flags |= ffi.FLAG_PIPELINED  # one of a stream of frames
            for k in range(4):
                r.trace_device(scene, q, planes[k % 2].data_ptr(), None, streams[k % 2].cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(16):
                r.trace_device(scene, q, planes[k % 2].data_ptr(), None, streams[k % 2].cuda_stream)
            torch.cuda.synchronize()
            stream_ms = (time.perf_counter() - t0) / 16 * 1e3
            print("%-44s %-10s trace kernel (+ fold) %.3f ms, call %.3f ms, in a stream of frames %.3f ms" % (
                name, "passes" if with_passes else "wave sums", kernel_ms, call_ms, stream_ms), flush=True)
r.close()
print("FAILED: %d" % bad if bad else "all identical")
sys.exit(1 if bad else 0)
