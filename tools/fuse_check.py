"""Pixel sums taken inside the trace kernel (RT_HIP_FLAG_KERNEL_SUMS) against the column sums + accumulate pass: the
planes of rt_hip_trace_device must be identical bit for bit, counters included, on whole frames, slices by samples and by runs
of pixels, resident, tiled and grid scenes; then the two are timed (trace kernel alone, and the whole call).

python tools/fuse_check.py [quick | time]
Environment: RT_HIP_RING_SLOTS=n caps the windows a wave may keep unfinished (default: what fits, at most 8).
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


def with_sums(p, fused):
    q = type(p).from_buffer_copy(p)
    q.flags = (q.flags | ffi.FLAG_KERNEL_SUMS) if fused else (q.flags & ~ffi.FLAG_KERNEL_SUMS)
    return q


def planes_of(scene, p, fused):
    p = with_sums(p, fused)
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
    ("200 spheres 320x240 4x4", 3, 200, dict(w=320, h=240, s=4), {}),
    ("200 spheres 320x240 4x4 grid", 3, 200, dict(w=320, h=240, s=4, flags=ffi.FLAG_GRID), {}),
    ("3000 spheres 160x120 2x2 (tiled)", 3, 3000, dict(w=160, h=120, s=2), {}),
    ("pretty 1024x768 2x2", 1, 10000, dict(w=1024, h=768, s=2), {}),
    ("pretty 1920x1080 16x16", 1, 10000, dict(w=1920, h=1080, s=16), {}),
    ("pretty 1920x1080 16x16, slice 3 of 8 by runs", 1, 10000, dict(w=1920, h=1080, s=16), dict(rank=3, world=8, runs=True)),
    ("glass 1024x768 32x32", 2, 10000, dict(w=1024, h=768, s=32), {}),
]
if quick:
    cases = cases[:6]
if timing_only:
    cases = []
bad = 0
for name, demo, spheres, fr, sl in cases:
    scene = ffi.Scene.demo(demo, 1, spheres)
    mk = lambda: ffi.params(fr["w"], fr["h"], fr["s"], seed=5, flags=fr.get("flags", 0))  # noqa: E731
    p = sharding.slice_params(mk, sl["rank"], sl["world"], pixel_runs=sl["runs"]) if sl else mk()
    a, sa = planes_of(scene, p, True)
    b, sb = planes_of(scene, p, False)
    same = torch.equal(a, b)
    counters = sa[:4] == sb[:4]
    print("%-50s planes %s, counters %s, error word %d, mean %.6f" % (name, "identical" if same else "DIFFER", "equal" if counters else "DIFFER",
                                                                    sa[10], float(a.mean())), flush=True)
    if not same:
        d = (a != b).nonzero().flatten()
        print("   %d of %d values differ, first at %s" % (d.numel(), a.numel(), d[:8].tolist()))
    bad += (not same) + (not counters) + (sa[10] != 0)

if not quick:
    for name, demo, fr, sl in (("pretty 1920x1080 16x16", 1, dict(w=1920, h=1080, s=16), None),
                               ("pretty 1920x1080 16x16, 1/8 by runs", 1, dict(w=1920, h=1080, s=16), dict(rank=0, world=8, runs=True)),
                               ("pretty 1024x768 2x2", 1, dict(w=1024, h=768, s=2), None),
                               ("glass 1024x768 32x32", 2, dict(w=1024, h=768, s=32), None)):
        scene = ffi.Scene.demo(demo, 1)
        mk = lambda: ffi.params(fr["w"], fr["h"], fr["s"], seed=1)  # noqa: E731
        p = sharding.slice_params(mk, sl["rank"], sl["world"], pixel_runs=sl["runs"]) if sl else mk()
        n = p.width * p.height
        planes = torch.empty(3 * n, dtype=torch.float32, device="cuda")
        for fused in (True, False, True, False):
            q = with_sums(p, fused)
            kernel_ms, _ = r.time_trace_kernel(scene, q, 5)
            for _ in range(3):
                r.trace_device(scene, q, planes.data_ptr(), None, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                r.trace_device(scene, q, planes.data_ptr(), None, stream)
            torch.cuda.synchronize()
            call_ms = (time.perf_counter() - t0) / 10 * 1e3
            print("%-40s %-8s trace kernel %.3f ms, call %.3f ms" % (name, "fused" if fused else "unfused", kernel_ms, call_ms), flush=True)
r.close()
print("FAILED: %d" % bad if bad else "all identical")
sys.exit(1 if bad else 0)
