"""Does a device-to-host copy on another stream run beside a persistent trace kernel, or does it wait for block slots?
Times a 24.9 MB copy into pinned host memory alone, beside a running 3840x2160 job (which leaves 64 block slots free)
and beside a 1080p job (which fills every slot).  On the MI355X boxes of the pool: 0.45-0.55 ms in every case.

python tools/d2h_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_amd import ffi
r = ffi.HipRenderer(0)
scene = ffi.Scene.demo(2, 1)
p = ffi.params(3840, 2160, 32, seed=3)
src = torch.ones(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
dst = torch.empty(1920 * 1080 * 3, dtype=torch.float32, pin_memory=True)
s2 = torch.cuda.Stream()
torch.cuda.synchronize()
# alone
for _ in range(2):
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        dst.copy_(src, non_blocking=True)
    s2.synchronize()
    print("D2H 24.9 MB alone: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
job = r.start(scene, p)
time.sleep(0.05)
for _ in range(3):
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        dst.copy_(src, non_blocking=True)
    s2.synchronize()
    print("D2H 24.9 MB beside a running trace: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
out, st = r.finish(job, p)
print("job finished %.1f ms later" % ((time.perf_counter() - t0) * 1e3))
# the same for a frame without room for other kernels (headline size: short job)
p2 = ffi.params(1920, 1080, 16, seed=3)
scene1 = ffi.Scene.demo(1, 1)
for _ in range(2):
    job = r.start(scene1, p2)
    time.sleep(0.002)
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        dst.copy_(src, non_blocking=True)
    s2.synchronize()
    t1 = time.perf_counter()
    out, st = r.finish(job, p2)
    print("D2H beside a running 1080p trace (no free block slots): %.2f ms; job finished %.1f ms after that" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
r.close()
