"""A stream of whole frames through the job API, every image downloaded into pinned host memory while the next frame is
traced (what bench.py reports as ms_per_frame_with_d2h), with the host's view of each call.

python tools/job_stream_probe.py [frames]        (under rocprofv3 --kernel-trace for the device's view)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rt_amd import ffi  # noqa: E402

r = ffi.HipRenderer(0)
scene = ffi.Scene.demo(1, 1)
p = ffi.params(1920, 1080, 16, seed=1, flags=ffi.FLAG_PIPELINED)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
host = [torch.empty(1920 * 1080 * 3, dtype=torch.float32, pin_memory=True) for _ in range(2)]


def finish_into(job, k):
    rc = r.lib.rt_hip_job_finish(job, host[k % 2].data_ptr(), None)
    assert rc == 0


IN_FLIGHT = int(os.environ.get("RT_JOBS_IN_FLIGHT", "3"))
for k, job in enumerate([r.start(scene, p) for _ in range(IN_FLIGHT)]):
    finish_into(job, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
jobs = [r.start(scene, p) for _ in range(min(IN_FLIGHT - 1, frames))]
started = len(jobs)
marks = []
for k in range(frames):
    a = time.perf_counter()
    if started < frames:
        jobs.append(r.start(scene, p))
        started += 1
    b = time.perf_counter()
    finish_into(jobs.pop(0), k)
    c = time.perf_counter()
    marks.append((a - t0, b - a, c - b))
total = time.perf_counter() - t0
for k, (at, start_ms, finish_ms) in enumerate(marks):
    print("frame %2d: at %7.2f ms start(next) took %5.2f ms, finish took %6.2f ms" % (k, at * 1e3, start_ms * 1e3, finish_ms * 1e3))
print("%.3f ms per frame" % (total / frames * 1e3))
r.close()
