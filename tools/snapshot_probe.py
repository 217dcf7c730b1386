"""Does a snapshot of a running job run beside the trace kernel, or does it wait for the frame to end?

Starts a 1080p 16x16 glass_spheres job, takes one snapshot right away and reports how long the snapshot took,
whether the job was still running afterwards and what share of the pixels was still unfinished in it.

python tools/snapshot_probe.py [WIDTH HEIGHT SAMPLES_IN_ROW]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rt_amd import ffi  # noqa: E402

r = ffi.HipRenderer(0)
sc = ffi.Scene.demo(2, 1)
w, h, s_row = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1920, 1080, 16)
p = ffi.params(w, h, s_row, seed=3)
r.render(sc, p)  # warm-up
for _ in range(2):
    t0 = time.perf_counter()
    job = r.start(sc, p)
    t1 = time.perf_counter()
    snap = r.snapshot(job, p)
    t2 = time.perf_counter()
    running = r.poll(job) == ffi.RT_HIP_ERR_NOT_READY
    final, st = r.finish(job, p)
    t3 = time.perf_counter()
    print("start %.2f ms, snapshot %.2f ms, still running %s, finish %.2f ms, kernel %.2f ms, unfinished %.3f" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, running, (t3 - t2) * 1e3, st.kernel_ms,
        float((snap.sum(axis=2) < final.sum(axis=2) * 0.999).mean())))
r.close()
