"""Scratch: does a snapshot run beside the trace?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rt_amd import ffi
r = ffi.HipRenderer(0)
sc = ffi.Scene.demo(2, 1)
p = ffi.params(1920, 1080, 16, seed=3)
r.render(sc, p)
for _ in range(2):
    t0 = time.perf_counter()
    job = r.start(sc, p)
    t1 = time.perf_counter()
    snap = r.snapshot(job, p)
    t2 = time.perf_counter()
    running = r.poll(job) == ffi.RT_HIP_ERR_NOT_READY
    final, st = r.finish(job, p)
    t3 = time.perf_counter()
    print("start %.2f ms, snapshot %.2f ms, still running %s, finish %.2f ms, kernel %.2f ms, dark %.3f" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, running, (t3 - t2) * 1e3, st.kernel_ms,
        float((snap.sum(axis=2) < final.sum(axis=2) * 0.999).mean())))
r.close()
