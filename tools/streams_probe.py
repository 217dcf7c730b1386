"""A stream of frames on 2, 3 or 4 streams of equal or alternating priority: the headline frame whole and as the 1/4
and 1/8 slices a rank of four or eight traces (rt_hip_trace_device only: no collective, no resolve pass).  Configurations
take turns, five times over; minimum and median period per frame.

python tools/streams_probe.py
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rt_amd import ffi, sharding  # noqa: E402

r = ffi.HipRenderer(0)
scene = ffi.Scene.demo(1, 1)
w, h, s = 1920, 1080, 16
configs = ((0, -1), (0, 0), (0, 0, 0), (0, -1, -1), (0, -1, 0), (0, 0, 0, 0))
for sl, room in ((8, True), (8, False), (4, True), (None, False)):
    mk = lambda: ffi.params(w, h, s, seed=1, flags=ffi.FLAG_PIPELINED | (ffi.FLAG_LEAVE_ROOM if room else 0))  # noqa: E731
    p = sharding.slice_params(mk, 0, sl, pixel_runs=True) if sl else mk()
    planes = [torch.empty(3 * w * h, dtype=torch.float32, device="cuda") for _ in range(4)]
    sets = {c: [torch.cuda.Stream(priority=q) for q in c] for c in configs}
    res = {c: [] for c in configs}
    for rep in range(5):
        for c in configs:
            streams, n = sets[c], len(c)
            for k in range(2 * n):
                r.trace_device(scene, p, planes[k % n].data_ptr(), None, streams[k % n].cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frames = 24 if sl else 12
            for k in range(frames):
                r.trace_device(scene, p, planes[k % n].data_ptr(), None, streams[k % n].cuda_stream)
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / frames * 1e3)
    for c in configs:
        print("1/%s of the frame, block slots left free %d, stream priorities %-14s: %.3f ms per frame at least, median %.3f"
              % (sl or 1, room, c, min(res[c]), statistics.median(res[c])), flush=True)
r.close()
