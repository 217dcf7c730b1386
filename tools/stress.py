"""Soak test of the job API: a few hundred frames of changing sizes and scenes with snapshots and cancels thrown in,
checking that device memory does not creep and every frame still matches its first rendering.

python tools/stress.py [SECONDS]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device memory counters only)

from rt_amd import ffi  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
r = ffi.HipRenderer(0)
scenes = [ffi.Scene.demo(1, 1), ffi.Scene.demo(2, 1), ffi.Scene.demo(3, 1, 500)]
sizes = [(64, 48, 2), (320, 180, 4), (200, 120, 3), (640, 360, 2), (33, 17, 5)]
rng = np.random.default_rng(1)
first = {}
free0 = None
frames = cancels = snaps = 0
t_end = time.time() + budget
while time.time() < t_end:
    si = int(rng.integers(len(scenes)))
    w, h, s = sizes[int(rng.integers(len(sizes)))]
    flags = ffi.FLAG_GRID if (si == 2 and rng.random() < 0.5) else 0
    p = ffi.params(w, h, s, seed=3, flags=flags)
    job = r.start(scenes[si], p)
    action = rng.random()
    if action < 0.2:
        r.snapshot(job, p)
        snaps += 1
    if action > 0.9:
        r.cancel(job)
        r.finish(job, p, allow_cancelled=True)
        cancels += 1
    else:
        out, st = r.finish(job, p)
        key = (si, w, h, s)
        if key in first:
            assert np.array_equal(out, first[key]), "frame changed: %r" % (key,)
        else:
            first[key] = out
        assert st.samples == w * h * s * s
    frames += 1
    if frames % 5000 == 0:
        # (a run that prints nothing for minutes is taken for hung)
        print("%d frames so far, device memory drift %.1f MB" % (frames, (free0 - torch.cuda.mem_get_info()[0]) / 1e6), flush=True)
    if frames == 50:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print("frames %d, snapshots %d, cancels %d, device memory drift %.1f MB" % (frames, snaps, cancels, (free0 - free1) / 1e6))
assert abs(free0 - free1) < 64e6
r.close()
print("stress ok")
