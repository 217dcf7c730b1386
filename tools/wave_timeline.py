"""When the waves of the resident trace kernel start, finish staging and finish: one launch with the probe build
(make prof) per slice count, summarised.  Shows the fixed cost of a launch (ramp-up, staging) and its tail (the
time between the first and the last wave running out of work).

python tools/wave_timeline.py [width height samples_in_row]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rt_amd import ffi, sharding  # noqa: E402

ffi._LIB_DIR = os.path.join(ROOT, "rt_amd", "lib_prof")
WAVES = 8192

r = ffi.HipRenderer(0)
lib = ffi.load_hip()
scene = ffi.Scene.demo(1, 1)
w, h, s = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 16)
planes = torch.empty(3 * w * h, dtype=torch.float32, device="cuda")
stats = torch.zeros(16, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
buf = (C.c_ulonglong * (4 * WAVES))()
for g in (1, 8):
    p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1), 0, g)
    for _ in range(2):
        r.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream)
    torch.cuda.synchronize()
    lib.rt_hip_debug_wave_times(buf, WAVES)
    r.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream)
    torch.cuda.synchronize()
    lib.rt_hip_debug_wave_times(buf, WAVES)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(WAVES, 4).astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0  # 100 MHz clock
    q = [0, 1, 10, 50, 90, 99, 100]
    print("G = %d: %d waves" % (g, len(t)))
    for k, name in enumerate(("entry", "staged", "done", "dry")):
        print("  %-7s us after the first entry, percentiles %s: %s" % (name, q, np.round(np.percentile(us[:, k], q), 1)))
    print("  done - dry, us, percentiles %s: %s" % (q, np.round(np.percentile(us[:, 2] - us[:, 3], q), 1)))
    busy = (us[:, 2] - us[:, 1]).sum()
    span = us[:, 2].max()
    print("  span %.1f us; waves inside the loop %.1f %% of waves x span" % (span, 100.0 * busy / (span * len(t))))
r.close()
