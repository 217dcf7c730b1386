"""Renders one frame on the HIP backend and prints its counters; the command rocprofv3 wraps.

python tools/render_once.py DEMO WIDTH HEIGHT SAMPLES_IN_ROW [REPEAT] [SPHERES] [FLAGS] [MAX_DEPTH]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rt_amd import ffi  # noqa: E402

if os.environ.get("RT_LIB_DIR"):  # experiments: another build of the libraries
    ffi._LIB_DIR = os.path.abspath(os.environ["RT_LIB_DIR"])

demo, w, h, s = (int(x) for x in sys.argv[1:5])
repeat = int(sys.argv[5]) if len(sys.argv) > 5 else 1
spheres = int(sys.argv[6]) if len(sys.argv) > 6 else 10000
flags = int(sys.argv[7]) if len(sys.argv) > 7 else 0
depth = int(sys.argv[8]) if len(sys.argv) > 8 else 0
r = ffi.HipRenderer(0)
sc = ffi.Scene.demo(demo, 1, spheres)
p = ffi.params(w, h, s, seed=1, flags=flags, max_depth=depth)
for _ in range(repeat):
    t0 = time.perf_counter()
    out, st = r.render(sc, p)
    wall_ms = (time.perf_counter() - t0) * 1e3
d = st.as_dict()
flop = 20.0 * sc.sphere_count * d["ray_queries"] + 30.0 * d["hit_queries"] + 250.0 * d["shaded_bounces"]
print(d)
print("kernel %.3f ms  %.1f Msamples/s  %.2f TFLOP/s algorithmic  mean %.5f  host-to-host %.3f ms" %
      (d["kernel_ms"], d["samples"] / d["kernel_ms"] / 1e3, flop / d["kernel_ms"] / 1e9, float(out.mean()), wall_ms))
r.close()
