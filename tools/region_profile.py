"""Where the trace kernel's time goes: renders one frame with the probe build (make prof) and prints the share
of wave clock ticks per region of the loop.

python tools/region_profile.py DEMO WIDTH HEIGHT SAMPLES_IN_ROW [SPHERES]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rt_amd import ffi  # noqa: E402

ffi._LIB_DIR = os.path.join(ROOT, "rt_amd", "lib_prof")

NAMES = ["0 round top: fetch item, begin column", "1 primary ray set-up", "2 primary filter", "3 primary resolve",
         "4 primary hit point / normal", "5 material, scatter draw, scatter, specular", "6 (pass loop overhead)",
         "7 shade filter (6 rays)", "8 shade resolve", "9 shadow draws, ray set-up", "10 light sum, push, accumulate",
         "11 -"]

demo, w, h, s = (int(x) for x in sys.argv[1:5])
spheres = int(sys.argv[5]) if len(sys.argv) > 5 else 10000
r = ffi.HipRenderer(0)
lib = ffi.load_hip()
sc = ffi.Scene.demo(demo, 1, spheres)
p = ffi.params(w, h, s, seed=1)
buf = (C.c_ulonglong * 14)()
r.render(sc, p)  # warm-up
lib.rt_hip_debug_region_ticks(buf, 14)  # reading clears the counters
out, st = r.render(sc, p)
lib.rt_hip_debug_region_ticks(buf, 14)
total = float(sum(buf[:11]))
print(st.as_dict())
for k in range(11):
    print("%6.2f %%  %s" % (100.0 * buf[k] / total, NAMES[k]))
# slots 11 to 13 count wave-level loop trips (one count per wave and trip) of the three resolution loops
wave_trips = st.lane_trips / 64.0
print("per loop trip of a wave: %.2f primary rounds with %.2f trips of the primary candidate loop, %.2f trips of the "
      "cone-sphere loop, %.2f trips of the scattered-ray loop; %.2f exact tests and %.2f cone spheres per lane"
      % (st.lane_rounds / st.lane_trips, buf[11] / wave_trips, buf[12] / wave_trips, buf[13] / wave_trips,
         st.exact_tests / st.lane_trips, st.cone_tests / st.lane_trips))
r.close()
