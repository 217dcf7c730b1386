#!/usr/bin/env python3
"""Kernel time of one configuration (HIP events around the trace kernel, launched alone):
  python tools/time_kernel.py DEMO WIDTH HEIGHT SAMPLES [SPHERES] [LAUNCHES] [grid] [slices=N] [runs]
RT_AMD_LIB_DIR selects another build of the libraries."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rt_amd import ffi  # noqa: E402

demo, w, h, s = (int(x) for x in sys.argv[1:5])
spheres = int(sys.argv[5]) if len(sys.argv) > 5 else 10000
launches = int(sys.argv[6]) if len(sys.argv) > 6 else 5
flags = ffi.FLAG_GRID if "grid" in sys.argv[7:] else 0
slices = max([int(a.split("=")[1]) for a in sys.argv[7:] if a.startswith("slices=")] + [1])
r = ffi.HipRenderer(0)
scene = ffi.Scene.demo(demo, 1, spheres)
if "runs" in sys.argv[7:]:  # slices by runs of pixels instead of by samples
    flags |= ffi.FLAG_PIXEL_RUNS
p = ffi.params(w, h, s, seed=1, flags=flags, slice_index=slices - 1, slice_count=slices)
r.time_trace_kernel(scene, p, 1)
ms, st = r.time_trace_kernel(scene, p, launches)
print("%s demo %d %dx%d s=%d spheres=%d slice 1/%d: kernel %.3f ms, %.1f Msamples/s, filter tests %d, exact %d"
      % (os.environ.get("RT_AMD_LIB_DIR", "lib"), demo, w, h, s, scene.sphere_count, slices, ms, st.samples / ms / 1e3,
         st.sphere_tests, st.exact_tests), flush=True)
r.close()
