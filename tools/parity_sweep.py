"""A larger deterministic-mode parity sweep than the test-suite runs: every pixel of bigger frames against the
oracle, all demo scenes, both depth modes, brute force and grid.  Prints the worst relative error per case.

python tools/parity_sweep.py [WIDTH HEIGHT SAMPLES_IN_ROW]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi  # noqa: E402
from rt_amd import ffi  # noqa: E402

w, h, s = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 768, 2)
r = ffi.HipRenderer(0)
worst = 0.0
for demo, spheres in ((1, 0), (2, 0), (3, 300), (3, 3000)):
    scene = ffi.Scene.demo(demo, 1, spheres or 10000)
    for depth in (1, 0):
        for grid in ((0, ffi.FLAG_GRID) if spheres else (0,)):
            if spheres == 3000 and not grid and w * h > 300000:
                continue  # the oracle is the slow side here, not the flag
            p = ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC | grid, max_depth=depth)
            ref, rs = oracle_ffi.render(scene, ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC, max_depth=depth))
            out, st = r.render(scene, p)
            err = np.abs(ref - out) / np.maximum(np.abs(ref), 1e-3)
            same = (st.ray_queries, st.hit_queries, st.shaded_bounces) == (rs.ray_queries, rs.hit_queries, rs.shaded_bounces)
            print("demo %d spheres %5d depth %2d grid %d: max rel err %.3g, pixels > 1e-4: %d, counters equal: %s" % (
                demo, scene.sphere_count, depth or 10, 1 if grid else 0, err.max(), int((err.max(axis=2) > 1e-4).sum()), same))
            worst = max(worst, float(err.max()))
r.close()
print("worst", worst)
sys.exit(0 if worst < 1e-4 else 1)
