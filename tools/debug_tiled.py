import os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle_ffi
from rt_amd import ffi
r = ffi.HipRenderer(0)
for n in (300, 700, 1100):
    sc = ffi.Scene.demo(3, 1, n)
    p = ffi.params(64, 48, 2, flags=1, max_depth=1)
    ref, rs = oracle_ffi.render(sc, p)
    for tile in ("1024", "512", "256", "64", "32"):
        os.environ["RT_HIP_FORCE_TILE"] = tile
        out, st = r.render(sc, p)
        print("n", n, "tile", tile, "hits", rs.hit_queries, st.hit_queries, "exact", st.exact_tests)
