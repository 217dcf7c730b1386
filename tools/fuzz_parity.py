"""Random small frames through the HIP backend against the oracle: sizes, samples per row, slices, flags and
scenes drawn at random.  Deterministic frames must match the oracle per pixel (1e-4 relative) with equal query
counters; stochastic frames must not change a bit when sample sharing is switched off or RT_HIP_FLAG_PIPELINED is set.

python tools/fuzz_parity.py [SECONDS] [SEED]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi  # noqa: E402
from rt_amd import ffi  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r = ffi.HipRenderer(0)
scenes = [ffi.Scene.demo(1, 1), ffi.Scene.demo(2, 1), ffi.Scene.demo(3, 1, 100), ffi.Scene.demo(3, 1, 700)]
cases = worst = 0
worst_case = ""
t_end = time.time() + budget
while time.time() < t_end:
    si = int(rng.integers(len(scenes)))
    scene = scenes[si]
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    s = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33, 70]))
    if w * h * s * s > 3_000_000:
        continue
    g = int(rng.choice([1, 1, 2, 3, 5, 8, 16]))
    idx = int(rng.integers(g))
    depth = int(rng.choice([0, 1, 3]))
    grid = ffi.FLAG_GRID if (si >= 2 and rng.random() < 0.5) else 0
    seed = int(rng.integers(1, 1000))
    det = ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC | grid, max_depth=depth, slice_index=idx, slice_count=g)
    ref, rs = oracle_ffi.render(scene, ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC, max_depth=depth, slice_index=idx,
                                                  slice_count=g), 8)
    out, st = r.render(scene, det)
    err = float((np.abs(ref - out) / np.maximum(np.abs(ref), 1e-3)).max())
    if err > worst:
        worst, worst_case = err, "scene %d %dx%d s=%d slice %d/%d depth %d grid %d" % (si, w, h, s, idx, g, depth, grid)
    same = (st.samples, st.ray_queries, st.hit_queries, st.shaded_bounces) == (rs.samples, rs.ray_queries, rs.hit_queries,
                                                                              rs.shaded_bounces)
    sto = ffi.params(w, h, s, flags=grid, seed=seed, max_depth=depth, slice_index=idx, slice_count=g)
    a, sa = r.render(scene, sto)
    os.environ["RT_HIP_NO_SHARING"] = "1"
    b, sb = r.render(scene, sto)
    del os.environ["RT_HIP_NO_SHARING"]
    c, sc = r.render(scene, ffi.params(w, h, s, flags=grid | ffi.FLAG_PIPELINED, seed=seed, max_depth=depth, slice_index=idx,
                                       slice_count=g))
    stable = np.array_equal(a, b) and np.array_equal(a, c) and sa.ray_queries == sb.ray_queries == sc.ray_queries
    cases += 1
    if err >= 1e-4 or not same or not stable or not np.isfinite(out).all():
        print("FAILED: scene %d %dx%d s=%d slice %d/%d depth %d grid %d seed %d: err %.3g counters %s stable %s" % (
            si, w, h, s, idx, g, depth, grid, seed, err, same, stable))
        r.close()
        sys.exit(1)
r.close()
print("%d cases, worst deterministic error %.3g (%s)" % (cases, worst, worst_case))
