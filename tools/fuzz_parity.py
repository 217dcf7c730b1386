"""Random small frames through the HIP backend against the oracle: sizes, samples per row, slices, flags and
scenes drawn at random.  Deterministic frames must match the oracle per pixel (1e-4 relative; 3e-4 above 64 samples per row) with equal query
counters; stochastic frames must not change a bit when sample sharing is switched off, RT_HIP_FLAG_PIPELINED is set
or the shadow cones are switched off (scenes with several lights of any length included), and the images of a
frame's slices by runs of pixels must add up to the frame's image bit for bit.

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


def random_scene(k, offset=(0.0, 0.0, 0.0)):
    """A few dozen spheres of all materials under one to three lights whose directions are not normalised (the
    shadow cone's opening depends on the light vector's length; a short one has no cone at all).  offset moves
    the whole scene and the camera away from the origin: the filter's per-sphere constants are taken in world
    coordinates and must stay on the safe side where those are large against the spheres."""
    g = np.random.default_rng(1000 + k)
    ox, oy, oz = offset
    mats = [(0, (0.6, 0.6, 0.6), 0.7, 0.0)]
    spheres = [(ox, oy - 1001.0, oz, 1000.0, 0)]
    for _ in range(int(g.integers(5, 60))):
        kind = int(g.integers(0, 3))
        mats.append((kind, tuple(float(x) for x in g.uniform(0.1, 1.0, 3)), float(g.uniform(0.0, 0.9)) if kind == 1 else 0.7,
                     float(g.uniform(0.8, 0.9)) if kind == 2 else float(g.uniform(0.0, 0.2))))
        rad = float(g.uniform(0.15, 0.6))
        spheres.append((ox + float(g.uniform(-4, 4)), oy + rad - 1.0 + float(g.uniform(0, 1.5)), oz + float(g.uniform(0, 8)), rad,
                        len(mats) - 1))
    lights = []
    for _ in range(int(g.integers(1, 4))):
        d = g.normal(size=3)
        d[1] = -abs(d[1]) - 0.3
        d *= float(g.choice([0.2, 0.27, 0.5, 1.0, 1.0, 3.0])) / np.linalg.norm(d)
        lights.append((tuple(float(x) for x in d), tuple(float(x) for x in g.uniform(0.2, 1.0, 3))))
    return ffi.Scene.build(spheres, mats, lights, dict(position=(ox + 0.0, oy + 0.8, oz - 4.0), direction=(0.0, -0.15, 1.0),
                                                        background=(0.8, 0.85, 0.9)))


def split_scene(k, gap):
    """Two groups of small spheres `gap` apart with the camera between them and to the side, looking across both: the
    scene's centre lies in the empty middle, so every sphere and every hit point is far from it -- where the shade
    filter's products are large against the distances that decide a hit."""
    g = np.random.default_rng(2000 + k)
    mats = [(0, (0.6, 0.6, 0.6), 0.7, 0.0)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0)]
    for side in (-0.5, 0.5):
        for _ in range(int(g.integers(8, 25))):
            kind = int(g.integers(0, 3))
            mats.append((kind, tuple(float(x) for x in g.uniform(0.1, 1.0, 3)), float(g.uniform(0.0, 0.9)) if kind == 1 else 0.7,
                         float(g.uniform(0.8, 0.9)) if kind == 2 else float(g.uniform(0.0, 0.2))))
            rad = float(g.uniform(0.1, 0.5)) * max(1.0, gap / 400.0)
            spheres.append((side * gap + float(g.uniform(-3, 3)) * max(1.0, gap / 400.0), rad - 1.0 + float(g.uniform(0, 1.5)),
                            float(g.uniform(-3, 3)) * max(1.0, gap / 400.0), rad, len(mats) - 1))
    d = g.normal(size=3)
    d[1] = -abs(d[1]) - 0.5
    d /= np.linalg.norm(d)
    lights = [(tuple(float(x) for x in d), (0.9, 0.9, 0.8))]
    return ffi.Scene.build(spheres, mats, lights, dict(position=(-0.5 * gap - 8.0 * max(1.0, gap / 400.0), 1.5, -6.0 * max(1.0, gap / 400.0)),
                                                        direction=(0.98, -0.05, 0.2), background=(0.8, 0.85, 0.9)))


def without_cones(make):
    os.environ["RT_HIP_NO_CONES"] = "1"
    try:
        scene = make()
        r.upload(scene)
    finally:
        del os.environ["RT_HIP_NO_CONES"]
    return scene


makers = [lambda: ffi.Scene.demo(1, 1), lambda: ffi.Scene.demo(2, 1), lambda: ffi.Scene.demo(3, 1, 100),
          lambda: ffi.Scene.demo(3, 1, 700)] + [(lambda k=k: random_scene(k)) for k in range(8)] + [
    (lambda k=k, o=o: random_scene(k, o)) for k, o in ((8, (30.0, -20.0, 50.0)), (9, (-300.0, 200.0, 100.0)), (10, (1000.0, 3000.0, -2000.0)),
                                                       (11, (-2e4, 1e4, 3e4)))] + [
    (lambda k=k, gap=gap: split_scene(k, gap)) for k, gap in ((0, 60.0), (1, 200.0), (2, 900.0), (3, 3000.0))]
scenes = [m() for m in makers]
plain = [without_cones(m) for m in makers]
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
    grid = ffi.FLAG_GRID if (si in (2, 3) and rng.random() < 0.5) else 0
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
    d, sd = r.render(plain[si], sto)  # the same scene without shadow cones
    stable = np.array_equal(a, b) and np.array_equal(a, c) and sa.ray_queries == sb.ray_queries == sc.ray_queries
    stable = stable and np.array_equal(a, d) and sa.hit_queries == sd.hit_queries
    # slices by runs of pixels: the images of g2 slices add up to the whole frame's image bit for bit
    if cases % 4 == 0:
        g2 = int(rng.choice([2, 3, 5]))
        whole, sw = r.render(scene, ffi.params(w, h, s, flags=grid, seed=seed, max_depth=depth))
        total = np.zeros_like(whole)
        queries = 0
        for k in range(g2):
            part, sp = r.render(scene, ffi.params(w, h, s, flags=grid | ffi.FLAG_PIXEL_RUNS, seed=seed, max_depth=depth,
                                                  slice_index=k, slice_count=g2))
            total += part
            queries += sp.ray_queries
        stable = stable and np.array_equal(total.view(np.uint32), whole.view(np.uint32)) and queries == sw.ray_queries
    cases += 1
    if cases % 50 == 0:
        print("%d cases so far, worst deterministic error %.3g" % (cases, worst), flush=True)
    # The oracle adds a pixel's s x s samples in one running float sum (ray_tracing/frame.cpp:55-67), the kernel as s
    # column sums of s samples.  Over a constant region (sky) the long chain rounds the same way at every step, so the
    # two drift apart with the sample count: 8.8e-5 at 70 x 70 on most scenes, 1.3e-4 on some.  The bar of 1e-4 is for
    # the sample counts the reference uses (2, 10, 16, 32); above 64 the check is looser.
    bar = 1e-4 if s <= 64 else 3e-4
    if err >= bar or not same or not stable or not np.isfinite(out).all():
        print("FAILED: scene %d %dx%d s=%d slice %d/%d depth %d grid %d seed %d: err %.3g counters %s stable %s" % (
            si, w, h, s, idx, g, depth, grid, seed, err, same, stable))
        r.close()
        sys.exit(1)
r.close()
print("%d cases, worst deterministic error %.3g (%s)" % (cases, worst, worst_case))
