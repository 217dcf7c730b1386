"""The benchmark configurations at their own sizes, on a real MI355X, through the C ABI.

Each configuration is checked twice: against the oracle where the oracle can afford it (a band of rows of the
full-size frame, or a small frame with the configuration's scene and samples per pixel), and at full size
through properties that need no CPU render (exact sample count, bitwise repeatability, slices adding up to the
frame).  Tolerances are those of tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_err(ref, out):
    return np.abs(ref - out) / np.maximum(np.abs(ref), 1e-3)


def query_counters(st):
    return (st.samples, st.ray_queries, st.hit_queries, st.shaded_bounces)


# ---- pretty_spheres 1920x1080, 16x16: bands of the real frame against the CPU path, same seed ----

def test_headline_frame_rows_against_the_cpu_path(hip, oracle, ffi, scenes):
    """Rows of the 1920x1080 16x16 frame itself (not a smaller stand-in): the oracle renders two bands of rows of
    that frame (the row loop of ray_tracing/frame.cpp:44 over a sub-range), the GPU the whole frame."""
    scene = scenes(1)
    bands = [(400, 416), (700, 716)]
    # deterministic mode, full depth: per pixel within 1e-4
    p = ffi.params(1920, 1080, 16, flags=ffi.FLAG_DETERMINISTIC)
    out, _ = hip.render(scene, p)
    for lo, hi in bands:
        ref, _ = oracle.render_rows(scene, p, lo, hi)
        err = rel_err(ref, out[lo:hi])
        assert err.max() < 1e-4, "rows %d-%d: max relative error %.3g" % (lo, hi, err.max())
    # stochastic mode, the benchmark's seed: same random streams on both sides
    p = ffi.params(1920, 1080, 16, seed=1)
    out, st = hip.render(scene, p)
    assert st.samples == 1920 * 1080 * 256
    for lo, hi in bands:
        ref, _ = oracle.render_rows(scene, p, lo, hi)
        band = out[lo:hi]
        assert np.abs(ref - band).mean() < 5e-6
        assert (rel_err(ref, band).max(axis=2) > 1e-3).mean() < 1e-3
        assert abs(float(ref.mean()) - float(band.mean())) < 1e-5


# ---- pretty_spheres 1024x768, 2x2: the reference's default window (main.cpp:58-62), the whole frame ----

def test_default_window_whole_frame_against_the_cpu_path(hip, oracle, ffi, scenes):
    scene = scenes(1)
    w, h, s = 1024, 768, 2
    p = ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    err = rel_err(ref, out)
    assert err.max() < 1e-4, "max relative error %.3g" % err.max()
    assert query_counters(st) == (rs.samples, rs.ray_queries, rs.hit_queries, rs.shaded_bounces)
    p = ffi.params(w, h, s, seed=1)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    assert st.samples == w * h * s * s == rs.samples
    assert np.abs(ref - out).mean() < 5e-6
    assert (rel_err(ref, out).max(axis=2) > 1e-3).mean() < 1e-3
    assert abs(int(st.ray_queries) - int(rs.ray_queries)) < 1e-4 * rs.ray_queries
    again, st2 = hip.render(scene, p)
    np.testing.assert_array_equal(out, again)
    assert query_counters(st) == query_counters(st2)


# ---- glass_spheres 1024x768, 10x10: the viewer's high-quality mode (app.cpp:92-99) ----

def test_glass_hq_frame_rows_against_the_cpu_path(hip, oracle, ffi, scenes):
    """Rows of the 1024x768 10x10 glass frame itself: the oracle renders two bands of rows of that frame (one through
    the spheres, one across the ground in front of them), the GPU the whole frame."""
    scene = scenes(2)
    w, h, s = 1024, 768, 10
    bands = [(300, 316), (520, 536)]
    p = ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC)
    out, st = hip.render(scene, p)
    assert st.samples == w * h * s * s
    for lo, hi in bands:
        ref, _ = oracle.render_rows(scene, p, lo, hi)
        err = rel_err(ref, out[lo:hi])
        assert err.max() < 1e-4, "rows %d-%d: max relative error %.3g" % (lo, hi, err.max())
    p = ffi.params(w, h, s, seed=1)
    out, st = hip.render(scene, p)
    assert st.samples == w * h * s * s
    assert st.samples + 5 * st.shaded_bounces <= st.ray_queries <= st.samples + 6 * st.shaded_bounces
    for lo, hi in bands:
        ref, _ = oracle.render_rows(scene, p, lo, hi)
        band = out[lo:hi]
        assert np.abs(ref - band).mean() < 5e-6
        assert (rel_err(ref, band).max(axis=2) > 1e-3).mean() < 1e-3
        assert abs(float(ref.mean()) - float(band.mean())) < 1e-5
    again, st2 = hip.render(scene, p)
    np.testing.assert_array_equal(out, again)
    assert query_counters(st) == query_counters(st2)


# ---- glass_spheres 3840x2160, 32x32 ----

def test_glass_32x32_small_frame_against_the_cpu_path(hip, oracle, ffi, scenes):
    scene = scenes(2)
    p = ffi.params(48, 27, 32, flags=ffi.FLAG_DETERMINISTIC)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    # 1024 samples per pixel: the oracle adds them in one float run, the kernel as 32 column sums of 32
    assert rel_err(ref, out).max() < 1e-4
    assert query_counters(st) == (rs.samples, rs.ray_queries, rs.hit_queries, rs.shaded_bounces)
    p = ffi.params(48, 27, 32, seed=4)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    assert np.abs(ref - out).mean() < 2e-5
    assert st.samples == rs.samples
    assert abs(int(st.ray_queries) - int(rs.ray_queries)) < 1e-4 * rs.ray_queries


def test_glass_4k_32x32_full_size_properties(hip, ffi, scenes):
    """3840x2160 with 32x32 samples per pixel: 8.49 G samples, deep refraction paths."""
    scene = scenes(2)
    w, h, s = 3840, 2160, 32
    p = ffi.params(w, h, s, seed=1)
    a, sa = hip.render(scene, p)
    assert sa.samples == w * h * s * s
    assert sa.samples + 5 * sa.shaded_bounces <= sa.ray_queries <= sa.samples + 6 * sa.shaded_bounces
    assert np.isfinite(a).all() and a.min() >= 0.0
    sky = np.float32([scene.camera.background[k] for k in range(3)])
    np.testing.assert_allclose(a[0], np.broadcast_to(sky, a[0].shape), rtol=2e-6)  # the top row sees only sky
    b, sb = hip.render(scene, p)
    np.testing.assert_array_equal(a, b)
    assert query_counters(sa) == query_counters(sb)
    # the same view at 1/8 of the resolution and the same samples per pixel has the same mean radiance
    small, _ = hip.render(scene, ffi.params(w // 8, h // 8, s, seed=2))
    assert abs(float(a.mean()) - float(small.mean())) < 2e-3


# ---- synthetic field of 10 000 spheres, 1920x1080, 8x8 (every sphere tested, LDS tiles) ----

@pytest.mark.parametrize("depth", [1, 0], ids=["D1", "D2"])
def test_ten_thousand_spheres_small_frame_against_the_cpu_path(hip, oracle, ffi, scenes, depth):
    scene = scenes(3, 1, 10000)
    assert scene.sphere_count == 10001
    p = ffi.params(48, 36, 2, flags=ffi.FLAG_DETERMINISTIC, max_depth=depth)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    assert rel_err(ref, out).max() < 1e-4
    assert query_counters(st) == (rs.samples, rs.ray_queries, rs.hit_queries, rs.shaded_bounces)
    assert st.sphere_tests > 0  # every sphere tested: not the grid


def test_ten_thousand_spheres_stochastic_against_the_cpu_path(hip, oracle, ffi, scenes):
    scene = scenes(3, 1, 10000)
    p = ffi.params(48, 36, 8, seed=3)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    assert np.abs(ref - out).mean() < 5e-5
    assert st.samples == rs.samples
    assert abs(int(st.ray_queries) - int(rs.ray_queries)) < 1e-3 * rs.ray_queries


def test_ten_thousand_spheres_full_size_properties(hip, ffi, scenes):
    """1920x1080 8x8 against 10 001 spheres through the tiled kernel: the reservation sizes, the end of the queue and
    sample sharing all take the values of the real configuration here."""
    import torch

    scene = scenes(3, 1, 10000)
    w, h, s = 1920, 1080, 8
    p = ffi.params(w, h, s, seed=1)
    a, sa = hip.render(scene, p)
    assert sa.samples == w * h * s * s
    assert sa.samples + 5 * sa.shaded_bounces <= sa.ray_queries <= sa.samples + 6 * sa.shaded_bounces
    assert sa.sphere_tests == (2 * sa.shaded_bounces + sa.samples) * scene.sphere_count  # cone + scattered ray, primary
    assert np.isfinite(a).all() and a.min() >= 0.0
    # the eight slices of eight ranks add up to the frame, and so do their counters
    stream = torch.cuda.current_stream().cuda_stream
    total = torch.zeros(3 * w * h, dtype=torch.float32, device="cuda")
    planes = torch.empty(3 * w * h, dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    counted = np.zeros(4, dtype=np.int64)
    for r in range(8):
        q = ffi.params(w, h, s, seed=1, slice_index=r, slice_count=8)
        hip.trace_device(scene, q, planes.data_ptr(), stats.data_ptr(), stream)
        torch.cuda.synchronize()
        total += planes
        counted += stats[:4].cpu().numpy()
    assert tuple(counted) == query_counters(sa)
    rgb = torch.empty(w * h * 3, dtype=torch.float32, device="cuda")
    hip.resolve_device(total.data_ptr(), rgb.data_ptr(), p, stream)
    torch.cuda.synchronize()
    np.testing.assert_allclose(rgb.cpu().numpy().reshape(h, w, 3), a, rtol=2e-6, atol=1e-7)
    # one slice again: bitwise repeatable
    q = ffi.params(w, h, s, seed=1, slice_index=7, slice_count=8)
    hip.trace_device(scene, q, total.data_ptr(), stats.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(total, planes)


def test_cost_scaled_reservations_with_sample_sharing(hip, ffi, scenes, monkeypatch):
    """A scene of 200 spheres on a frame large enough that a wave's reservation is several windows long and sized
    by the cost of a sample (three times a 64-sphere sample here): sample sharing at the end of the queue and the
    reservation policy change no bit of the frame."""
    scene = scenes(3, 1, 200)
    p = ffi.params(1024, 768, 4, seed=6)
    shared, ss = hip.render(scene, p)
    monkeypatch.setenv("RT_HIP_NO_SHARING", "1")
    plain, sp = hip.render(scene, p)
    monkeypatch.delenv("RT_HIP_NO_SHARING")
    np.testing.assert_array_equal(shared, plain)
    assert query_counters(ss) == query_counters(sp)
    piped, _ = hip.render(scene, ffi.params(1024, 768, 4, seed=6, flags=ffi.FLAG_PIPELINED))
    np.testing.assert_array_equal(piped, plain)


# ---- scenes the demos do not reach ----

def test_scene_too_large_for_resident_lds_with_path_stacks(hip, oracle, ffi):
    """200 spheres with a material each, one of them a metal rough enough (0.8: emissivity 0.18, 0.18^10 < 1e-5)
    that paths must keep the reference's reverse loop over a stack: stacks plus the resident scene would need
    more LDS than a block may have, so the scene is traced through tiles."""
    rng = np.random.default_rng(5)
    mats = [(0, (0.6, 0.6, 0.6), 0.7, 0.0)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0)]
    k = 0
    for gx in range(-7, 7):
        for gz in range(0, 14):
            k += 1
            kind = k % 3
            rough = 0.8 if k == 17 else float(rng.uniform(0.0, 0.2))
            refr = float(rng.uniform(0.8, 0.9)) if kind == 2 else float(rng.uniform(0.0, 0.2))
            mats.append((kind, tuple(float(x) for x in rng.uniform(0.2, 1.0, 3)), rough if kind == 1 else 0.7 * (kind == 0), refr))
            r = float(rng.uniform(0.25, 0.45))
            spheres.append((gx + 0.5, r - 1.0, gz + 0.5, r, len(mats) - 1))
    assert len(spheres) == 197
    light = np.array([-0.4, -1.0, 0.6]) / np.linalg.norm([-0.4, -1.0, 0.6])
    scene = ffi.Scene.build(spheres, mats, [(tuple(light), (1.0, 1.0, 1.0))],
                            dict(position=(0.0, 2.5, -6.0), direction=(0.0, -0.45, 1.0), background=(0.9, 0.9, 0.9)))
    for depth in (1, 0):
        p = ffi.params(96, 72, 2, flags=ffi.FLAG_DETERMINISTIC, max_depth=depth)
        ref, rs = oracle.render(scene, p)
        out, st = hip.render(scene, p)
        assert rel_err(ref, out).max() < 1e-4
        assert query_counters(st) == (rs.samples, rs.ray_queries, rs.hit_queries, rs.shaded_bounces)
    p = ffi.params(96, 72, 3, seed=2)
    ref, _ = oracle.render(scene, p)
    out, _ = hip.render(scene, p)
    assert np.abs(ref - out).mean() < 1e-4
