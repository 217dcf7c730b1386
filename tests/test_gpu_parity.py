"""Parity of the HIP backend with the oracle, through the C ABI, on a real MI355X.

Tolerances:
  * deterministic mode (no stochastic scattering): every pixel within 1e-4 relative (floor 1e-3 absolute
    radiance in the denominator) of the CPU path;
  * stochastic mode, same Philox streams on both sides: mean absolute error below 5e-6 and at most 0.1 % of
    the pixels off by more than 1e-3 relative (pixels where a last-bit difference flips a branch; measured at
    640x480 4x4: 1.6e-7 and 0.007 %);
  * converged renders (144 samples per pixel) with different seeds: PSNR above 35 dB, mean absolute error
    within 1.25x of what two CPU renders with different seeds differ by, mean radiance within 1e-3.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_err(ref, out):
    return np.abs(ref - out) / np.maximum(np.abs(ref), 1e-3)


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / max(mse, 1e-20))


@pytest.mark.parametrize("demo,spheres", [(1, 0), (2, 0), (3, 300), (3, 3000)], ids=["pretty", "glass", "field300", "field3000"])
@pytest.mark.parametrize("depth", [1, 0], ids=["D1", "D2"])
def test_deterministic_mode_per_pixel(hip, oracle, ffi, scenes, demo, spheres, depth):
    scene = scenes(demo, 1, spheres)
    w, h = (128, 96) if spheres < 1000 else (64, 48)
    p = ffi.params(w, h, 2, flags=ffi.FLAG_DETERMINISTIC, max_depth=depth)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    err = rel_err(ref, out)
    assert err.max() < 1e-4, "max relative error %.3g at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert st.samples == rs.samples and st.ray_queries == rs.ray_queries
    assert st.hit_queries == rs.hit_queries and st.shaded_bounces == rs.shaded_bounces


@pytest.mark.parametrize("name,demo,spheres", [("pretty", 1, 0), ("glass", 2, 0), ("synthetic300", 3, 300)])
@pytest.mark.parametrize("mode,depth", [("d1", 1), ("d2", 0)])
def test_against_committed_golden_frames(hip, ffi, scenes, name, demo, spheres, mode, depth):
    golden = np.load(os.path.join(GOLDEN, "frame_%s_%s.npy" % (name, mode)))
    p = ffi.params(64, 48, 2, flags=ffi.FLAG_DETERMINISTIC, seed=1, max_depth=depth)
    out, _ = hip.render(scenes(demo, 1, spheres), p)
    assert rel_err(golden, out).max() < 1e-4


@pytest.mark.parametrize("w,h,s", [(1, 1, 1), (3, 5, 1), (37, 23, 3), (130, 2, 4), (64, 48, 5)])
def test_ragged_frame_shapes(hip, oracle, ffi, scenes, w, h, s):
    scene = scenes(1)
    p = ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC)
    ref, _ = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    assert st.samples == w * h * s * s
    assert rel_err(ref, out).max() < 1e-4


@pytest.mark.parametrize("demo", [1, 2], ids=["pretty", "glass"])
def test_stochastic_mode_shares_the_random_streams(hip, oracle, ffi, scenes, demo):
    scene = scenes(demo)
    p = ffi.params(320, 240, 3, seed=7)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    assert np.abs(ref - out).mean() < 5e-6
    assert (rel_err(ref, out).max(axis=2) > 1e-3).mean() < 1e-3
    assert st.samples == rs.samples
    assert abs(int(st.ray_queries) - int(rs.ray_queries)) < 1e-4 * rs.ray_queries


@pytest.mark.parametrize("demo", [1, 2], ids=["pretty", "glass"])
def test_converged_render(hip, oracle, ffi, scenes, demo):
    """Different seeds on the two sides: only the converged image can agree."""
    scene = scenes(demo)
    ref, _ = oracle.render(scene, ffi.params(160, 120, 12, seed=11))
    ref2, _ = oracle.render(scene, ffi.params(160, 120, 12, seed=77))
    out, _ = hip.render(scene, ffi.params(160, 120, 12, seed=1234))
    assert psnr(ref, out) > 35.0
    # the Monte-Carlo noise floor at 144 samples per pixel is what two CPU renders differ by
    noise = float(np.abs(ref - ref2).mean())
    assert np.abs(ref - out).mean() < 1.25 * noise < 1e-2
    assert abs(float(ref.mean()) - float(out.mean())) < 1e-3


def test_empty_scene_and_no_lights(hip, oracle, ffi):
    empty = ffi.Scene.build([], [], [], dict(position=(0, 0, 0), direction=(0, 0, 1), background=(0.2, 0.5, 0.9)))
    out, st = hip.render(empty, ffi.params(33, 17, 2))
    np.testing.assert_allclose(out, np.broadcast_to(np.float32([0.2, 0.5, 0.9]), out.shape), rtol=1e-6)
    assert st.hit_queries == 0 and st.ray_queries == st.samples
    mats = [(0, (0.7, 0.6, 0.5), 0.7, 0.0), (2, (0.3, 0.9, 0.8), 0.0, 0.85)]
    dark = ffi.Scene.build([(0.0, -1001.0, 0.0, 1000.0, 0), (0.0, -0.4, 1.0, 0.6, 1)], mats, [],
                           dict(position=(0.0, 0.5, -3.0), direction=(0.0, -0.2, 1.0)))
    p = ffi.params(64, 48, 2, flags=ffi.FLAG_DETERMINISTIC)
    ref, rs = oracle.render(dark, p)
    out, st = hip.render(dark, p)
    assert rel_err(ref, out).max() < 1e-4 and st.ray_queries == rs.ray_queries


def test_several_lights(hip, oracle, ffi):
    def unit(d):
        d = np.asarray(d, dtype=np.float64)
        return tuple(d / np.linalg.norm(d))

    mats = [(0, (0.7, 0.6, 0.5), 0.7, 0.0), (1, (0.9, 0.8, 0.3), 0.1, 0.15), (2, (0.3, 0.9, 0.8), 0.0, 0.85)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0), (-1.2, -0.5, 2.0, 0.5, 1), (0.0, -0.4, 1.0, 0.6, 2), (1.3, -0.6, 2.5, 0.4, 0)]
    lights = [(unit((0.0, -1.0, 0.0)), (1.0, 0.5, 0.25)), (unit((0.6, -0.6, 0.5)), (0.2, 0.9, 0.2)),
              (unit((-0.6, -0.6, -0.5)), (0.5, 0.5, 0.9))]
    scene = ffi.Scene.build(spheres, mats, lights, dict(position=(0.0, 0.5, -3.0), direction=(0.0, -0.2, 1.0)))
    p = ffi.params(96, 72, 2, flags=ffi.FLAG_DETERMINISTIC)
    ref, rs = oracle.render(scene, p)
    out, st = hip.render(scene, p)
    assert rel_err(ref, out).max() < 1e-4 and st.ray_queries == rs.ray_queries
    p = ffi.params(96, 72, 3, seed=2)
    ref, _ = oracle.render(scene, p)
    out, _ = hip.render(scene, p)
    assert np.abs(ref - out).mean() < 1e-4


def test_energy_break_and_absorbing_metal(hip, oracle, ffi):
    mats = [(1, (0.9, 0.9, 0.9), 0.95, 0.1), (0, (0.5, 0.5, 0.5), 0.7, 0.0)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0), (0.0, 0.0, 2.0, 1.0, 0), (2.1, 0.0, 2.0, 1.0, 0), (-2.1, 0.0, 2.0, 1.0, 1)]
    scene = ffi.Scene.build(spheres, mats, [((0.0, -1.0, 0.0), (1.0, 1.0, 1.0))],
                            dict(position=(0.0, 0.5, -4.0), direction=(0.0, -0.1, 1.0)))
    p = ffi.params(96, 72, 3, seed=8)
    ref, _ = oracle.render(scene, p)
    out, _ = hip.render(scene, p)
    assert np.abs(ref - out).mean() < 1e-4


@pytest.mark.parametrize("demo", [1, 2], ids=["pretty", "glass"])
def test_stack_path_equals_forward_accumulation(hip, oracle, ffi, demo, monkeypatch):
    """Demo scenes accumulate path colours forward; RT_HIP_FORCE_STACK keeps the reference's reverse loop over
    the LDS stack for scenes created while it is set."""
    fwd = ffi.Scene.demo(demo, 1)
    p = ffi.params(128, 96, 2, flags=ffi.FLAG_DETERMINISTIC)
    ref, rs = oracle.render(fwd, p)
    a, sa = hip.render(fwd, p)
    monkeypatch.setenv("RT_HIP_FORCE_STACK", "1")
    stack = ffi.Scene.demo(demo, 1)
    b, sb = hip.render(stack, p)
    monkeypatch.delenv("RT_HIP_FORCE_STACK")
    assert rel_err(ref, a).max() < 1e-4 and rel_err(ref, b).max() < 1e-4
    assert rel_err(a, b).max() < 2e-6
    assert sa.ray_queries == sb.ray_queries == rs.ray_queries


@pytest.mark.parametrize("width,height,s,slices", [(64, 48, 80, 1), (160, 120, 5, 3), (320, 240, 16, 8), (96, 72, 2, 8)])
def test_sample_sharing_changes_no_bit(hip, ffi, scenes, monkeypatch, width, height, s, slices):
    """At the end of a launch idle lanes trace samples of other lanes' columns and a fold pass completes the sums;
    RT_HIP_NO_SHARING switches that off.  Same frame, same counters (long columns, strided slices, columns of
    which a slice owns no sample)."""
    for demo in (1, 2):
        scene = scenes(demo)
        p = ffi.params(width, height, s, seed=6, slice_index=slices - 1, slice_count=slices)
        shared, ss = hip.render(scene, p)
        monkeypatch.setenv("RT_HIP_NO_SHARING", "1")
        plain, sp = hip.render(scene, p)
        monkeypatch.delenv("RT_HIP_NO_SHARING")
        np.testing.assert_array_equal(shared, plain)
        assert (ss.samples, ss.ray_queries, ss.hit_queries, ss.shaded_bounces) == \
               (sp.samples, sp.ray_queries, sp.hit_queries, sp.shaded_bounces)
        # RT_HIP_FLAG_PIPELINED is a hint (it changes how work is reserved at the end of the queue): the image must not change
        q = ffi.params(width, height, s, seed=6, slice_index=slices - 1, slice_count=slices, flags=ffi.FLAG_PIPELINED)
        piped, _ = hip.render(scene, q)
        np.testing.assert_array_equal(piped, plain)


@pytest.mark.parametrize("slices", [2, 3, 8])
def test_slices_add_up(hip, oracle, ffi, scenes, slices):
    scene = scenes(1)
    full, fs = hip.render(scene, ffi.params(96, 72, 2, seed=9))
    total = np.zeros(full.shape, dtype=np.float64)
    samples = 0
    for i in range(slices):
        p = ffi.params(96, 72, 2, seed=9, slice_index=i, slice_count=slices)
        part, st = hip.render(scene, p)
        ref, _ = oracle.render(scene, p)
        assert np.abs(ref - part).mean() < 1e-4
        total += part
        samples += st.samples
    assert samples == fs.samples
    np.testing.assert_allclose(total, full, rtol=2e-6, atol=1e-7)


def test_full_size_frame_properties(hip, ffi, scenes):
    """The benchmark's frame (1920x1080, 16x16): properties that need no CPU render."""
    scene = scenes(1)
    p = ffi.params(1920, 1080, 16, seed=1)
    a, sa = hip.render(scene, p)
    b, sb = hip.render(scene, p)
    np.testing.assert_array_equal(a, b)  # no atomics on pixels: bitwise reproducible
    assert sa.samples == 1920 * 1080 * 256
    assert sa.samples + 5 * sa.shaded_bounces <= sa.ray_queries <= sa.samples + 6 * sa.shaded_bounces
    # executed filter tests: per shaded bounce the shadow cone and the scattered ray against every sphere, primary
    # rays once per column; the spheres a cone touched (few) went through five more tests each
    columns = 1920 * 1080 * 16
    assert sa.sphere_tests == (2 * sa.shaded_bounces + columns) * scene.sphere_count
    assert 0 < sa.cone_tests < 0.05 * sa.shaded_bounces * scene.sphere_count
    np.testing.assert_allclose(a[0, :, :], 0.9, rtol=1e-6)  # the top row sees only sky
    assert np.isfinite(a).all() and a.min() >= 0.0
    # the low-resolution oracle render of the same view has the same mean radiance
    assert abs(float(a.mean()) - 0.746) < 0.01


@pytest.mark.parametrize("demo,spheres,size", [(1, 0, (640, 360, 4)), (2, 0, (320, 180, 6)), (3, 600, (160, 90, 2))],
                         ids=["pretty", "glass", "field600"])
def test_shadow_cones_change_no_bit(hip, ffi, demo, spheres, size, monkeypatch):
    """The five shadow rays of a light are filtered as one cone first; scenes created under RT_HIP_NO_CONES intersect
    them with every sphere.  Same frame, same query counters, far fewer exact intersections."""
    w, h, s = size
    with_cones = ffi.Scene.demo(demo, 1, spheres)
    monkeypatch.setenv("RT_HIP_NO_CONES", "1")
    without = ffi.Scene.demo(demo, 1, spheres)
    hip.upload(without)
    monkeypatch.delenv("RT_HIP_NO_CONES")
    for flags, seed in ((ffi.FLAG_DETERMINISTIC, 1), (0, 5)):
        p = ffi.params(w, h, s, flags=flags, seed=seed)
        a, sa = hip.render(with_cones, p)
        b, sb = hip.render(without, p)
        np.testing.assert_array_equal(a, b)
        assert (sa.samples, sa.ray_queries, sa.hit_queries, sa.shaded_bounces) == \
               (sb.samples, sb.ray_queries, sb.hit_queries, sb.shaded_bounces)
        assert sa.sphere_tests == sb.sphere_tests and sa.cone_tests < 0.2 * sb.cone_tests and sa.exact_tests < sb.exact_tests


@pytest.mark.parametrize("spheres", [300, 3000])
def test_grid_walk_equals_testing_every_sphere(hip, oracle, ffi, scenes, spheres):
    """RT_HIP_FLAG_GRID only changes which spheres the exact intersection is run on: the frame and the query
    counters are those of the brute-force trace, bit for bit, in both modes."""
    scene = scenes(3, 1, spheres)
    for flags, seed in ((ffi.FLAG_DETERMINISTIC, 1), (0, 5)):
        p = ffi.params(96, 72, 2, flags=flags, seed=seed)
        brute, sb = hip.render(scene, p)
        grid, sg = hip.render(scene, ffi.params(96, 72, 2, flags=flags | ffi.FLAG_GRID, seed=seed))
        np.testing.assert_array_equal(grid, brute)
        assert (sg.samples, sg.ray_queries, sg.hit_queries, sg.shaded_bounces) == \
               (sb.samples, sb.ray_queries, sb.hit_queries, sb.shaded_bounces)
        assert sg.sphere_tests == 0 and sg.exact_tests < 0.15 * sb.sphere_tests
    p = ffi.params(96, 72, 2, flags=ffi.FLAG_DETERMINISTIC | ffi.FLAG_GRID)
    ref, rs = oracle.render(scene, ffi.params(96, 72, 2, flags=ffi.FLAG_DETERMINISTIC))
    out, st = hip.render(scene, p)
    assert rel_err(ref, out).max() < 1e-4 and st.hit_queries == rs.hit_queries


def test_grid_walk_over_hundreds_of_cells(hip, ffi):
    """A row of 500 spheres seen at a grazing angle: walks of hundreds of cells far from where they start."""
    from test_emu_vs_oracle import long_row_scene

    scene = long_row_scene(ffi)
    for flags, seed in ((ffi.FLAG_DETERMINISTIC, 1), (0, 3)):
        brute, sb = hip.render(scene, ffi.params(320, 120, 3, flags=flags, seed=seed))
        grid, sg = hip.render(scene, ffi.params(320, 120, 3, flags=flags | ffi.FLAG_GRID, seed=seed))
        np.testing.assert_array_equal(grid, brute)
        assert (sg.samples, sg.ray_queries, sg.hit_queries, sg.shaded_bounces) == \
               (sb.samples, sb.ray_queries, sb.hit_queries, sb.shaded_bounces)
        assert sg.sphere_tests == 0 and sb.hit_queries > 0.3 * sb.samples


def test_grid_flag_is_ignored_for_small_scenes(hip, ffi, scenes):
    scene = scenes(1)
    a, sa = hip.render(scene, ffi.params(64, 48, 2, seed=2))
    b, sb = hip.render(scene, ffi.params(64, 48, 2, seed=2, flags=ffi.FLAG_GRID))
    np.testing.assert_array_equal(a, b)
    assert sa.sphere_tests == sb.sphere_tests > 0


def test_non_finite_spheres_are_refused(hip, ffi):
    """Sphere::TraceNearest rejects with `t < tmin || t > tmax`, which a NaN distance passes, so in the reference
    a sphere with a NaN in it "hits" wherever nothing nearer does.  The backend refuses such scenes instead."""
    mats = [(0, (0.7, 0.6, 0.5), 0.7, 0.0)]
    for bad in ((float("nan"), 0.0, 2.0, 0.5, 0), (0.0, 0.5, 3.0, float("inf"), 0)):
        scene = ffi.Scene.build([(0.0, -1001.0, 0.0, 1000.0, 0), bad], mats, [((0.0, -1.0, 0.0), (1.0, 1.0, 1.0))],
                                dict(position=(0.0, 0.5, -3.0), direction=(0.0, -0.2, 1.0)))
        with pytest.raises(ffi.HipError, match="non-finite sphere"):
            hip.render(scene, ffi.params(16, 16, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("s_row,slices", [(2, 8), (3, 2), (5, 5), (16, 8)])
def test_slice_rule_on_an_empty_scene(hip, ffi, s_row, slices):
    """No spheres: every sample is the background, so a slice's pixel counts the samples it owns
    (rt_amd.sharding.owner_of is the documented rule)."""
    from rt_amd import sharding

    w, h = 70, 33
    scene = ffi.Scene.build([], [(0, (0.5, 0.5, 0.5), 0.5, 0.0)], [], dict(position=(0, 0, -3), direction=(0, 0, 1)))
    bg = np.array([scene.camera.background[k] for k in range(3)], dtype=np.float64)
    pix = np.arange(w * h)
    for r in (0, slices - 1):
        owned = np.zeros(w * h)
        for sx in range(s_row):
            for sy in range(s_row):
                owned += (((pix * s_row + sx) * s_row + sy) % slices) == r
        assert sharding.owner_of(5, 1, 0, s_row, slices) == ((5 * s_row + 1) * s_row) % slices
        out, st = hip.render(scene, ffi.params(w, h, s_row, seed=1, slice_index=r, slice_count=slices))
        expect = owned.reshape(h, w)[:, :, None] * bg[None, None, :] / (s_row * s_row)
        np.testing.assert_allclose(out, expect, rtol=1e-5)
        assert st.samples == int(owned.sum())


@pytest.mark.parametrize("s", [2, 4, 8, 10, 16, 32])
def test_accumulate_pass_through_lds_changes_no_bit(hip, ffi, scenes, monkeypatch, s):
    """The accumulate pass has a variant per common sample count that reads whole lines and transposes them in LDS;
    RT_HIP_PLAIN_COLSUM keeps the lane-per-pixel kernel.  Same sums in the same order: identical frames, on frames
    whose pixel count is no multiple of anything."""
    scene = scenes(1)
    for w, h in ((131, 67), (64, 64), (1, 1), (257, 3)):
        p = ffi.params(w, h, s, seed=3)
        a, _ = hip.render(scene, p)
        monkeypatch.setenv("RT_HIP_PLAIN_COLSUM", "1")
        b, _ = hip.render(scene, p)
        monkeypatch.delenv("RT_HIP_PLAIN_COLSUM")
        np.testing.assert_array_equal(a, b)


def test_short_exact_sequences_equal_the_compilers(hip):
    """Exact intersections and normalisations take correctly rounded square roots and quotients from sequences shorter
    than the compiler's general ones (rt_core.h sqrt_rn_core, div_rn_core).  The device compares them: sqrt and
    1 / sqrt over every float, the quotient and the near root on one operand set per float pattern."""
    assert hip.check_exact_math() == [0, 0, 0, 0]


def test_device_philox_against_the_known_answers(hip):
    """The kernels' Philox4x32-10 takes both halves of each product from one v_mad_u64_u32 (inline assembly); the
    device runs it on the Random123 known-answer inputs, and on a million further blocks against the host's form."""
    words, run_device, run_host = hip.check_philox()
    assert words[0:4] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert words[4:8] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert words[8:12] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert run_device == run_host


def test_library_math_changes_no_bit(hip, ffi, scenes, monkeypatch):
    """RT_HIP_LIBRARY_MATH makes every exact intersection use the compiler's IEEE sequences: same frames, same
    counters, in deterministic and stochastic mode, brute force and grid."""
    for demo, spheres, flags in ((1, 0, 0), (2, 0, 0), (3, 300, 0), (3, 300, ffi.FLAG_GRID), (1, 0, ffi.FLAG_DETERMINISTIC)):
        scene = scenes(demo, 1, spheres)
        p = ffi.params(160, 120, 4, seed=11, flags=flags)
        a, sa = hip.render(scene, p)
        monkeypatch.setenv("RT_HIP_LIBRARY_MATH", "1")
        b, sb = hip.render(scene, p)
        monkeypatch.delenv("RT_HIP_LIBRARY_MATH")
        np.testing.assert_array_equal(a, b)
        assert (sa.ray_queries, sa.hit_queries, sa.shaded_bounces) == (sb.ray_queries, sb.hit_queries, sb.shaded_bounces)


def test_a_near_plane_below_the_short_sequences_bounds(hip, oracle, ffi):
    """A camera whose near plane is closer than 2^-20 (or a scene beyond the documented sizes) is traced by the
    kernels built with the compiler's IEEE sequences: same parity with the CPU path."""
    scene = ffi.Scene.demo(1, 1)
    scene.camera.znear = 5e-7
    for flags, depth in ((ffi.FLAG_DETERMINISTIC, 1), (ffi.FLAG_DETERMINISTIC, 0)):
        p = ffi.params(96, 72, 2, flags=flags, max_depth=depth)
        ref, rs = oracle.render(scene, p)
        out, st = hip.render(scene, p)
        assert rel_err(ref, out).max() < 1e-4
        assert (st.samples, st.ray_queries, st.hit_queries, st.shaded_bounces) == (rs.samples, rs.ray_queries, rs.hit_queries,
                                                                                  rs.shaded_bounces)
