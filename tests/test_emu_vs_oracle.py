"""The trace kernel's per-lane state machine, compiled for the host (tests/emu), against the oracle.

This checks on the CPU what the GPU tests check on the device: the kernel's control flow (stage P / stage S,
filter + exact resolution, multi-light passes, tiled scenes, slices) reproduces the reference's algorithm.
"""
import numpy as np
import pytest

import emu_ffi


def rel_err(ref, out):
    return np.abs(ref - out) / np.maximum(np.abs(ref), 1e-3)


@pytest.mark.parametrize("demo,spheres", [(1, 0), (2, 0), (3, 300)])
@pytest.mark.parametrize("depth", [1, 0])
def test_deterministic_frames_match(oracle, ffi, scenes, demo, spheres, depth):
    scene = scenes(demo, 1, spheres)
    p = ffi.params(64, 48, 2, flags=ffi.FLAG_DETERMINISTIC, max_depth=depth)
    ref, rs = oracle.render(scene, p, 4)
    out, es = emu_ffi.render(scene, p)
    assert rel_err(ref, out).max() < 1e-6  # column sums are added in a different association: ~1e-7
    for key in ("samples", "ray_queries", "hit_queries", "shaded_bounces"):
        assert es[key] == getattr(rs, key), key


@pytest.mark.parametrize("demo", [1, 2])
@pytest.mark.parametrize("column_masks,forward", [(True, False), (False, True), (True, True)])
def test_column_masks_and_forward_accumulation(oracle, ffi, scenes, demo, column_masks, forward):
    """The two shortcuts small scenes take: one primary filter pass per sub-sample column, and path colours
    accumulated as running products instead of the reverse loop over a stack."""
    scene = scenes(demo)
    assert emu_ffi.forward_ok(scene)
    for s, flags in ((2, ffi.FLAG_DETERMINISTIC), (5, ffi.FLAG_DETERMINISTIC), (3, 0)):
        p = ffi.params(64, 48, s, flags=flags, seed=5)
        ref, rs = oracle.render(scene, p, 4)
        out, es = emu_ffi.render(scene, p, column_masks=column_masks, forward=forward)
        if flags:
            assert rel_err(ref, out).max() < 2e-6
            for key in ("samples", "ray_queries", "hit_queries", "shaded_bounces"):
                assert es[key] == getattr(rs, key), key
        else:
            assert np.abs(ref - out).mean() < 1e-5
        if column_masks:  # one filter pass per column instead of one per sample
            assert es["sphere_tests"] < rs.sphere_tests


def test_column_masks_match_per_sample_filtering_bit_for_bit(ffi, scenes):
    scene = scenes(1)
    p = ffi.params(96, 72, 4, seed=3)
    a, sa = emu_ffi.render(scene, p)
    b, sb = emu_ffi.render(scene, p, column_masks=True)
    np.testing.assert_array_equal(a, b)
    assert sa["hit_queries"] == sb["hit_queries"] and sa["ray_queries"] == sb["ray_queries"]


@pytest.mark.parametrize("column_masks,forward,tile", [(False, False, 0), (True, True, 0), (False, True, 16)])
def test_shared_samples_leave_every_bit_of_the_frame_alone(ffi, scenes, column_masks, forward, tile):
    """Samples at the end of a column may be traced by another lane and folded into the column sum afterwards
    (what the kernel does when its queue runs dry); the sum must be the same floating-point sum."""
    for demo, s, flags, slices in ((1, 4, 0, 1), (2, 5, 0, 1), (1, 3, ffi.FLAG_DETERMINISTIC, 1), (1, 4, 0, 3)):
        scene = scenes(demo)
        p = ffi.params(48, 36, s, flags=flags, seed=11, slice_index=slices - 1, slice_count=slices)
        a, sa = emu_ffi.render(scene, p, tile, column_masks=column_masks, forward=forward)
        b, sb = emu_ffi.render(scene, p, tile, column_masks=column_masks, forward=forward, share=True)
        np.testing.assert_array_equal(a, b)
        assert sb["split_columns"] > 0.3 * 48 * 36 * s / slices
        for key in ("samples", "ray_queries", "hit_queries", "shaded_bounces", "exact_tests"):
            assert sa[key] == sb[key], key


@pytest.mark.parametrize("demo,spheres,grid", [(1, 0, False), (2, 0, False), (3, 300, True)])
def test_one_light_loop_equals_the_general_loop(ffi, scenes, demo, spheres, grid):
    """Scenes with one light run a loop that carries no shading state from trip to trip; same frame, bit for bit."""
    scene = scenes(demo, 1, spheres)
    assert scene.light_count == 1
    for flags in (0, ffi.FLAG_DETERMINISTIC):
        p = ffi.params(48, 36, 3, flags=flags, seed=4)
        a, sa = emu_ffi.render(scene, p, forward=True, grid=grid, general_lights=True)
        b, sb = emu_ffi.render(scene, p, forward=True, grid=grid)
        np.testing.assert_array_equal(a, b)
        # (the two loops take the shadow cone's sphere term differently, so a few spheres more or fewer reach the
        # exact intersection; what it finds is the same)
        sa.pop("exact_tests"), sb.pop("exact_tests")
        assert sa == sb


def test_forward_accumulation_is_refused_when_the_energy_break_can_trigger(ffi):
    rough = ffi.Scene.build([(0.0, 0.0, 2.0, 1.0, 0)], [(1, (0.9, 0.9, 0.9), 0.95, 0.1)], [], dict(position=(0, 0, -3), direction=(0, 0, 1)))
    assert not emu_ffi.forward_ok(rough)
    negative = ffi.Scene.build([(0.0, 0.0, 2.0, 1.0, 0)], [(1, (0.9, 0.9, 0.9), 1.5, 0.1)], [], dict(position=(0, 0, -3), direction=(0, 0, 1)))
    assert not emu_ffi.forward_ok(negative)
    absorbing = ffi.Scene.build([(0.0, 0.0, 2.0, 1.0, 0)], [(1, (0.9, 0.9, 0.9), 1.0, 0.1)], [], dict(position=(0, 0, -3), direction=(0, 0, 1)))
    assert emu_ffi.forward_ok(absorbing)  # emissivity 0: such a hit always ends the path


def test_filter_is_conservative_on_the_demo_scenes(oracle, ffi, scenes):
    """Same hit counts as the brute-force oracle: the filter never drops a ray the exact test accepts."""
    for demo in (1, 2):
        scene = scenes(demo)
        p = ffi.params(96, 72, 2, flags=ffi.FLAG_DETERMINISTIC)
        _, rs = oracle.render(scene, p, 8)
        _, es = emu_ffi.render(scene, p)
        assert es["hit_queries"] == rs.hit_queries and es["ray_queries"] == rs.ray_queries
        # ... and lets few candidates through (a sphere a shadow cone touches is intersected with all five rays)
        assert es["exact_tests"] < 0.03 * es["sphere_tests"]


@pytest.mark.parametrize("demo", [1, 2])
def test_stochastic_frames_share_the_random_streams(oracle, ffi, scenes, demo):
    scene = scenes(demo)
    p = ffi.params(64, 48, 2, seed=7)
    ref, rs = oracle.render(scene, p, 4)
    out, es = emu_ffi.render(scene, p)
    err = rel_err(ref, out).max(axis=2)
    assert (err > 1e-3).mean() < 2e-3  # a handful of pixels where rounding flips a branch
    assert np.abs(ref - out).mean() < 1e-5
    assert abs(es["ray_queries"] - rs.ray_queries) <= 1e-3 * rs.ray_queries


@pytest.mark.parametrize("tile", [8, 64, 128])
def test_tiled_scene_equals_resident_scene(ffi, scenes, tile):
    scene = scenes(3, 1, 300)
    p = ffi.params(40, 30, 2, seed=3)
    a, sa = emu_ffi.render(scene, p, 0)
    b, sb = emu_ffi.render(scene, p, tile)
    np.testing.assert_array_equal(a, b)
    assert sa == sb


@pytest.mark.parametrize("slices", [2, 5])
def test_slices(oracle, ffi, scenes, slices):
    scene = scenes(1)
    total = 0.0
    for i in range(slices):
        p = ffi.params(40, 30, 2, seed=9, slice_index=i, slice_count=slices)
        ref, _ = oracle.render(scene, p, 4)
        out, _ = emu_ffi.render(scene, p)
        assert np.abs(ref - out).mean() < 1e-5
        total = total + out.astype(np.float64)
    full, _ = emu_ffi.render(scene, ffi.params(40, 30, 2, seed=9))
    np.testing.assert_allclose(total, full, rtol=2e-6, atol=1e-7)


def small_scene(ffi, lights):
    mats = [(0, (0.7, 0.6, 0.5), 0.7, 0.0), (1, (0.9, 0.8, 0.3), 0.1, 0.15), (2, (0.3, 0.9, 0.8), 0.0, 0.85)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0), (-1.2, -0.5, 2.0, 0.5, 1), (0.0, -0.4, 1.0, 0.6, 2), (1.3, -0.6, 2.5, 0.4, 0)]
    return ffi.Scene.build(spheres, mats, lights, dict(position=(0.0, 0.5, -3.0), direction=(0.0, -0.2, 1.0),
                                                       background=(0.8, 0.85, 0.9)))


@pytest.mark.parametrize("lights", [
    [],
    [((-0.3, -0.9, 0.3), (1.0, 1.0, 1.0)), ((0.5, -0.7, -0.5), (0.4, 0.3, 0.6))],
    [((0.0, -1.0, 0.0), (1.0, 0.5, 0.25)), ((0.6, -0.6, 0.5), (0.2, 0.9, 0.2)), ((-0.6, -0.6, -0.5), (0.5, 0.5, 0.9))],
], ids=["no_light", "two_lights", "three_lights"])
def test_light_counts(oracle, ffi, lights):
    def normalise(d):
        d = np.asarray(d, dtype=np.float64)
        return tuple(d / np.linalg.norm(d))

    scene = small_scene(ffi, [(normalise(d), c) for d, c in lights])
    for flags in (ffi.FLAG_DETERMINISTIC, 0):
        p = ffi.params(48, 36, 2, flags=flags, seed=4)
        ref, rs = oracle.render(scene, p, 4)
        out, es = emu_ffi.render(scene, p)
        if flags:
            assert rel_err(ref, out).max() < 1e-6
            assert es["ray_queries"] == rs.ray_queries
        else:
            assert np.abs(ref - out).mean() < 2e-5


def test_absorbing_metal_and_energy_break(oracle, ffi):
    """A very rough metal (emissivity 0.9 * (1 - 0.95)) drives the reverse loop into its early break."""
    mats = [(1, (0.9, 0.9, 0.9), 0.95, 0.1), (0, (0.5, 0.5, 0.5), 0.7, 0.0)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0), (0.0, 0.0, 2.0, 1.0, 0), (2.1, 0.0, 2.0, 1.0, 0), (-2.1, 0.0, 2.0, 1.0, 1)]
    scene = ffi.Scene.build(spheres, mats, [((0.0, -1.0, 0.0), (1.0, 1.0, 1.0))],
                            dict(position=(0.0, 0.5, -4.0), direction=(0.0, -0.1, 1.0)))
    p = ffi.params(48, 36, 3, seed=8)
    ref, rs = oracle.render(scene, p, 4)
    out, es = emu_ffi.render(scene, p)
    assert np.abs(ref - out).mean() < 2e-5
    assert abs(es["shaded_bounces"] - rs.shaded_bounces) <= 2e-3 * rs.shaded_bounces


@pytest.mark.parametrize("spheres", [100, 1500])
@pytest.mark.parametrize("forward", [False, True])
def test_grid_walk_finds_what_the_linear_scan_finds(oracle, ffi, scenes, spheres, forward):
    """Rays walking the uniform grid: same hits, same counters and (stochastic mode included) the same frame as
    testing every sphere, because every candidate still goes through the exact intersection."""
    scene = scenes(3, 1, spheres)
    for flags, s in ((ffi.FLAG_DETERMINISTIC, 2), (0, 2)):
        p = ffi.params(48, 36, s, flags=flags, seed=6)
        brute, sb = emu_ffi.render(scene, p, forward=forward)
        grid, sg = emu_ffi.render(scene, p, forward=forward, grid=True)
        np.testing.assert_array_equal(grid, brute)
        for key in ("samples", "ray_queries", "hit_queries", "shaded_bounces"):
            assert sg[key] == sb[key], key
        assert sg["exact_tests"] < 0.2 * sb["sphere_tests"]
    ref, rs = oracle.render(scene, ffi.params(48, 36, 2, flags=ffi.FLAG_DETERMINISTIC, seed=6), 4)
    out, es = emu_ffi.render(scene, ffi.params(48, 36, 2, flags=ffi.FLAG_DETERMINISTIC, seed=6), forward=forward, grid=True)
    assert rel_err(ref, out).max() < 2e-6
    assert es["ray_queries"] == rs.ray_queries and es["hit_queries"] == rs.hit_queries


def test_grid_walk_with_rays_along_the_axes(oracle, ffi):
    """Axis-parallel rays (zero direction components) and a camera inside the grid's box."""
    mats = [(0, (0.7, 0.6, 0.5), 0.7, 0.0), (1, (0.9, 0.8, 0.3), 0.1, 0.15), (2, (0.3, 0.9, 0.8), 0.0, 0.85)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0)]
    rng = np.random.default_rng(3)
    for x in range(-4, 5):
        for z in range(0, 9):
            spheres.append((float(x), -0.6, float(z), 0.35 + 0.05 * float(rng.random()), int(rng.integers(0, 3))))
    scene = ffi.Scene.build(spheres, mats, [((0.0, -1.0, 0.0), (1.0, 1.0, 1.0))],
                            dict(position=(0.0, -0.6, 4.5), direction=(0.0, 0.0, 1.0), background=(0.8, 0.85, 0.9)))
    p = ffi.params(33, 33, 1, flags=ffi.FLAG_DETERMINISTIC)  # the centre pixel looks straight down +z
    ref, rs = oracle.render(scene, p, 2)
    out, es = emu_ffi.render(scene, p, grid=True)
    assert rel_err(ref, out).max() < 2e-6
    assert es["hit_queries"] == rs.hit_queries


def long_row_scene(ffi, count=500):
    """A row of `count` small spheres along x, seen at a grazing angle from one end: grid walks of hundreds of
    cells, far from the origin of the walk."""
    mats = [(0, (0.6, 0.6, 0.6), 0.7, 0.0), (1, (0.9, 0.7, 0.3), 0.1, 0.1), (2, (0.8, 0.9, 0.9), 0.0, 0.85)]
    spheres = [(0.0, -1000.4, 0.0, 1000.0, 0)]
    for i in range(count):
        spheres.append((float(i) * 0.9, 0.05 * ((i * 7) % 5), 0.3 * ((i * 3) % 4 - 1.5), 0.3 + 0.02 * (i % 3), i % 3))
    light = np.array([-0.3, -1.0, 0.2]) / np.linalg.norm([-0.3, -1.0, 0.2])
    return ffi.Scene.build(spheres, mats, [(tuple(light), (1.0, 1.0, 1.0))],
                           dict(position=(-3.0, 0.8, 0.1), direction=(1.0, -0.004, 0.0), background=(0.8, 0.85, 0.9)))


def test_grid_walk_over_hundreds_of_cells(ffi):
    """Long walks at a grazing angle: the wall parameters must not drift away from the cells the spheres are listed
    in.  Same frame and counters as the linear scan, bit for bit."""
    scene = long_row_scene(ffi)
    for flags, seed in ((ffi.FLAG_DETERMINISTIC, 1), (0, 3)):
        p = ffi.params(64, 24, 2, flags=flags, seed=seed)
        brute, sb = emu_ffi.render(scene, p, forward=True)
        grid, sg = emu_ffi.render(scene, p, forward=True, grid=True)
        np.testing.assert_array_equal(grid, brute)
        for key in ("samples", "ray_queries", "hit_queries", "shaded_bounces"):
            assert sg[key] == sb[key]
        assert sb["hit_queries"] > 0.3 * sb["samples"]  # the row is in view


@pytest.mark.parametrize("demo,spheres", [(1, 0), (2, 0), (3, 300)])
def test_shadow_cones_change_no_bit(ffi, scenes, demo, spheres):
    """Testing the cone that holds the five shadow rays first, or intersecting them with every sphere: same frame,
    same query counters, far fewer exact intersections."""
    scene = scenes(demo, 1, spheres)
    for flags, seed in ((ffi.FLAG_DETERMINISTIC, 1), (0, 9)):
        p = ffi.params(48, 36, 2, flags=flags, seed=seed)
        a, sa = emu_ffi.render(scene, p, forward=True)
        b, sb = emu_ffi.render(scene, p, forward=True, cones=False)
        np.testing.assert_array_equal(a, b)
        assert sa["exact_tests"] < sb["exact_tests"]
        sa.pop("exact_tests"), sb.pop("exact_tests")
        assert sa == sb


def _two_groups(ffi, gap, seed):
    """Two groups of small spheres `gap` apart, the camera beside one looking across both: every sphere and every
    hit point is far from the scene's centre, which lies in the empty middle."""
    g = np.random.default_rng(seed)
    scale = max(1.0, gap / 400.0)
    mats = [(0, (0.6, 0.6, 0.6), 0.7, 0.0)]
    spheres = [(0.0, -1001.0, 0.0, 1000.0, 0)]
    for side in (-0.5, 0.5):
        for _ in range(12):
            kind = int(g.integers(0, 3))
            mats.append((kind, tuple(float(x) for x in g.uniform(0.1, 1.0, 3)), float(g.uniform(0.0, 0.9)) if kind == 1 else 0.7,
                         float(g.uniform(0.8, 0.9)) if kind == 2 else float(g.uniform(0.0, 0.2))))
            rad = float(g.uniform(0.1, 0.5)) * scale
            spheres.append((side * gap + float(g.uniform(-3, 3)) * scale, rad - 1.0 + float(g.uniform(0, 1.5)),
                            float(g.uniform(-3, 3)) * scale, rad, len(mats) - 1))
    lights = [((0.3, -0.9, 0.2), (0.9, 0.9, 0.8))]
    return ffi.Scene.build(spheres, mats, lights, dict(position=(-0.5 * gap - 8.0 * scale, 1.5, -6.0 * scale),
                                                        direction=(0.98, -0.05, 0.2), background=(0.8, 0.85, 0.9)))


@pytest.mark.parametrize("gap", [60.0, 900.0, 3000.0])
def test_shade_filter_far_from_the_scene_centre(oracle, ffi, gap):
    """The shade filter multiplies scene-centred coordinates of sphere and hit point; where both are large against
    the distance between them its explicit slack has to cover the rounding.  Same hits as the brute-force oracle."""
    scene = _two_groups(ffi, gap, 7)
    for depth in (1, 0):
        p = ffi.params(64, 36, 2, flags=ffi.FLAG_DETERMINISTIC, max_depth=depth)
        ref, rs = oracle.render(scene, p, 8)
        for forward in (False, True):
            out, es = emu_ffi.render(scene, p, column_masks=True, forward=forward)
            assert rel_err(ref, out).max() < 1e-5
            for key in ("samples", "ray_queries", "hit_queries", "shaded_bounces"):
                assert es[key] == getattr(rs, key), key
    assert rs.hit_queries > 0.1 * rs.samples  # the spheres are in view


def test_view_direction_exactly_opposite_the_light(oracle, ffi):
    """The one place the kernel's arithmetic departs from the reference's on purpose: a light that shines exactly
    along the viewing ray, towards the camera, makes the half vector of DirectionalLight::GetSpecular
    (ray_tracing/directional_light.cpp:42) the normalisation of a zero vector.  glm's clamp and std::min hand the NaN
    through, so the reference (and the oracle) give that sample a NaN colour; the kernel clamps with the hardware's
    median-of-three and minimum, which give the lower bound for a NaN: the specular term of that hit is 0 and the
    sample is finite."""
    import ctypes as C

    lib = oracle.load()
    camera = dict(position=(0.0, 0.0, -5.0), direction=(0.1, -0.2, 1.0))
    probe = ffi.Scene.build([], [], [], camera)
    d = oracle.F3()
    lib.orc_primary_direction(C.byref(probe.camera), 1, 1, 1, 0, 0, 0, 0, C.byref(d))
    rd = np.array(list(d), dtype=np.float32)
    centre = np.float32([0.0, 0.0, -5.0]) + rd * np.float32(4.0)  # a sphere the one ray of the frame hits head-on
    scene = ffi.Scene.build([(float(centre[0]), float(centre[1]), float(centre[2]), 1.0, 0)], [(0, (0.6, 0.5, 0.4), 0.7, 0.0)],
                            [(tuple(float(-x) for x in rd), (1.0, 1.0, 1.0))], camera)
    p = ffi.params(1, 1, 1, flags=ffi.FLAG_DETERMINISTIC, max_depth=1)
    ref, rs = oracle.render(scene, p, 1)
    out, es = emu_ffi.render(scene, p, forward=True)
    assert rs.hit_queries >= 1 and es["shaded_bounces"] == rs.shaded_bounces == 1
    assert np.isnan(ref).all()
    # the surface faces away from the light (no diffuse term) and the specular term is 0: half the albedo
    np.testing.assert_allclose(out[0, 0], 0.5 * np.float32([0.6, 0.5, 0.4]), rtol=1e-6)
