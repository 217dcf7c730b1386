"""Size-independent properties of the oracle's frame: what must hold whatever the scene."""
import numpy as np
import pytest


def test_thread_count_does_not_change_the_frame(oracle, ffi, scenes):
    p = ffi.params(48, 31, 2, seed=5)
    a, sa = oracle.render(scenes(1), p, 1)
    b, sb = oracle.render(scenes(1), p, 7)
    np.testing.assert_array_equal(a, b)
    assert sa.as_dict() == sb.as_dict()


def test_more_threads_than_rows(oracle, ffi, scenes):
    p = ffi.params(16, 3, 1, seed=5)
    a, _ = oracle.render(scenes(2), p, 1)
    b, _ = oracle.render(scenes(2), p, 64)
    np.testing.assert_array_equal(a, b)


def test_empty_scene_is_background(oracle, ffi):
    scene = ffi.Scene.build([], [], [], dict(position=(0, 0, 0), direction=(0, 0, 1), background=(0.2, 0.5, 0.9)))
    rgb, st = oracle.render(scene, ffi.params(9, 7, 3), 2)
    np.testing.assert_allclose(rgb, np.broadcast_to(np.float32([0.2, 0.5, 0.9]), rgb.shape), rtol=1e-6)
    assert st.samples == 9 * 7 * 9 and st.ray_queries == st.samples and st.hit_queries == 0


@pytest.mark.parametrize("slices", [2, 3, 8])
def test_slices_add_up_to_the_frame(oracle, ffi, scenes, slices):
    scene = scenes(1)
    full, fs = oracle.render(scene, ffi.params(40, 30, 2, seed=9), 4)
    total = np.zeros_like(full, dtype=np.float64)
    samples = 0
    for i in range(slices):
        part, st = oracle.render(scene, ffi.params(40, 30, 2, seed=9, slice_index=i, slice_count=slices), 4)
        total += part
        samples += st.samples
    assert samples == fs.samples
    np.testing.assert_allclose(total, full, rtol=2e-6, atol=1e-7)


def test_seed_changes_the_noise_not_the_mean(oracle, ffi, scenes):
    scene = scenes(1)
    a, _ = oracle.render(scene, ffi.params(64, 48, 3, seed=1), 8)
    b, _ = oracle.render(scene, ffi.params(64, 48, 3, seed=2), 8)
    assert not np.array_equal(a, b)
    assert abs(float(a.mean()) - float(b.mean())) < 2e-3


def test_deterministic_mode_ignores_the_seed(oracle, ffi, scenes):
    scene = scenes(2)
    a, _ = oracle.render(scene, ffi.params(32, 24, 2, flags=ffi.FLAG_DETERMINISTIC, seed=1), 4)
    b, _ = oracle.render(scene, ffi.params(32, 24, 2, flags=ffi.FLAG_DETERMINISTIC, seed=77), 4)
    np.testing.assert_array_equal(a, b)


def test_depth_one_is_primary_hit_plus_direct_light(oracle, ffi, scenes):
    """max_depth 1: a hit pixel is 0.5 * (albedo + diffuse + specular); sky pixels are the background."""
    scene = scenes(1)
    rgb, st = oracle.render(scene, ffi.params(64, 48, 1, flags=ffi.FLAG_DETERMINISTIC, max_depth=1), 4)
    np.testing.assert_allclose(rgb[0, 0], [0.9, 0.9, 0.9], rtol=1e-6)  # top-left looks at the sky
    # one primary query per sample, five shadow queries per shaded hit and no scattered ray
    assert st.ray_queries == st.samples + 5 * st.shaded_bounces
    assert rgb.min() >= 0.0


def test_counters_are_consistent(oracle, ffi, scenes):
    scene = scenes(1)
    _, st = oracle.render(scene, ffi.params(64, 48, 2, seed=3), 4)
    assert st.samples == 64 * 48 * 4
    assert st.sphere_tests == st.ray_queries * scene.sphere_count
    # every sample asks one primary query; every shaded bounce asks five shadow queries and at most one more
    assert st.samples + 5 * st.shaded_bounces <= st.ray_queries <= st.samples + 6 * st.shaded_bounces
