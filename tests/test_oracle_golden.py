"""The oracle against the golden vectors of tests/golden (authored here: the reference has no fixtures for this
path, so parity is "unpinned" by the reference and pinned by these hand-checkable known answers instead)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "analytic.json")) as f:
    ANALYTIC = json.load(f)

TOL = 2e-6  # float32 restatement against float64 known answers


def ids(cases):
    return [c.get("name", str(i)) for i, c in enumerate(cases)]


@pytest.mark.parametrize("case", ANALYTIC["philox"], ids=["zeros", "ones", "pi"])
def test_philox_known_answers(oracle, case):
    lib = oracle.load()
    out = (C.c_uint32 * 4)()
    lib.orc_philox4x32_10(C.byref((C.c_uint32 * 4)(*case["ctr"])), C.byref((C.c_uint32 * 2)(*case["key"])), C.byref(out))
    assert list(out) == case["out"]  # bit-exact


@pytest.mark.parametrize("case", ANALYTIC["sphere"], ids=ids(ANALYTIC["sphere"]))
def test_sphere_trace_nearest(oracle, case):
    lib = oracle.load()
    t = C.c_float()
    pos, nrm = oracle.F3(), oracle.F3()
    hit = lib.orc_sphere_trace_nearest(C.byref(oracle.f3(case["origin"])), C.byref(oracle.f3(case["dir"])),
                                       C.byref(oracle.f3(case["center"])), case["radius"], case["tmin"], case["tmax"],
                                       C.byref(t), C.byref(pos), C.byref(nrm))
    assert bool(hit) == case["hit"]
    if case["hit"]:
        # the ground sphere loses digits to |d|^2 - r^2 in float32, exactly as the reference does
        tol = 2e-4 if case["name"] == "ground_sphere" else TOL
        assert t.value == pytest.approx(case["t"], rel=tol)
        np.testing.assert_allclose(list(pos), case["pos"], rtol=tol, atol=tol)
        np.testing.assert_allclose(list(nrm), case["normal"], rtol=tol, atol=tol)


@pytest.mark.parametrize("case", ANALYTIC["scatter"], ids=ids(ANALYTIC["scatter"]))
def test_scatter(oracle, ffi, case):
    lib = oracle.load()
    m = ffi.Material()
    m.type = case["type"]
    m.albedo[:] = case["albedo"]
    m.roughness, m.refraction = case["roughness"], case["refraction"]
    att, out = oracle.F3(), oracle.F3()
    e = C.c_float()
    lib.orc_scatter(C.byref(m), C.byref(oracle.f3(case["ray_dir"])), C.byref(oracle.f3(case["normal"])),
                    C.byref(oracle.f3(case["ball"])), case["xi"], C.byref(att), C.byref(out), C.byref(e))
    np.testing.assert_allclose(list(att), case["albedo"], rtol=1e-7)
    np.testing.assert_allclose(list(out), case["out_dir"], rtol=TOL, atol=TOL)
    assert e.value == pytest.approx(case["emissivity"], rel=TOL, abs=1e-9)


@pytest.mark.parametrize("case", ANALYTIC["specular"], ids=ids(ANALYTIC["specular"]))
def test_get_specular(oracle, ffi, case):
    lib = oracle.load()
    light = ffi.Light()
    light.direction[:] = case["light_dir"]
    light.color[:] = case["light_color"]
    m = ffi.Material()
    m.albedo[:] = case["albedo"]
    m.roughness, m.refraction = case["roughness"], case["refraction"]
    out = oracle.F3()
    lib.orc_get_specular(C.byref(light), C.byref(m), C.byref(oracle.f3(case["ray_dir"])),
                         C.byref(oracle.f3(case["normal"])), C.byref(out))
    # the Beckmann exponent divides by roughness^2 (down to 9e-4): input rounding is amplified ~1e3 times
    np.testing.assert_allclose(list(out), case["color"], rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("case", ANALYTIC["accumulate"], ids=ids(ANALYTIC["accumulate"]))
def test_accumulate(oracle, case):
    lib = oracle.load()
    l = np.asarray(case["l"], dtype=np.float32)
    e = np.asarray(case["e"], dtype=np.float32)
    out = oracle.F3()
    lib.orc_accumulate(len(e), l.ctypes.data, e.ctypes.data, C.byref(out))
    np.testing.assert_allclose(list(out), case["color"], rtol=TOL, atol=1e-12)


def test_accumulate_closed_form(oracle):
    """Without the early break the loop equals 0.5 * prod(l_i) * prod(e_k ** k)."""
    lib = oracle.load()
    rng = np.random.default_rng(3)
    for n in range(1, 11):
        l = rng.uniform(0.2, 1.5, size=(n, 3)).astype(np.float32)
        e = rng.uniform(0.7, 1.0, size=n).astype(np.float32)
        out = oracle.F3()
        lib.orc_accumulate(n, l.ctypes.data, e.ctypes.data, C.byref(out))
        expect = 0.5 * np.prod(l.astype(np.float64), axis=0) * np.prod(e.astype(np.float64) ** np.arange(n))
        np.testing.assert_allclose(list(out), expect, rtol=5e-6)


@pytest.mark.parametrize("case", ANALYTIC["primary"], ids=ids(ANALYTIC["primary"]))
def test_primary_direction(oracle, ffi, case):
    lib = oracle.load()
    scene = ffi.Scene.build([], [], [], dict(position=case["position"], direction=case["direction"]))
    out = oracle.F3()
    lib.orc_primary_direction(C.byref(scene.camera), case["width"], case["height"], case["samples"], case["x"],
                              case["y"], case["sx"], case["sy"], C.byref(out))
    np.testing.assert_allclose(list(out), case["dir"], rtol=TOL, atol=TOL)


def test_trace_light_zero_jitter(oracle, ffi):
    """DirectionalLight::TraceLight with zero jitter: unoccluded matte, occluded matte (black), occluded glass
    (tinted by its own material) -- ray_tracing/directional_light.cpp:22-32."""
    lib = oracle.load()
    mats = [(0, (0.6, 0.5, 0.4), 0.7, 0.0), (2, (0.2, 0.8, 0.6), 0.0, 0.85)]
    # sphere 0: the occluder, straight above the origin along -light direction
    ldir = np.array([0.0, -1.0, 0.0])
    scene = ffi.Scene.build([(0.0, 5.0, 0.0, 1.0, 0)], mats, [(tuple(ldir), (1.0, 0.9, 0.8))],
                            dict(position=(0, 0, -5), direction=(0, 0, 1)))
    s = oracle.orc_scene(scene)
    jitter = np.zeros((5, 3), dtype=np.float32)
    n = oracle.f3((0.0, 1.0, 0.0))
    out = oracle.F3()
    # unoccluded point far to the side
    lib.orc_trace_light(C.byref(s), scene.lights, scene.materials, C.byref(oracle.f3((10.0, 0.0, 0.0))), C.byref(n),
                        jitter.ctypes.data, C.byref(out), None)
    np.testing.assert_allclose(list(out), [1.0, 0.9, 0.8], rtol=1e-6)
    # occluded, matte: refraction 0 -> black
    lib.orc_trace_light(C.byref(s), scene.lights, scene.materials, C.byref(oracle.f3((0.0, 0.0, 0.0))), C.byref(n),
                        jitter.ctypes.data, C.byref(out), None)
    np.testing.assert_allclose(list(out), [0.0, 0.0, 0.0], atol=1e-12)
    # occluded, glass: refraction * 0.75 * mix(albedo, light colour, 0.75) * cos
    glass = C.byref(scene.materials, C.sizeof(ffi.Material))
    lib.orc_trace_light(C.byref(s), scene.lights, C.cast(glass, C.POINTER(ffi.Material)),
                        C.byref(oracle.f3((0.0, 0.0, 0.0))), C.byref(n), jitter.ctypes.data, C.byref(out), None)
    tint = 0.85 * 0.75 * (0.25 * np.array([0.2, 0.8, 0.6]) + 0.75 * np.array([1.0, 0.9, 0.8]))
    np.testing.assert_allclose(list(out), tint, rtol=1e-6)


def test_uniform_and_ball(oracle):
    lib = oracle.load()
    assert lib.orc_uniform(0) == 0.0
    assert lib.orc_uniform(0xFFFFFFFF) == pytest.approx(1.0 - 2.0 ** -24, rel=0, abs=0)
    assert lib.orc_uniform(0x80000000) == 0.5
    rng = np.random.default_rng(11)
    u = rng.random((20000, 3)).astype(np.float32)
    pts = np.empty((len(u), 3), dtype=np.float64)
    out = oracle.F3()
    for i, (a, b, c) in enumerate(u):
        lib.orc_ball(0.25, float(a), float(b), float(c), C.byref(out))
        pts[i] = list(out)
    r = np.linalg.norm(pts, axis=1)
    assert r.max() <= 0.25 * (1 + 1e-6)
    # uniform in a ball: P(r < x) = (x/R)^3 -> E[r] = 3R/4, E[r^2] = 3R^2/5; direction isotropic
    assert r.mean() == pytest.approx(0.75 * 0.25, rel=0.01)
    assert (r ** 2).mean() == pytest.approx(0.6 * 0.25 ** 2, rel=0.02)
    np.testing.assert_allclose(pts.mean(axis=0), 0.0, atol=0.004)
    np.testing.assert_allclose((pts ** 2).mean(axis=0), 0.25 ** 2 / 5.0, rtol=0.04)
    # against glm::ballRand's own method (rejection from the cube): same radial law
    cube = rng.uniform(-0.25, 0.25, size=(60000, 3))
    rej = np.linalg.norm(cube[np.linalg.norm(cube, axis=1) <= 0.25], axis=1)
    qs = np.linspace(0.05, 0.95, 10)
    np.testing.assert_allclose(np.quantile(r, qs), np.quantile(rej, qs), rtol=0.02)


@pytest.mark.parametrize("name,demo,spheres", [("pretty", 1, 0), ("glass", 2, 0), ("synthetic300", 3, 300)])
@pytest.mark.parametrize("mode,depth", [("d1", 1), ("d2", 0)])
def test_deterministic_frames_are_pinned(oracle, ffi, scenes, name, demo, spheres, mode, depth):
    golden = np.load(os.path.join(GOLDEN, "frame_%s_%s.npy" % (name, mode)))
    p = ffi.params(64, 48, 2, flags=ffi.FLAG_DETERMINISTIC, seed=1, max_depth=depth)
    rgb, _ = oracle.render(scenes(demo, 1, spheres), p, 3)
    np.testing.assert_array_equal(rgb, golden)


@pytest.mark.parametrize("demo,spheres", [(1, 0), (2, 0), (3, 120)])
def test_depth_one_frames_against_an_independent_numpy_statement(oracle, ffi, scenes, demo, spheres):
    """The oracle's D1 frame (primary hit + direct light + specular, no random draws) against a float64 numpy
    rendering written separately from the reference's text (tests/numpy_d1.py).  float32 against float64: a few
    1e-6 apart, except where a sample sits on an edge and the two roundings disagree about a hit."""
    import numpy_d1

    scene = scenes(demo, 1, spheres or 10000)
    w, h, s = 48, 36, 2
    ref = numpy_d1.render_d1(scene, w, h, s)
    out, _ = oracle.render(scene, ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC, max_depth=1), 4)
    err = np.abs(out - ref).max(axis=2) / np.maximum(np.abs(ref).max(axis=2), 1e-3)
    assert (err > 1e-4).mean() < 0.01, "pixels off: %d" % int((err > 1e-4).sum())
    assert np.median(err) < 1e-5
    assert abs(float(out.mean()) - float(ref.mean())) < 1e-3


@pytest.mark.parametrize("demo,spheres", [(1, 0), (2, 0), (3, 120)])
def test_full_depth_deterministic_frames_against_numpy(oracle, ffi, scenes, demo, spheres):
    """The bounce loop, the three materials' deterministic scattering (glass refraction included) and the reverse
    accumulation, against the separate numpy statement.  Deep glass paths amplify float32 rounding, so the bar is
    on the bulk of the pixels and on the mean."""
    import numpy_d1

    scene = scenes(demo, 1, spheres or 10000)
    w, h, s = 40, 30, 2
    ref = numpy_d1.render_full_depth(scene, w, h, s)
    out, _ = oracle.render(scene, ffi.params(w, h, s, flags=ffi.FLAG_DETERMINISTIC), 4)
    err = np.abs(out - ref).max(axis=2) / np.maximum(np.abs(ref).max(axis=2), 1e-3)
    assert (err > 1e-3).mean() < 0.03, "pixels off: %d of %d" % (int((err > 1e-3).sum()), err.size)
    assert np.median(err) < 1e-5
    assert abs(float(out.mean()) - float(ref.mean())) < 2e-3
    # and depth 1 of the same function is the depth-1 statement
    np.testing.assert_allclose(numpy_d1.render_full_depth(scene, 16, 12, 2, max_depth=1),
                               numpy_d1.render_d1(scene, 16, 12, 2), rtol=1e-9, atol=1e-12)
