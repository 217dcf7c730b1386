"""The C++ host: object model, demo scenes and their flattening (librt_host, include/rt_host.h)."""
import numpy as np
import pytest


def test_demo_scenes_have_the_reference_shape(scenes):
    for demo in (1, 2):
        s = scenes(demo)
        assert s.sphere_count == 41 and s.light_count == 1  # 40 + ground (demos/pretty_spheres.hpp:58-83)
        np.testing.assert_allclose(s.centers_radii[40], [0.0, -1001.0, 0.0, 1000.0])
        r = s.centers_radii[:40, 3]
        assert r.min() >= 0.25 and r.max() <= 0.5
        np.testing.assert_allclose(s.centers_radii[:40, 1], r - 1.0, rtol=1e-6)  # resting on y = -1
        xz = {(float(x), float(z)) for x, _, z, _ in s.centers_radii[:40]}
        assert len(xz) == 40 and all(x == round(x) and -5 <= x <= 5 and z == round(z) and -5 <= z <= 5 for x, z in xz)
        d = np.array(list(s.lights[0].direction))
        np.testing.assert_allclose(d, np.array([-0.4, -1.0, 0.6]) / np.linalg.norm([-0.4, -1.0, 0.6]), rtol=1e-6)
        assert np.linalg.norm(list(s.camera.direction)) == pytest.approx(1.0, rel=1e-6)
        assert s.camera.fov == pytest.approx(2.0 * np.pi / 3.0, rel=1e-6)
        assert (s.camera.znear, s.camera.zfar) == (pytest.approx(0.001), 1000.0)


def test_material_getters_are_flattened_as_the_reference_does(scenes):
    pretty = scenes(1)
    mats = [pretty.materials[i] for i in range(pretty.material_count)]
    kinds = [mats[i].type for i in pretty.material_index]
    assert kinds[:15] == [0] * 15 and kinds[15:40] == [1] * 25 and kinds[40] == 0  # 15 matte, 25 metal, ground
    for m in mats:
        if m.type == 0:  # Matte keeps Material's defaults (ray_tracing/material.hpp:25-26)
            assert m.roughness == pytest.approx(0.7) and m.refraction == 0.0
        else:
            assert 0.0 <= m.roughness <= 0.2 and 0.0 <= m.refraction <= 0.2
    glass = scenes(2)
    for i in range(40):
        m = glass.materials[glass.material_index[i]]
        assert m.type == 2 and m.roughness == 0.0 and 0.8 <= m.refraction <= 0.9
    ground = glass.materials[glass.material_index[40]]
    assert ground.type == 0 and list(ground.albedo) == pytest.approx([0.4, 0.4, 0.4])


def test_scenes_are_reproducible_and_seeded(ffi):
    a, b, c = ffi.Scene.demo(1, 1), ffi.Scene.demo(1, 1), ffi.Scene.demo(1, 2)
    np.testing.assert_array_equal(a.centers_radii, b.centers_radii)
    assert not np.array_equal(a.centers_radii, c.centers_radii)


def test_synthetic_field(ffi):
    s = ffi.Scene.demo(3, 1, 10000)
    assert s.sphere_count == 10001 and s.light_count == 1
    c = s.centers_radii[:10000]
    assert c[:, 3].min() >= 0.25 and c[:, 3].max() <= 0.5
    # one sphere per unit cell, inside its cell: no two can overlap
    cells = np.floor(c[:, [0, 2]] + 50.0).astype(int)
    assert len({(int(x), int(z)) for x, z in cells}) == 10000
    kinds = np.array([s.materials[i].type for i in s.material_index[:10000]])
    assert set(kinds) == {0, 1, 2} and min(np.bincount(kinds)) > 3000


def test_update_texture_matches_the_viewer(ffi):
    rgb = np.array([[-0.5, 0.0, 0.25], [0.5, 0.999, 1.0], [1.5, 0.0039, 0.9961]], dtype=np.float32)
    out = ffi.update_texture(rgb)
    expect = (255 * np.clip(rgb, 0.0, 1.0)).astype(np.uint8)  # truncation, app.cpp:132-134
    np.testing.assert_array_equal(out[:, :3], expect)
    assert (out[:, 3] == 255).all()


def test_sharding_rule():
    from rt_amd import sharding

    for s_row, count in ((2, 8), (16, 8), (3, 2), (10, 4)):
        owners = [sharding.owner_of(p, sx, sy, s_row, count) for p in range(64) for sx in range(s_row)
                  for sy in range(s_row)]
        per = np.bincount(owners, minlength=count)
        assert per.max() - per.min() <= 1  # balanced whatever divides what


def test_scene_file_round_trip(ffi, scenes, tmp_path):
    """Saving a flattened demo and loading it back through the C++ parser gives the same arrays, bit for bit
    (material de-duplication may renumber nothing: the file already lists unique materials)."""
    for demo in (1, 2):
        scene = scenes(demo)
        path = tmp_path / ("demo%d.rtscene" % demo)
        scene.save(str(path))
        back = ffi.Scene.load(str(path))
        np.testing.assert_array_equal(back.centers_radii, scene.centers_radii)
        np.testing.assert_array_equal(back.material_index, scene.material_index)
        assert back.material_count == scene.material_count and back.light_count == scene.light_count
        for i in range(scene.material_count):
            a, b = scene.materials[i], back.materials[i]
            assert (a.type, list(a.albedo), a.roughness, a.refraction) == (b.type, list(b.albedo), b.roughness, b.refraction)
        for i in range(scene.light_count):
            assert list(scene.lights[i].direction) == list(back.lights[i].direction)
            assert list(scene.lights[i].color) == list(back.lights[i].color)
        for field in ("position", "direction", "background"):
            np.testing.assert_allclose(list(getattr(back.camera, field)), list(getattr(scene.camera, field)), rtol=3e-7)  # the direction is normalised again on load
        assert (back.camera.fov, back.camera.znear, back.camera.zfar) == (scene.camera.fov, scene.camera.znear, scene.camera.zfar)
    # a lens other than the reference's fixed one survives the file too
    scene = scenes(1)
    import copy
    odd = ffi.Scene(scene.centers_radii, scene.material_index, scene.materials, scene.lights, copy.copy(scene.camera))
    odd.material_count, odd.light_count = scene.material_count, scene.light_count
    odd.camera.fov, odd.camera.znear, odd.camera.zfar = 1.25, 0.01, 250.0
    path = tmp_path / "lens.rtscene"
    odd.save(str(path))
    back = ffi.Scene.load(str(path))
    assert (back.camera.fov, back.camera.znear, back.camera.zfar) == (odd.camera.fov, odd.camera.znear, odd.camera.zfar)
    path.write_text("rt-scene 1\nlens 1.0 0.5 0.1\n")
    with pytest.raises(ValueError, match="lens needs"):
        ffi.Scene.load(str(path))


def test_scene_file_errors_name_the_line(ffi, tmp_path):
    path = tmp_path / "bad.rtscene"
    path.write_text("rt-scene 1\nmaterial a matte 1 1 1\nsphere 0 0 0 1 b\n")
    with pytest.raises(ValueError, match=r"bad.rtscene:3: unknown material 'b'"):
        ffi.Scene.load(str(path))
    path.write_text("# nothing here\n")
    with pytest.raises(ValueError, match="empty file"):
        ffi.Scene.load(str(path))
    path.write_text("scene\n")
    with pytest.raises(ValueError, match="header"):
        ffi.Scene.load(str(path))


def test_object_model_ray_queries_match_the_oracle(oracle, ffi, scenes):
    """Sphere::Trace / TraceNearest and the two collection queries of the host object model
    (ray_tracing/sphere.cpp:11-81, hitable_object.hpp:53-87) against the oracle's restatement, bit for bit, on
    random rays through the demo scenes and on the analytic edge cases."""
    import ctypes as C

    import numpy as np

    lib = oracle.load()
    rng = np.random.default_rng(11)
    for demo, spheres in ((1, 0), (2, 0), (3, 200)):
        scene = scenes(demo, 1, spheres)
        s = oracle.orc_scene(scene)
        for _ in range(150):
            origin = rng.uniform(-6, 6, 3) * [1, 0.3, 1] + [0, 1.5, 0]
            direction = rng.normal(size=3)
            if rng.random() < 0.5:
                direction /= np.linalg.norm(direction)
            tmin, tmax = (0.001, 1000.0) if rng.random() < 0.7 else (float(rng.uniform(0, 5)), float(rng.uniform(5, 30)))
            o, d = oracle.f3(origin), oracle.f3(direction)
            ref = (oracle.OrcHit * 64)()
            n_ref = lib.orc_trace_all(C.byref(s), C.byref(o), C.byref(d), tmin, tmax, ref, 64)
            n, hits = ffi.trace_hits(scene, origin, direction, tmin, tmax)
            assert n == n_ref
            for a, b in zip(hits, ref):
                assert a.t == b.t and tuple(a.position) == tuple(b.position) and tuple(a.normal) == tuple(b.normal)
            # hits with equal t may come in either order (std::sort is not stable): compare spheres as sets per t
            assert sorted((h.t, h.sphere) for h in hits) == sorted((ref[i].t, ref[i].sphere) for i in range(min(n, 64)))
            t, pos, nrm = C.c_float(), oracle.F3(), oracle.F3()
            i_ref = lib.orc_trace_nearest(C.byref(s), C.byref(o), C.byref(d), tmin, tmax, C.byref(t), C.byref(pos),
                                          C.byref(nrm))
            n1, near = ffi.trace_hits(scene, origin, direction, tmin, tmax, nearest_only=True)
            assert n1 == (1 if i_ref >= 0 else 0)
            if n1:
                assert near[0].sphere == i_ref and near[0].t == t.value
                assert tuple(near[0].position) == tuple(pos) and tuple(near[0].normal) == tuple(nrm)
    # analytic cases: head-on (two hits, nearest at 4), origin inside (far root only; TraceNearest misses), behind
    one = ffi.Scene.build([(0.0, 0.0, 5.0, 1.0, 0)], [(0, (0.5, 0.5, 0.5), 0.7, 0.0)], [],
                          dict(position=(0, 0, 0), direction=(0, 0, 1)))
    n, hits = ffi.trace_hits(one, (0, 0, 0), (0, 0, 1), 0.001, 1000.0)
    assert n == 2 and (hits[0].t, hits[1].t) == (4.0, 6.0) and tuple(hits[0].normal) == (0.0, 0.0, -1.0)
    n, hits = ffi.trace_hits(one, (0, 0, 5), (0, 0, 1), 0.001, 1000.0)
    assert n == 1 and hits[0].t == 1.0
    assert ffi.trace_hits(one, (0, 0, 5), (0, 0, 1), 0.001, 1000.0, nearest_only=True)[0] == 0
    assert ffi.trace_hits(one, (0, 0, 0), (0, 0, -1), 0.001, 1000.0)[0] == 0
    n, hits = ffi.trace_hits(one, (1.0, 0, 0), (0, 0, 1), 0.001, 1000.0)  # tangent: one hit at -b / 2a
    assert n == 1 and hits[0].t == 5.0
