"""Writes the golden vectors of tests/golden/.

Two kinds of file:

* analytic.json -- known answers for the building blocks of the trace loop, computed HERE in float64 from the
  formulas of the reference's source (file:line in each entry), independently of oracle/rt_oracle.c.  The
  reference holds no fixtures of its own for this path, so these are authored by this repository ("parity
  unpinned" by the reference; pinned by hand-checkable maths instead).  The Philox vectors are the published
  Random123 known-answer tests.
* frame_*.npy -- deterministic-mode frames rendered by the oracle (64x48, 2x2), kept as regression pins for
  the oracle and as fixed targets for the HIP backend.  Regenerate only on purpose.

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def norm(v):
    v = np.asarray(v, dtype=np.float64)
    return v / math.sqrt(float(v @ v))


def sphere_cases():
    # Sphere::TraceNearest, ray_tracing/sphere.cpp:63-81
    cases = []

    def add(name, o, d, c, r, tmin, tmax):
        o, d, c = (np.asarray(x, dtype=np.float64) for x in (o, d, c))
        dd = o - c
        a = float(d @ d)
        b = 2.0 * float(dd @ d)
        cc = float(dd @ dd) - r * r
        disc = b * b - 4 * a * cc
        hit, t, p, n = False, 0.0, [0, 0, 0], [0, 0, 0]
        if disc >= 0.0:
            s = math.sqrt(disc)
            t = min((-b - s) / (2 * a), (-b + s) / (2 * a))
            if tmin <= t <= tmax:
                hit = True
                p = o + d * t
                n = norm(p - c)
        cases.append(dict(name=name, origin=list(map(float, o)), dir=list(map(float, d)), center=list(map(float, c)),
                          radius=r, tmin=tmin, tmax=tmax, hit=hit, t=float(t), pos=list(map(float, p)),
                          normal=list(map(float, n))))

    add("head_on", [0, 0, 0], [0, 0, 1], [0, 0, 5], 1.0, 0.001, 1000.0)
    add("offset", [0.25, -0.5, 0], norm([0.1, 0.2, 1]), [0.5, 0.25, 6], 1.5, 0.001, 1000.0)
    add("origin_inside_misses", [0, 0, 5], [0, 0, 1], [0, 0, 5], 1.0, 0.001, 1000.0)  # near root negative
    add("behind", [0, 0, 0], [0, 0, -1], [0, 0, 5], 1.0, 0.001, 1000.0)
    add("beyond_tmax", [0, 0, 0], [0, 0, 1], [0, 0, 50], 1.0, 0.001, 10.0)
    add("before_tmin", [0, 0, 0], [0, 0, 1], [0, 0, 1.0005], 1.0, 0.001, 1000.0)
    add("miss_sideways", [0, 0, 0], [1, 0, 0], [0, 0, 5], 1.0, 0.001, 1000.0)
    add("unnormalised_dir", [0, 0, 0], [0, 0, 2], [0, 0, 5], 1.0, 0.001, 1000.0)
    add("ground_sphere", [7.5, 1.0, -7.5], norm([-2.0, -1.5, 2.0]), [0, -1001, 0], 1000.0, 0.001, 1000.0)
    return cases


def reflect(i, n):
    return i - n * float(n @ i) * 2.0


def refract(i, n, eta):
    d = float(n @ i)
    k = 1.0 - eta * eta * (1.0 - d * d)
    if k >= 0.0:
        return eta * i - (eta * d + math.sqrt(k)) * n
    return np.zeros(3)


def scatter_cases():
    # Material::Scatter, ray_tracing/default_materials.cpp:11-67
    cases = []

    def add(name, mtype, albedo, rough, refr, rd, n, ball, xi):
        rd, n, ball = norm(rd), norm(n), np.asarray(ball, dtype=np.float64)
        if mtype == 0:
            out, e = norm(n + ball), 0.7
        elif mtype == 1:
            r = norm(reflect(rd, n) + ball * rough)
            out, e = r, (0.0 if float(n @ r) <= 0.0 else 0.9 * (1.0 - rough))
        else:
            refl = reflect(rd, n)
            vdn = float(rd @ n)
            if vdn > 0.0:
                outward, eta = -n, refr
                vdn *= eta
            else:
                outward, eta = n, 1.0 / refr
                vdn = -vdn
            rf = refract(rd, outward, eta)
            ref = 1.0
            if float(rf @ rf) > 1e-5:
                f0 = (1.0 - refr) / (1.0 + refr)
                ref = f0 * f0 + (1.0 - f0 * f0) * (1.0 - vdn) ** 5
            out, e = (refl if xi < ref else rf), 1.0
        cases.append(dict(name=name, type=mtype, albedo=albedo, roughness=rough, refraction=refr,
                          ray_dir=list(map(float, rd)), normal=list(map(float, n)), ball=list(map(float, ball)), xi=xi,
                          out_dir=list(map(float, out)), emissivity=float(e)))

    add("matte_no_jitter", 0, [0.2, 0.4, 0.6], 0.7, 0.0, [0.3, -1, 0.2], [0, 1, 0], [0, 0, 0], 0.5)
    add("matte_jitter", 0, [0.2, 0.4, 0.6], 0.7, 0.0, [0.3, -1, 0.2], [0, 1, 0], [0.3, -0.2, 0.5], 0.5)
    add("metal_mirror", 1, [0.9, 0.8, 0.7], 0.0, 0.1, [1, -1, 0], [0, 1, 0], [0.5, 0.5, 0.5], 0.5)
    add("metal_rough", 1, [0.9, 0.8, 0.7], 0.15, 0.1, [1, -1, 0.5], [0, 1, 0], [0.2, -0.6, 0.1], 0.5)
    add("metal_absorbed", 1, [0.9, 0.8, 0.7], 0.2, 0.1, [1, -0.01, 0], [0, 1, 0], [0, -1, 0], 0.5)
    add("glass_enter_refract", 2, [0.5, 0.9, 0.9], 0.0, 0.85, [0.2, -1, 0.1], [0, 1, 0], [0, 0, 0], 0.5)
    add("glass_enter_reflect", 2, [0.5, 0.9, 0.9], 0.0, 0.85, [0.2, -1, 0.1], [0, 1, 0], [0, 0, 0], 0.0)
    add("glass_exit", 2, [0.5, 0.9, 0.9], 0.0, 0.85, [0.2, 1, 0.1], [0, 1, 0], [0, 0, 0], 0.5)
    add("glass_total_internal_reflection", 2, [0.5, 0.9, 0.9], 0.0, 0.85, [1, -0.05, 0], [0, 1, 0], [0, 0, 0], 0.99)
    return cases


def clamp(x, lo, hi):
    return min(max(x, lo), hi)


def specular_cases():
    # DirectionalLight::GetSpecular, ray_tracing/directional_light.cpp:37-60
    cases = []

    def add(name, ldir, lcol, albedo, rough, refr, rd, n):
        ldir, rd, n = norm(ldir), norm(rd), norm(n)
        vdn = clamp(float(-rd @ n), 0.0, 1.0)
        ndl = clamp(float(n @ -ldir), 0.0, 1.0)
        h = -norm(ldir + rd)
        ndh = clamp(float(n @ h), 1e-5, 1.0)
        vdh = clamp(float(-rd @ h), 0.0, 1.0)
        g = min(1.0, 2.0 * ndh * min(vdn, ndl) / vdh) if abs(vdh) >= 1e-5 else 1.0
        r = max(rough, 0.03)
        r2 = r * r
        d = math.exp((ndh * ndh - 1.0) / (r2 * ndh * ndh)) / (4.0 * r2 * ndh ** 4)
        f0 = (1.0 - refr) / (1.0 + refr)
        f = f0 * f0 + (1.0 - f0 * f0) * (1.0 - vdn) ** 5
        spec = clamp(g * d * f, 0.0, 1.0)
        col = (np.asarray(lcol) * 0.5 + np.asarray(albedo) * 0.5) * spec
        cases.append(dict(name=name, light_dir=list(map(float, ldir)), light_color=lcol, albedo=albedo, roughness=rough,
                          refraction=refr, ray_dir=list(map(float, rd)), normal=list(map(float, n)),
                          color=list(map(float, col)), spec=float(spec)))

    L = [-0.4, -1.0, 0.6]
    add("matte_defaults", L, [1, 1, 1], [0.65, 0.65, 0.65], 0.7, 0.0, [0.3, -0.8, 0.4], [0, 1, 0])
    add("metal_rough", L, [1, 1, 1], [0.9, 0.5, 0.2], 0.2, 0.1, [0.5, -0.6, -0.4], [0.1, 1, -0.1])
    add("glass_floor_roughness", L, [1, 1, 1], [0.3, 0.9, 0.6], 0.0, 0.85, [0.39, -1.0, -0.61], [0, 1, 0])
    add("grazing", L, [1, 0.9, 0.8], [0.2, 0.2, 0.9], 0.1, 0.15, [1, -0.02, 0], [0, 1, 0])
    add("facing_away", L, [1, 1, 1], [0.5, 0.5, 0.5], 0.3, 0.0, [0, 1, 0], [0, 1, 0])
    return cases


def accumulate_cases():
    # demos/demo_frame_cpu.cpp:60-75
    cases = []

    def add(name, l, e):
        att = np.ones(3)
        rest = 1.0
        i = len(l) - 1
        while i >= 0:
            att = att * (np.asarray(l[i], dtype=np.float64) * rest)
            rest *= e[i]
            i -= 1
            if abs(rest) < 1e-5:
                break
        cases.append(dict(name=name, l=l, e=e, color=list(map(float, 0.5 * att))))

    add("single", [[0.5, 0.6, 0.7]], [0.7])
    add("two", [[0.5, 0.6, 0.7], [1.2, 0.9, 0.3]], [0.7, 0.9])
    add("closed_form_depth4", [[0.9, 0.8, 0.7], [1.1, 1.0, 0.9], [0.6, 0.7, 0.8], [1.3, 0.2, 0.5]], [0.7, 0.81, 1.0, 0.72])
    add("absorbed_terminal", [[0.9, 0.8, 0.7], [0.9, 0.9, 0.9]], [0.85, 1.0])
    add("break_on_small_energy", [[0.9, 0.8, 0.7], [1.1, 1.0, 0.9], [0.6, 0.7, 0.8]], [0.5, 1e-3, 1e-3])
    return cases


def primary_cases():
    # Frame::ForEachRayImpl, ray_tracing/frame.cpp:31-62
    cases = []

    def add(name, pos, direction, w, h, s, x, y, sx, sy):
        d = norm(direction)
        fov, znear = 2.0 * math.pi / 3.0, 0.001
        right = np.cross([0.0, 1.0, 0.0], d)
        up = np.cross(d, right)
        dw = znear / math.tan(0.5 * fov)
        dh = dw * (h / w)
        cell_x, cell_y = 2.0 * dw / w, 2.0 * dh / h
        dy = (-2.0 * y / h + 1.0) * dh
        dx = (2.0 * x / w - 1.0) * dw
        sdx = dx + cell_x * sx / s
        sdy = dy - cell_y * sy / s
        v = norm(d * znear + up * sdy + right * sdx)
        cases.append(dict(name=name, position=pos, direction=direction, width=w, height=h, samples=s, x=x, y=y, sx=sx,
                          sy=sy, dir=list(map(float, v))))

    for demo, pos, direction in (("pretty", [7.5, 1.0, -7.5], [-2.0, -0.5, 2.0]), ("glass", [0.0, 2.0, -10.0], [0.0, -1.0, 3.0])):
        add(demo + "_top_left", pos, direction, 1024, 768, 2, 0, 0, 0, 0)
        add(demo + "_bottom_right_last_sample", pos, direction, 1024, 768, 2, 1023, 767, 1, 1)
        add(demo + "_centre", pos, direction, 1920, 1080, 16, 960, 540, 7, 9)
    return cases


def philox_cases():
    # Random123 kat_vectors, philox4x32 10 rounds
    return [
        dict(ctr=[0, 0, 0, 0], key=[0, 0], out=[0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        dict(ctr=[0xFFFFFFFF] * 4, key=[0xFFFFFFFF] * 2, out=[0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        dict(ctr=[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], key=[0xA4093822, 0x299F31D0],
             out=[0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]


def main():
    analytic = dict(sphere=sphere_cases(), scatter=scatter_cases(), specular=specular_cases(),
                    accumulate=accumulate_cases(), primary=primary_cases(), philox=philox_cases())
    with open(os.path.join(HERE, "analytic.json"), "w") as f:
        json.dump(analytic, f, indent=1)

    from oracle import oracle_ffi
    from rt_amd import ffi

    for demo, name, spheres in ((1, "pretty", 0), (2, "glass", 0), (3, "synthetic300", 300)):
        scene = ffi.Scene.demo(demo, 1, spheres)
        for mode, depth in (("d1", 1), ("d2", 0)):
            p = ffi.params(64, 48, 2, flags=ffi.FLAG_DETERMINISTIC, seed=1, max_depth=depth)
            rgb, _ = oracle_ffi.render(scene, p, 4)
            np.save(os.path.join(HERE, "frame_%s_%s.npy" % (name, mode)), rgb)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
