"""Quick device check: parity of the HIP frame against the oracle on small frames, then timing of larger ones.

Run on a GPU box from the repository root:  python tools/gpu_check.py [--big]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle_ffi  # noqa: E402  (checker)
from rt_amd import ffi  # noqa: E402


def compare(name, ref, out):
    d = np.abs(ref - out)
    rel = d / np.maximum(np.abs(ref), 1e-3)
    bad = (rel.max(axis=2) > 1e-4).mean()
    print("%-34s max_abs %.3g max_rel %.3g frac_pixels>1e-4 %.3g mean_abs %.3g" % (name, d.max(), rel.max(), bad, d.mean()),
          flush=True)


def main():
    r = ffi.HipRenderer(0)
    print(r.device_info(), flush=True)
    for demo in (1, 2):
        sc = ffi.Scene.demo(demo, 1)
        for label, flags, depth, s in (("D1", 1, 1, 2), ("D2", 1, 0, 2), ("stochastic", 0, 0, 2), ("stochastic S3", 0, 0, 3)):
            p = ffi.params(128, 96, s, flags=flags, seed=7, max_depth=depth)
            ref, rst = oracle_ffi.render(sc, p)
            out, st = r.render(sc, p)
            compare("demo %d %s" % (demo, label), ref, out)
            a, b = rst.as_dict(), st.as_dict()
            same = all(a[k] == b[k] for k in ("samples", "ray_queries", "hit_queries", "shaded_bounces"))
            print("   counters equal: %s  oracle %s  hip %s" % (same, a, b), flush=True)
    if "--big" in sys.argv:
        for demo, w, h, s in ((1, 1024, 768, 2), (2, 1024, 768, 10), (1, 1920, 1080, 4), (1, 1920, 1080, 16)):
            sc = ffi.Scene.demo(demo, 1)
            p = ffi.params(w, h, s, seed=1)
            for _ in range(2):
                t0 = time.time()
                out, st = r.render(sc, p)
                dt = time.time() - t0
            d = st.as_dict()
            flop = 20.0 * sc.sphere_count * d["ray_queries"] + 30.0 * d["hit_queries"] + 250.0 * d["shaded_bounces"]
            print("demo %d %dx%d S=%d: wall %.1f ms kernel %.2f ms  %.1f Msamples/s  %.1f Mrays/s  %.2f TFLOP/s (alg)  q/sample %.2f  exec tests/alg tests %.3f  mean %.4f"
                  % (demo, w, h, s, dt * 1e3, d["kernel_ms"], d["samples"] / d["kernel_ms"] / 1e3,
                     d["ray_queries"] / d["kernel_ms"] / 1e3, flop / d["kernel_ms"] / 1e9,
                     d["ray_queries"] / d["samples"], d["sphere_tests"] / (d["ray_queries"] * sc.sphere_count), out.mean()),
                  flush=True)
    r.close()


if __name__ == "__main__":
    main()
