#!/usr/bin/env python3
"""The five configurations of the baseline on this machine: the HIP backend at the configuration's own size and
the CPU restatement of the reference (the oracle; -O2 and -O3 builds, all host threads and one thread) on a bounded
sample of it.  Prints a markdown table and writes gpurun_out/baseline_table.json.

python tools/baseline_table.py [--cpu-seconds 6]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi  # noqa: E402  (the checker, timed here as the reported CPU baseline)
from rt_amd import ffi  # noqa: E402

CONFIGS = [
    ("cfg1 pretty_spheres 1024x768 2x2", 1, 0, 1024, 768, 2),
    ("cfg2 glass_spheres 1024x768 10x10", 2, 0, 1024, 768, 10),
    ("cfg3 pretty_spheres 1920x1080 16x16", 1, 0, 1920, 1080, 16),
    ("cfg4 glass_spheres 3840x2160 32x32", 2, 0, 3840, 2160, 32),
    ("cfg5 synthetic 10k spheres 1920x1080 8x8", 3, 10000, 1920, 1080, 8),
]


def cpu_rate(scene, w, h, s, threads, build, seconds):
    """Msamples/s of the oracle on a frame scaled down (same camera, same samples per pixel) to about `seconds`."""
    fw, fh = 16, 9  # a probe small enough for one thread against 10 000 spheres
    print("  cpu %s, %d thread(s) ..." % (build or "o2", threads), flush=True)
    t0 = time.perf_counter()
    _, st = oracle_ffi.render(scene, ffi.params(fw, fh, s, seed=1), threads, build=build)
    rate = st.samples / (time.perf_counter() - t0)
    scale = min(max((rate * seconds / (w * h * s * s)) ** 0.5, 1.0 / 120), 1.0)
    fw, fh = max(int(w * scale), 16), max(int(h * scale), 16)
    t0 = time.perf_counter()
    _, st = oracle_ffi.render(scene, ffi.params(fw, fh, s, seed=1), threads, build=build)
    dt = time.perf_counter() - t0
    return st.samples / dt / 1e6, "%dx%d, %.1f s" % (fw, fh, dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    args = ap.parse_args()
    threads = min(os.cpu_count() or 1, 16)
    r = ffi.HipRenderer(0)
    rows = []
    for name, demo, spheres, w, h, s in CONFIGS:
        scene = ffi.Scene.demo(demo, 1, spheres or 10000)
        p = ffi.params(w, h, s, seed=1)
        r.render(scene, p)
        t0 = time.perf_counter()
        reps = 3 if w * h * s * s < 2e9 else 1
        for _ in range(reps):
            _, st = r.render(scene, p)
        frame_ms = (time.perf_counter() - t0) / reps * 1e3
        kernel_ms, ks = r.time_trace_kernel(scene, p, reps)
        flop = 20.0 * scene.sphere_count * ks.ray_queries + 30.0 * ks.hit_queries + 250.0 * ks.shaded_bounces
        row = {
            "config": name, "samples": int(st.samples), "gpu_kernel_ms": round(kernel_ms, 3),
            "gpu_frame_ms_host_to_host": round(frame_ms, 3),
            "gpu_msamples_per_s": round(st.samples / frame_ms / 1e3, 1),
            "gpu_mrays_per_s": round(st.ray_queries / frame_ms / 1e3, 1),
            "roofline_frac_algorithmic": round(flop / (kernel_ms * 1e-3) / 157.3e12, 3),
        }
        for build, key in ((None, "o2"), ("o3", "o3")):
            rate, what = cpu_rate(scene, w, h, s, threads, build, args.cpu_seconds)
            row["cpu_%s_msamples_per_s" % key] = round(rate, 3)
            row["cpu_%s_sample" % key] = what
        rate1, what1 = cpu_rate(scene, w, h, s, 1, None, args.cpu_seconds)
        row["cpu_o2_1_thread_sample"] = what1
        row["cpu_o2_1_thread_msamples_per_s"] = round(rate1, 3)
        row["cpu_threads"] = threads
        rows.append(row)
        print(json.dumps(row), flush=True)
    r.close()
    with open("/proc/cpuinfo") as f:
        model = [l.split(":", 1)[1].strip() for l in f if l.lower().startswith("model name")]
    out = {"cpu_model": model[0] if model else "unknown", "host_cpus": os.cpu_count(), "threads_used": threads, "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "baseline_table.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("| configuration | GPU frame ms (host to host) | GPU kernel ms | GPU Msamples/s | GPU Mrays/s | algorithmic frac | CPU -O2 Msamples/s (%d thr) | CPU -O3 Msamples/s | CPU -O2 1 thread |" % threads)
    print("|---|---|---|---|---|---|---|---|---|")
    for row in rows:
        print("| %s | %.2f | %.2f | %.0f | %.0f | %.3f | %.2f (%s) | %.2f | %.3f |" % (
            row["config"], row["gpu_frame_ms_host_to_host"], row["gpu_kernel_ms"], row["gpu_msamples_per_s"],
            row["gpu_mrays_per_s"], row["roofline_frac_algorithmic"], row["cpu_o2_msamples_per_s"], row["cpu_o2_sample"],
            row["cpu_o3_msamples_per_s"], row["cpu_o2_1_thread_msamples_per_s"]))


if __name__ == "__main__":
    main()
