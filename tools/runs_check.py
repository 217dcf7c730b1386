"""Slices by runs of pixels (RT_HIP_FLAG_PIXEL_RUNS) against the whole frame, bit for bit, over a handful of frame
shapes, scenes and slice counts (the same property tests/test_gpu_pixel_runs.py checks, with the 1080p frame).

python tools/runs_check.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rt_amd import ffi, sharding
r = ffi.HipRenderer(0)
def planes_of(scene, p, w, h):
    planes = torch.empty(3 * w * h, dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    r.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return planes.cpu().numpy(), stats.cpu().numpy()
ok = True
for demo, spheres, w, h, s, flags in [(1, 0, 161, 97, 16, 0), (2, 0, 200, 120, 10, 0), (1, 0, 333, 77, 3, 0), (1, 0, 128, 64, 32, 0),
                                      (3, 300, 96, 64, 4, 0), (3, 3000, 64, 48, 2, ffi.FLAG_GRID), (1, 0, 40, 30, 2, 0), (1, 0, 1920, 1080, 16, 0)]:
    scene = ffi.Scene.demo(demo, 1, spheres or 10000)
    r.upload(scene)
    whole, st = planes_of(scene, ffi.params(w, h, s, seed=5, flags=flags), w, h)
    for g in (2, 3, 8):
        total = np.zeros_like(whole)
        samples = 0
        for rank in range(g):
            p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=5, flags=flags), rank, g, pixel_runs=True)
            pl, sst = planes_of(scene, p, w, h)
            # foreign pixels must be exactly zero
            owner = (np.arange(w * h) // 64) % g
            assert np.all(pl.reshape(3, -1)[:, owner != rank] == 0.0), "foreign pixels not zero"
            total += pl
            samples += int(sst[0])
        same = np.array_equal(total.view(np.uint32), whole.view(np.uint32))
        print("demo %d n=%d %dx%d s=%d flags=%d G=%d: bit-equal %s, samples %d of %d" % (demo, scene.sphere_count, w, h, s, flags, g, same, samples, int(st[0])), flush=True)
        ok &= same and samples == int(st[0])
print("ALL OK" if ok else "MISMATCH")
