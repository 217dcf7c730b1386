"""What one rank of G does per frame: the time of rt_hip_trace_device (trace + accumulate pass) on slice 0 of G
of the headline frame, against 1/G of the whole frame.  Leaves out the RCCL reduce, which needs the other GPUs.

python tools/slice_timing.py [width height samples_in_row [slice counts [demo [spheres]]]]
Environment: RT_SLICE_BY=samples for slices by samples (default: by runs of 64 pixels, as bench.py shards),
RT_LANES=n for n frames in flight in the second measurement (default 2), RT_LEAVE_ROOM=1 for launches that leave 64
block slots free (what bench.py asks for on several GPUs, so that the reduce finds compute units), RT_AMD_LIB_DIR=dir for another build of
the libraries.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rt_amd import ffi, sharding  # noqa: E402

r = ffi.HipRenderer(0)
demo = int(sys.argv[5]) if len(sys.argv) >= 6 else 1
scene = ffi.Scene.demo(demo, 1, int(sys.argv[6]) if len(sys.argv) >= 7 else 10000)
w, h, s = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 16)
planes = [torch.empty(3 * w * h, dtype=torch.float32, device="cuda") for _ in range(4)]
stats = [torch.zeros(16, dtype=torch.int64, device="cuda") for _ in range(4)]
# one stream, then two streams taking turns (different priorities, so that HIP gives them hardware queues of
# their own): the second number is the period of a stream of frames whose ends and starts overlap
LANES = int(os.environ.get("RT_LANES", "2"))  # frames in flight in the second measurement
print("stream priority range", torch.cuda.Stream.priority_range())
streams = [torch.cuda.Stream(priority=-(i % 3)) for i in range(max(LANES, 2))]
RUNS = os.environ.get("RT_SLICE_BY", "runs") != "samples"  # slices by runs of pixels (bench.py's default) or by samples
print("slices by %s" % ("runs of 64 pixels" if RUNS else "samples"))
base = None
for g in ([int(v) for v in sys.argv[4].split(",")] if len(sys.argv) >= 5 else (1, 2, 4, 8)):
    ms = []
    for lanes in (1, LANES):
        flags = (ffi.FLAG_PIPELINED if lanes >= 2 else 0) | (ffi.FLAG_LEAVE_ROOM if os.environ.get("RT_LEAVE_ROOM") else 0)
        p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1, flags=flags), 0, g, pixel_runs=RUNS)
        for k in range(4 if demo != 3 else 1):
            r.trace_device(scene, p, planes[k % lanes].data_ptr(), stats[k % lanes].data_ptr(), streams[k % lanes].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 12 if demo != 3 else 2
        for k in range(n):
            r.trace_device(scene, p, planes[k % lanes].data_ptr(), stats[k % lanes].data_ptr(), streams[k % lanes].cuda_stream)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / n * 1e3)
    base = base or ms
    print("G = %d: %.3f ms per frame slice (%.2fx of the whole frame's time; ideal %.3f ms); %.3f ms with two frames "
          "in flight (%.2fx)" % (g, ms[0], base[0] / ms[0], base[0] / g, ms[1], base[1] / ms[1]))
r.close()
