"""Two builds of the library side by side: kernel time and a stream of frames for the headline frame, a 1/8 slice of
it, glass 1024x768 10x10 and 32x32, each build in a process of its own, taking turns.

python tools/ab_kernel.py <lib dir A> <lib dir B> [rounds]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from rt_amd import ffi, sharding
r = ffi.HipRenderer(0)
streams = [torch.cuda.Stream(priority=-(i %% 2)) for i in range(2)]
for name, demo, w, h, s, sl in (("pretty 1080p 16x16", 1, 1920, 1080, 16, None), ("1/8 of it by runs", 1, 1920, 1080, 16, (0, 8)),
                                ("glass 1024x768 10x10", 2, 1024, 768, 10, None), ("glass 1024x768 32x32", 2, 1024, 768, 32, None)):
    scene = ffi.Scene.demo(demo, 1)
    mk = lambda: ffi.params(w, h, s, seed=1)
    p = sharding.slice_params(mk, sl[0], sl[1], pixel_runs=True) if sl else mk()
    planes = [torch.empty(3 * w * h, dtype=torch.float32, device="cuda") for _ in range(2)]
    kernel_ms, _ = r.time_trace_kernel(scene, p, 5)
    q = type(p).from_buffer_copy(p)
    q.flags |= ffi.FLAG_PIPELINED  # one of a stream of frames
    for k in range(4):
        r.trace_device(scene, q, planes[k %% 2].data_ptr(), None, streams[k %% 2].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(16):
        r.trace_device(scene, q, planes[k %% 2].data_ptr(), None, streams[k %% 2].cuda_stream)
    torch.cuda.synchronize()
    print("  %%-22s kernel %%.3f ms, stream %%.3f ms" %% (name, kernel_ms, (time.perf_counter() - t0) / 16 * 1e3), flush=True)
r.close()
''' % ROOT

dirs = sys.argv[1:3]
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for d in dirs:
        print("%s (round %d)" % (d, rnd), flush=True)
        env = dict(os.environ, RT_AMD_LIB_DIR=os.path.join(ROOT, d))
        rc = subprocess.run([sys.executable, "-c", CHILD], env=env).returncode
        if rc != 0:
            sys.exit(rc)
