#!/usr/bin/env python3
"""Headline benchmark: pretty_spheres, 1920x1080, 16x16 supersampling, traced on N MI355X GPUs.

A step is one whole frame with the scene already resident in HBM: every rank traces its slice of the
sample space (rt_hip_trace_device), the planar sums are reduced to rank 0 over RCCL, and rank 0 runs the
resolve pass (rt_hip_resolve_device).  The frame is fixed, so more GPUs means strong scaling.  Two frames
are in flight, on two streams with their own planes: while the planes of frame k are being reduced, every rank
already traces frame k + 1, whose first waves also take over the compute units that the last waves of frame k
leave idle (--sync-reduce runs one frame at a time instead); all K frames are reduced and resolved inside
the timed region, so the value is the throughput of a stream of frames, not the inverse of one frame's latency.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `roofline` prices the trace kernel's algorithmic FLOPs (20 per ray-sphere test,
30 per hit, 250 per shaded bounce; counted on the device) against the FP32 vector peak, with the kernel's
duration measured by HIP events on its launch stream; `cpu_baseline` times the CPU restatement of the
reference's multithreaded path on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rt_amd import ffi, sharding  # noqa: E402

PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X: 256 CUs x 4 SIMDs x 32 lanes x 2 FLOP x 2.4 GHz


def algorithmic_flop(stats, n_spheres):
    return 20.0 * n_spheres * stats["ray_queries"] + 30.0 * stats["hit_queries"] + 250.0 * stats["shaded_bounces"]


def profiled_traffic(workload_key):
    """HBM bytes per trace launch from the committed rocprofv3 counter passes (FETCH_SIZE doubled as the gfx950
    guide prescribes for wide reads, plus WRITE_SIZE), or None when this workload was not profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload_key, {}).get("trace_kernel_bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, width, height, samples_in_row):
    """The oracle (a port of the reference's CPU path) on a bounded sample of the workload: all the threads
    this process may use (at most 16), and one thread on a sixteenth of that sample."""
    from oracle import oracle_ffi  # checker; used here as the reported CPU baseline only

    threads = min(os.cpu_count() or 1, 16)
    p = ffi.params(width, height, samples_in_row, seed=1)
    t0 = time.perf_counter()
    _, st = oracle_ffi.render(scene, p, threads)
    dt = time.perf_counter() - t0
    p1 = ffi.params(max(width // 4, 16), max(height // 4, 16), samples_in_row, seed=1)
    t0 = time.perf_counter()
    _, st1 = oracle_ffi.render(scene, p1, 1)
    dt1 = time.perf_counter() - t0
    return {
        "value": round(st.samples / dt / 1e6, 3),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": "same scene and camera at %dx%d, %dx%d samples per pixel (%.1f Msamples, %.1f s)"
                  % (width, height, samples_in_row, samples_in_row, st.samples / 1e6, dt),
        "value_1_thread": round(st1.samples / dt1 / 1e6, 3),
        "sample_1_thread": "%dx%d, %dx%d samples per pixel (%.1f Msamples, %.1f s)"
                           % (p1.width, p1.height, samples_in_row, samples_in_row, st1.samples / 1e6, dt1),
        "cpu_model": cpu_model(),
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--samples", type=int, default=16, help="samples per row: S x S per pixel")
    ap.add_argument("--demo", type=int, default=1, help="1 pretty_spheres, 2 glass_spheres, 3 synthetic field")
    ap.add_argument("--spheres", type=int, default=10000, help="sphere count of demo 3")
    ap.add_argument("--grid", action="store_true", help="rays walk the scene's uniform grid (scenes of 64+ spheres)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-reduce", action="store_true",
                    help="one frame at a time on one stream (default: two frames in flight on two streams)")
    ap.add_argument("--rehearse", action="store_true",
                    help="rehearsal of the multi-rank path on a one-GPU box: every rank uses cuda:0 and the "
                         "collectives run over gloo on host copies (timings are not meaningful)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world:
        if rank == 0 and distributed:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
        if not distributed and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))

    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist  # nccl == RCCL on ROCm

        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    renderer = ffi.HipRenderer(local_rank)  # raises if librt_hip.so or the GPU is missing: no fallback
    scene = ffi.Scene.demo(args.demo, 1, args.spheres)
    w, h, s = args.width, args.height, args.samples
    # Two frames in flight pay off where the end of a launch is short against the launch (the small scenes of the
    # reference's demos, or grid walks).  With hundreds of spheres tested per ray the last samples of a frame take
    # long, the next frame's waves (on the higher-priority stream) crowd them out, and the frames end up slower
    # than one at a time (measured: 251 spheres 43.6 ms alone, 50.6 ms overlapped): those run one at a time.
    if scene.sphere_count > 64 and not args.grid:
        args.sync_reduce = True
    flags = (ffi.FLAG_GRID if args.grid else 0) | (0 if args.sync_reduce else ffi.FLAG_PIPELINED)
    p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1, flags=flags), rank, world)
    full = ffi.params(w, h, s, seed=1, flags=flags)

    device = torch.device("cuda", local_rank)
    # Two frames in flight, each with its own stream, sum planes, image and counters: frame k + 1 is traced while
    # the planes of frame k are still being reduced, and its first waves fill the compute units that the last
    # waves of frame k leave idle (rt_hip_trace_device keeps a workspace per stream).
    lanes = 1 if args.sync_reduce else 2
    # (different priorities: HIP then gives the streams hardware queues of their own, without which the second
    # frame's kernels would simply queue behind the first's)
    try:
        streams = [torch.cuda.Stream(device=device, priority=-(i % 2)) for i in range(lanes)]
    except (RuntimeError, ValueError):  # no stream priorities here: still correct, the frames just take turns
        streams = [torch.cuda.Stream(device=device) for _ in range(lanes)]
    planes = [torch.empty(3 * w * h, dtype=torch.float32, device=device) for _ in range(2)]
    rgbs = [torch.empty(w * h * 3, dtype=torch.float32, device=device) for _ in range(2)] if rank == 0 else None
    d_stats = [torch.zeros(16, dtype=torch.int64, device=device) for _ in range(2)]
    renderer.upload(scene)
    rehearsing = args.rehearse and distributed
    host = [torch.empty(3 * w * h, dtype=torch.float32) for _ in range(2)] if rehearsing else None

    def on_stream(k):
        return torch.cuda.stream(streams[k % lanes])

    def resolve(k, summed):
        if rehearsing:
            planes[k % 2].copy_(summed)
        stream = torch.cuda.current_stream().cuda_stream
        renderer.resolve_device(planes[k % 2].data_ptr(), rgbs[k % 2].data_ptr(), full, stream)

    pipe = sharding.ReducePipeline(dist, world, host if rehearsing else planes, on_done=resolve, rank=rank,
                                   context=on_stream)
    frame = [0]

    def step():
        k = frame[0]
        frame[0] += 1
        pipe.acquire(k)
        with on_stream(k):
            stream = torch.cuda.current_stream().cuda_stream
            renderer.trace_device(scene, p, planes[k % 2].data_ptr(), d_stats[k % 2].data_ptr(), stream)
            if rehearsing:
                host[k % 2].copy_(planes[k % 2])
            pipe.submit(k)
        if args.sync_reduce:
            pipe.drain()

    def fence():
        pipe.drain()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The same frame one at a time (no overlap of frames, no flag that announces one): what a caller that needs
    # each image before it asks for the next would see.  Reported beside the headline, never as it.
    p_alone = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1, flags=flags & ~ffi.FLAG_PIPELINED), rank, world)
    singles = max(2, min(args.steps, 5))
    fence()
    t1 = time.perf_counter()
    for _ in range(singles):
        stream = torch.cuda.current_stream().cuda_stream
        renderer.trace_device(scene, p_alone, planes[0].data_ptr(), d_stats[1].data_ptr(), stream)
        if rehearsing:
            host[0].copy_(planes[0])
            sharding.reduce_to_root(dist, host[0], world)
            planes[0].copy_(host[0])
        else:
            sharding.reduce_to_root(dist, planes[0], world)
        if rank == 0:
            renderer.resolve_device(planes[0].data_ptr(), rgbs[1].data_ptr(), full, stream)
    fence()
    single_ms = (time.perf_counter() - t1) / singles * 1e3
    if distributed:
        t = torch.tensor([single_ms], dtype=torch.float64, device="cpu" if args.rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        single_ms = float(t.item())

    # counters of this rank's last trace, summed over ranks
    counters = d_stats[0][:8].clone()
    if distributed:
        if args.rehearse:
            counters = counters.cpu()
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    names = ["samples", "ray_queries", "hit_queries", "shaded_bounces", "sphere_tests", "exact_tests", "lane_trips",
             "lane_rounds"]
    stats = dict(zip(names, (int(x) for x in counters.tolist())))

    # roofline of the dominant kernel: this rank's trace kernel, HIP events on its launch stream
    # (launched alone, one after the other, so without the flag that tells a launch its successor overlaps it)
    launches = max(2, min(args.steps, 5))
    kernel_ms, kstats = renderer.time_trace_kernel(scene, p_alone, launches)
    k = kstats.as_dict()
    kernel_flop = algorithmic_flop(k, scene.sphere_count)
    achieved = kernel_flop / (kernel_ms * 1e-3) / 1e12

    # the resolve pass alone (HBM-bound: reads 12 B and writes 12 B per pixel), events on its launch stream
    resolve_ms = None
    if rank == 0:
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            renderer.resolve_device(planes[0].data_ptr(), rgbs[0].data_ptr(), full, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        resolve_ms = e0.elapsed_time(e1) / 10.0

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_samples = w * h * s * s
        mean_radiance = float(rgbs[0].mean().item())
        out = {
            "metric": "Msamples/s (camera samples, each traced to its end: primary, scattered and shadow rays)",
            "value": round(total_samples / (elapsed / args.steps) / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s %dx%d, %dx%d SSAA, %d spheres, samples sharded over %d GPU(s), RCCL sum-reduce"
                            % ({1: "pretty_spheres", 2: "glass_spheres", 3: "synthetic_spheres"}[args.demo], w, h, s, s,
                               scene.sphere_count, world),
                "scene_seed": 1,
                "render_seed": 1,
                "max_depth": 10,
                "intersection": "uniform grid" if args.grid else "every sphere (LDS-resident or LDS-tiled)",
                "frames_in_flight": lanes,
            },
            "ms_per_frame_one_at_a_time": round(single_ms, 3),
            "mrays_per_s": round(stats["ray_queries"] / (elapsed / args.steps) / 1e6, 1),
            "ray_queries_per_sample": round(stats["ray_queries"] / max(stats["samples"], 1), 3),
            "mean_radiance": round(mean_radiance, 5),
            "roofline": {
                "bound": "valu_fp32" if not args.grid else "latency (grid walk; achieved/frac below still credit "
                                                          "the brute-force test count and do not apply)",
                "bound_note": "FP32 vector-ALU issue; this kernel is neither HBM- nor MFMA-bound "
                              "(0.43 GB of HBM traffic per 3.2 TFLOP launch, no matrix work)",
                "achieved": round(achieved, 2),
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 4),
                "traffic": profiled_traffic("demo%d_%dx%d_s%d_g%d" % (args.demo, w, h, s, world)),
                "kernel": "trace_kernel",
                "kernel_ms": round(kernel_ms, 3),
                "flop_per_launch": kernel_flop,
                "executed_filter_tests": k["sphere_tests"],
                "executed_exact_tests": k["exact_tests"],
                "executed_over_algorithmic_tests": round(
                    k["sphere_tests"] / max(k["ray_queries"] * scene.sphere_count, 1), 4),
                "lane_utilisation": round(k["shaded_bounces"] / max(k["lane_trips"], 1), 4),
            },
        }
        out["resolve_pass"] = {
            "bound": "hbm",
            "bytes": 24 * w * h,
            "ms": round(resolve_ms, 4),
            "achieved": round(24.0 * w * h / (resolve_ms * 1e-3) / 1e9, 1),
            "peak": 8000.0,
            "unit": "GB/s",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, max(w // 2, 16), max(h // 2, 16), s)
        print(json.dumps(out), flush=True)

    renderer.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
