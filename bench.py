#!/usr/bin/env python3
"""Headline benchmark: pretty_spheres, 1920x1080, 16x16 supersampling, traced on N MI355X GPUs.

A step is one whole frame with the scene already resident in HBM: every rank traces its slice of the
frame (rt_hip_trace_device; runs of 64 pixels dealt out to the ranks in turn, --sample-slices for every rank's share
of every pixel's samples), the planar sums are put together on rank 0 over RCCL (a gather of each rank's own runs;
a sum-reduce with --sample-slices), and rank 0 runs the resolve pass (rt_hip_resolve_device).  The frame is fixed, so more GPUs means strong scaling.  Two frames
are in flight, on two streams with their own planes: while the planes of frame k are being reduced, every rank
already traces frame k + 1, whose first waves also take over the compute units that the last waves of frame k
leave idle (--sync-reduce runs one frame at a time instead); all K frames are reduced and resolved inside
the timed region, so the value is the throughput of a stream of frames, not the inverse of one frame's latency.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `value` / `ms_per_step` are the throughput of that stream of frames (two in flight),
images left on the device; beside them the line carries the latency of a frame traced alone
(`ms_per_frame_one_at_a_time`), the same with the download of the image into pinned host memory
(`ms_per_frame_one_at_a_time_with_d2h`) and, on one GPU, the period of a stream of frames through the job API
with every image downloaded while the next frame is traced (`ms_per_frame_with_d2h`).

`roofline` prices what the trace kernel executed: the vector instructions it issued (SQ_INSTS_VALU x 64 lanes from
the committed counter profile of the workload, or modelled from this run's device counters when there is none) over
the kernel's duration (HIP events on its launch stream) against one instruction per lane and clock -- a fraction of
the hardware that cannot exceed 1.  Beside it: `fp32_flop_frac`, the FP32 arithmetic among those instructions
(SQ_INSTS_VALU_{FMA,ADD,MUL}_F32, an FMA two operations) against the FP32 vector peak; `lane_utilisation`, the share
of the lanes of those instructions that were switched on (SQ_THREAD_CYCLES_VALU); and `algorithmic_frac`, the
algorithm's own credit (20 FLOP per ray-sphere test the reference would run, 30 per hit, 250 per shaded bounce; counted
on the device) against the same peak -- a measure of how fast the algorithm's work gets done, which passes 1 because
the filter spends far fewer than 20 FLOP per test.  `cpu_baseline` times the CPU restatement of the
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
PEAK_VALU_LANE_OPS = 78.64e12    # ... of which one vector instruction per lane and clock: 256 x 4 x 32 x 2.4 GHz

# Vector instructions per unit of work, counted in the kernel's ISA and fitted to SQ_INSTS_VALU of the profiled
# workloads (the model lands within 2.5 % of the counter on all five): what the executed-work model multiplies the
# device counters with for workloads that have no committed counter profile.
INSTR_SHADE_SPHERE = 13.5  # one sphere in a shade pass: squared bound, ray projection, the two tests (12) + loop
INSTR_PRIMARY_TEST = 4.5   # one camera-table entry against one direction (3 FMA + the mask shift) + loop
INSTR_CONE_SPHERE = 40     # one sphere the cone touched: its record, offset and c, loop bookkeeping
INSTR_EXACT_TEST = 25      # one exact intersection (b, discriminant, short correctly rounded sqrt and quotient, range tests)
INSTR_SHADED_BOUNCE = 678  # scatter, specular, two Philox blocks, six ray set-ups, light sum, path bookkeeping
INSTR_SAMPLE = 470         # a primary round: primary direction, item bookkeeping, hit set-up, share of column preparation


def algorithmic_flop(stats, n_spheres):
    return 20.0 * n_spheres * stats["ray_queries"] + 30.0 * stats["hit_queries"] + 250.0 * stats["shaded_bounces"]


def executed_lane_ops(stats, n_spheres, grid, column_masks):
    """Vector instructions x lanes the trace kernel issued for these counters (see INSTR_*); None for grid walks,
    whose work is not in the counters.  A wave issues for all its lanes, busy or idle, so the passes over the
    scene are counted per loop trip of a lane (lane_trips: one shade pass each; lane_rounds: one primary filter
    pass each where primary rays are filtered per sample)."""
    if grid:
        return None
    shade = stats["lane_trips"] * n_spheres
    if column_masks:
        primary = max(stats["sphere_tests"] - 2 * stats["shaded_bounces"] * n_spheres, 0)  # once per column
    else:
        primary = stats["lane_rounds"] * n_spheres
    return (INSTR_SHADE_SPHERE * shade + INSTR_PRIMARY_TEST * primary + INSTR_CONE_SPHERE * stats["cone_tests"] +
            INSTR_EXACT_TEST * stats["exact_tests"] + INSTR_SHADED_BOUNCE * stats["lane_trips"] +
            INSTR_SAMPLE * stats["lane_rounds"])


def measured_issue_rates():
    """The same fraction for loops of one instruction kind on this part, four waves on every SIMD of the chip, from
    the committed run of tools/micro/valu_rate.hip (profiles/r2_valu_rate_256cu.txt): what `executed.frac` can reach
    at all for code made of such instructions."""
    import re

    path = os.path.join(ROOT, "profiles", "r2_valu_rate_256cu.txt")
    nominal_ns = 2.0 / 2.4  # a wave64 instruction every two cycles at 2.4 GHz
    out = {}
    try:
        with open(path) as f:
            for line in f:
                m = re.match(r"(v_fma_f32 \(3 vgpr\)|v_fmac_f32_e32|v_alignbit_b32|v_rcp_f32)\s{2,}.*?([0-9.]+) ns per instruction", line)
                if m:
                    out[m.group(1).split(" ")[0] + "_loop_frac"] = round(nominal_ns / float(m.group(2)), 3)
    except OSError:
        return None
    if not out:
        return None
    out["source"] = "profiles/r2_valu_rate_256cu.txt"
    return out


def profiled(workload_key):
    """What the committed rocprofv3 counter passes hold for this workload (profiles/traffic.json): HBM bytes per
    trace launch (FETCH_SIZE doubled as the gfx950 guide prescribes for wide reads, plus WRITE_SIZE) and the
    vector instruction count; {} when the workload was not profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload_key, {})
    except (OSError, ValueError):
        return {}


def overlap_profile(workload_key):
    """The trace kernels of successive frames in the default run of this benchmark (two frames in flight), from the
    committed rocprofv3 kernel trace of that command (profiles/overlap.json, written by tools/summarise_overlap.py):
    average duration of a trace kernel, average distance between the starts of successive ones and the time two
    successive ones run side by side."""
    try:
        with open(os.path.join(ROOT, "profiles", "overlap.json")) as f:
            return json.load(f).get(workload_key)
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
    # the same sample with the oracle built -O3 -march=x86-64-v3 (still no FMA contraction: same frames)
    o3 = None
    try:
        t0 = time.perf_counter()
        _, st3 = oracle_ffi.render(scene, p, threads, build="o3")
        o3 = round(st3.samples / (time.perf_counter() - t0) / 1e6, 3)
    except OSError:
        pass
    return {
        "value": round(st.samples / dt / 1e6, 3),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "compiler_flags": "-O2 -ffp-contract=off",
        "sample": "same scene and camera at %dx%d, %dx%d samples per pixel (%.1f Msamples, %.1f s)"
                  % (width, height, samples_in_row, samples_in_row, st.samples / 1e6, dt),
        "value_o3_x86_64_v3": o3,
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
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="frames traced side by side (2 to 4), each on a stream with planes of its own (--sync-reduce: 1). "
                         "One GPU, with the resolve pass behind every trace: 17.3 ms per frame with two, 17.5 with three or "
                         "four; bare 1/8 slices on three streams of one priority ran 1.7 %% faster than on two "
                         "(tools/streams_probe.py) -- for a first run on a node with several GPUs to try")
    ap.add_argument("--equal-priorities", action="store_true",
                    help="the frames' streams all take the default priority instead of two priorities in turn")
    ap.add_argument("--sample-slices", action="store_true",
                    help="several GPUs: divide the frame by samples (every rank visits every column) instead of by "
                         "runs of 64 pixels (a rank traces every sample of its pixels; the reduced frame then equals "
                         "the one-GPU frame bit for bit)")
    ap.add_argument("--no-room", action="store_true",
                    help="several GPUs: do not leave block slots free for the RCCL reduce kernel")
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
    # With a reduce to overlap, the trace leaves 64 of the chip's 1024 block slots free: the RCCL kernel that sums
    # frame k would otherwise find every compute unit taken by the trace of frame k + 1.  One GPU has no reduce
    # and uses every slot.
    if distributed and not args.rehearse and not args.no_room:
        flags |= ffi.FLAG_LEAVE_ROOM
    by_runs = not args.sample_slices
    p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1, flags=flags), rank, world, pixel_runs=by_runs)
    full = ffi.params(w, h, s, seed=1, flags=flags)

    device = torch.device("cuda", local_rank)
    # Two frames in flight, each with its own stream, sum planes, image and counters: frame k + 1 is traced while
    # the planes of frame k are still being reduced, and its first waves fill the compute units that the last
    # waves of frame k leave idle (rt_hip_trace_device keeps a workspace per stream).
    lanes = 1 if args.sync_reduce else max(2, min(args.frames_in_flight, 4))
    sets = max(2, lanes)  # planes, image and counters per frame in flight (a frame's set belongs to its stream)
    # (different priorities: HIP then gives the streams hardware queues of their own for certain)
    try:
        streams = [torch.cuda.Stream(device=device, priority=0 if args.equal_priorities else -(i % 2)) for i in range(lanes)]
    except (RuntimeError, ValueError):  # no stream priorities here: still correct
        streams = [torch.cuda.Stream(device=device) for _ in range(lanes)]
    planes = [torch.empty(3 * w * h, dtype=torch.float32, device=device) for _ in range(sets)]
    rgbs = [torch.empty(w * h * 3, dtype=torch.float32, device=device) for _ in range(sets)] if rank == 0 else None
    d_stats = [torch.zeros(16, dtype=torch.int64, device=device) for _ in range(sets)]
    renderer.upload(scene)
    rehearsing = args.rehearse and distributed
    host = [torch.empty(3 * w * h, dtype=torch.float32) for _ in range(sets)] if rehearsing else None

    def on_stream(k):
        return torch.cuda.stream(streams[k % lanes])

    def resolve(k, summed):
        if rehearsing:
            planes[k % sets].copy_(summed)
        stream = torch.cuda.current_stream().cuda_stream
        renderer.resolve_device(planes[k % sets].data_ptr(), rgbs[k % sets].data_ptr(), full, stream)

    # slices by runs of pixels: every pixel comes from one rank, so the planes are put together by a gather of each
    # rank's own runs (1 / N of the bytes, nothing to add); slices by samples need the sum of the whole planes
    gather = sharding.RunGather(w * h, world, rank, "cpu" if rehearsing else device) if (distributed and by_runs) else None
    gather_note = None
    if gather is not None:
        # one gather before anything is timed: a backend that cannot do it is found out here, and the run goes on
        # with the sum-reduce (the planes of the other ranks' pixels are zero, so the sum is the same frame)
        try:
            probe = (host if rehearsing else planes)[0]
            probe.zero_()
            gather.finish(gather.start(dist, probe), probe)
            if not rehearsing:
                torch.cuda.synchronize()
            ok = torch.ones(1, dtype=torch.int32, device="cpu" if rehearsing else device)
        except Exception as e:  # noqa: BLE001 (whatever the backend raises)
            gather_note = "gather failed (%s: %s), sum-reduce instead" % (type(e).__name__, str(e)[:120])
            ok = torch.zeros(1, dtype=torch.int32, device="cpu" if rehearsing else device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # all ranks take the same path
        if int(ok.item()) == 0:
            gather = None
            gather_note = gather_note or "gather failed on another rank, sum-reduce instead"
    pipe = sharding.ReducePipeline(dist, world, host if rehearsing else planes, on_done=resolve, rank=rank,
                                   context=on_stream, gather=gather)
    frame = [0]

    def step():
        k = frame[0]
        frame[0] += 1
        pipe.acquire(k)
        with on_stream(k):
            stream = torch.cuda.current_stream().cuda_stream
            renderer.trace_device(scene, p, planes[k % sets].data_ptr(), d_stats[k % sets].data_ptr(), stream)
            if rehearsing:
                host[k % sets].copy_(planes[k % sets])
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
    # each image before it asks for the next would see.  Reported beside the headline, never as it.  Measured twice:
    # image left on the device, and downloaded into pinned host memory (rank 0; the whole-job metric with PCIe).
    p_alone = sharding.slice_params(lambda: ffi.params(w, h, s, seed=1, flags=flags & ~ffi.FLAG_PIPELINED), rank, world,
                                    pixel_runs=by_runs)
    singles = max(2, min(args.steps, 5))
    pinned = torch.empty(w * h * 3, dtype=torch.float32, pin_memory=True) if rank == 0 else None

    def frames_alone(download):
        fence()
        t = time.perf_counter()
        for _ in range(singles):
            stream = torch.cuda.current_stream().cuda_stream
            renderer.trace_device(scene, p_alone, planes[0].data_ptr(), d_stats[1].data_ptr(), stream)
            def collect(buf):
                if gather is not None:
                    gather.finish(gather.start(dist, buf), buf)
                else:
                    sharding.reduce_to_root(dist, buf, world)

            if rehearsing:
                host[0].copy_(planes[0])
                collect(host[0])
                planes[0].copy_(host[0])
            else:
                collect(planes[0])
            if rank == 0:
                renderer.resolve_device(planes[0].data_ptr(), rgbs[1].data_ptr(), full, stream)
                if download:
                    pinned.copy_(rgbs[1], non_blocking=True)
                    torch.cuda.current_stream().synchronize()  # the caller has the image before the next frame
        fence()
        ms = (time.perf_counter() - t) / singles * 1e3
        if distributed:
            tt = torch.tensor([ms], dtype=torch.float64, device="cpu" if args.rehearse else device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms = float(tt.item())
        return ms

    single_ms = frames_alone(False)
    single_d2h_ms = frames_alone(True)

    # One GPU: a stream of whole frames through the job API, every image downloaded (pinned host memory) on the
    # library's copy stream while the next frame is being traced -- the PCIe-inclusive period of a frame.
    stream_d2h_ms = None
    if not distributed:
        import numpy as np

        frames = max(3, min(args.steps, 8))
        host_images = [torch.empty(w * h * 3, dtype=torch.float32, pin_memory=True) for _ in range(2)]
        p_job = ffi.params(w, h, s, seed=1, flags=flags)  # (a stream too: successive jobs take turns on two library streams)

        def finish_into(job, k):
            rc = renderer.lib.rt_hip_job_finish(job, host_images[k % 2].data_ptr(), None)
            if rc != 0:
                raise ffi.HipError("rt_hip_job_finish failed (%d): %s" % (rc, renderer.lib.rt_hip_last_error().decode()))

        fence()
        # Three jobs in flight: the one being traced, the one whose first waves take over as that one's last waves
        # leave (the library gives successive jobs two streams in turn), and one more behind the first, so that what
        # follows a trace on its stream -- resolve pass, download, the host's next call -- is not what the third waits for.
        in_flight = 3
        warm = [renderer.start(scene, p_job) for _ in range(in_flight)]  # buffers of the job path exist from here on
        for k, job in enumerate(warm):
            finish_into(job, k)
        t = time.perf_counter()
        jobs = [renderer.start(scene, p_job) for _ in range(min(in_flight - 1, frames))]
        started = len(jobs)
        for k in range(frames):
            if started < frames:
                jobs.append(renderer.start(scene, p_job))
                started += 1
            finish_into(jobs.pop(0), k)
        stream_d2h_ms = (time.perf_counter() - t) / frames * 1e3
        del np

    # counters of this rank's last trace, summed over ranks
    counters = d_stats[0][:9].clone()
    if distributed:
        if args.rehearse:
            counters = counters.cpu()
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    names = ["samples", "ray_queries", "hit_queries", "shaded_bounces", "sphere_tests", "exact_tests", "lane_trips",
             "lane_rounds", "cone_tests"]
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
        demo_name = {1: "pretty_spheres", 2: "glass_spheres", 3: "synthetic_spheres"}[args.demo]
        key = "demo%d_%dx%d_s%d_g%d" % (args.demo, w, h, s, world) + ("_n%d" % args.spheres if args.demo == 3 else "") + \
              ("_grid" if args.grid else "")
        prof = profiled(key)
        traffic = prof.get("trace_kernel_bytes_per_launch")
        # what the kernel's algorithm moves through HBM: one column sum (12 B) written per sub-sample column; where the
        # kernel's waves add up their own pixels as they end, they read those sums back and write 12 B per pixel
        plan = renderer.describe(scene, p)  # (the frames of the timed stream)
        column_bytes = 12 * w * h * s * (2 if plan.wave_sums else 1) // (world if by_runs else 1) + (12 * w * h if plan.wave_sums else 0)
        # vector instructions the kernel issued: counted by SQ_INSTS_VALU (64 lanes each) when this workload has a
        # committed counter profile, else modelled from this run's device counters
        lane_ops = executed_lane_ops(k, scene.sphere_count, args.grid, scene.sphere_count <= 64)
        issued, issued_source = None, None
        if prof.get("valu_wave_instructions"):
            issued, issued_source = 64.0 * prof["valu_wave_instructions"], "SQ_INSTS_VALU x 64, " + str(prof.get("source"))
        elif lane_ops is not None:
            issued, issued_source = float(lane_ops), "model: this run's device counters x instructions per unit of work"
        seconds = kernel_ms * 1e-3
        executed_rate = issued / seconds if issued is not None else None
        fp32_frac = None
        if prof.get("fp32_wave_operations"):
            # FP32 arithmetic among the issued instructions (an FMA two operations), 64 lanes each, over the kernel time
            fp32_frac = round(64.0 * prof["fp32_wave_operations"] / seconds / (PEAK_FP32_VECTOR_TFLOPS * 1e12), 4)
        if args.grid:
            bound = "latency (grid walk: divergent cell walks, spheres read through the L2)"
        else:
            bound = "valu_fp32"
        hbm_bytes = traffic if traffic is not None else column_bytes
        bound_note = ("FP32 vector-ALU issue, neither HBM nor MFMA: %.3f GB of HBM traffic (%s) per launch of %.3g FLOP "
                      "is %.0f FLOP per byte and %.1f GB/s, no matrix work"
                      % (hbm_bytes / 1e9, "PMC counters" if traffic is not None else "algorithmic column sums",
                         kernel_flop, kernel_flop / max(hbm_bytes, 1), hbm_bytes / seconds / 1e9))
        roofline = {
            "bound": bound,
            "bound_note": bound_note,
            # executed work: vector instructions x 64 lanes over the kernel time, against one instruction per lane and
            # clock (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz)
            "achieved": round(executed_rate / 1e12, 2) if executed_rate is not None else None,
            "peak": round(PEAK_VALU_LANE_OPS / 1e12, 2),
            "unit": "T lane-instructions/s",
            "frac": round(executed_rate / PEAK_VALU_LANE_OPS, 4) if executed_rate is not None else None,
            "frac_source": issued_source,
            "modelled_frac": round(lane_ops / seconds / PEAK_VALU_LANE_OPS, 4) if lane_ops is not None else None,
            "fp32_flop_frac": fp32_frac,
            "lane_utilisation": prof.get("valu_lane_utilisation"),
            "measured_reference": measured_issue_rates(),
            # the algorithm's credit (20 FLOP per ray-sphere test the reference would run, 30 per hit, 250 per shaded
            # bounce) against the FP32 vector peak: how fast the algorithm's work gets done, not how busy the ALUs are
            "algorithmic_tflops": round(achieved, 2),
            "algorithmic_peak_tflops": PEAK_FP32_VECTOR_TFLOPS,
            "algorithmic_frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 4),
            "executed_over_algorithmic_tests": round(k["sphere_tests"] / max(k["ray_queries"] * scene.sphere_count, 1), 4),
            "traffic": traffic,
            "traffic_source": prof.get("source") if traffic is not None else None,
            "algorithmic_hbm_bytes": column_bytes,
            "kernel": "trace_kernel",
            "kernel_ms": round(kernel_ms, 3),
            "flop_per_launch": kernel_flop,
            "executed_filter_tests": k["sphere_tests"],
            "executed_cone_tests": k["cone_tests"],
            "executed_exact_tests": k["exact_tests"],
            "lane_trips": k["lane_trips"],
            "lane_rounds": k["lane_rounds"],
            "shading_trip_share": round(k["shaded_bounces"] / max(k["lane_trips"], 1), 4),
            "primary_round_lane_share": round(k["samples"] / max(k["lane_rounds"], 1), 4),
        }
        collective = {"backend": None, "world_size": 1}
        if distributed:
            collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                          "op": ("gather(dst=0) of each rank's own runs of 64 pixels, packed" if gather is not None
                                 else "reduce(SUM, dst=0) of 3*W*H float32"),
                          "bytes_per_rank_per_frame": gather.bytes_per_rank if gather is not None else 12 * w * h,
                          "trace_leaves_room_for_it": bool(flags & ffi.FLAG_LEAVE_ROOM)}
            if gather_note:
                collective["note"] = gather_note
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
                "workload": "%s %dx%d, %dx%d SSAA, %d spheres, %s"
                            % (demo_name, w, h, s, s, scene.sphere_count,
                               "one GPU" if not distributed else
                               ("runs of 64 pixels dealt out to %d GPUs, RCCL %s" % (world, "gather of the runs" if gather is not None else "sum-reduce")
                                if by_runs else "samples dealt out to %d GPUs, RCCL sum-reduce" % world)),
                "scene_seed": 1,
                "render_seed": 1,
                "max_depth": 10,
                "intersection": "uniform grid" if args.grid else "every sphere (LDS-resident or LDS-tiled)",
                "frames_in_flight": lanes,
                "pixel_sums": ("by the trace kernel's waves as they run out of work" if plan.wave_sums else
                               "inside the trace kernel's loop" if plan.kernel_sums else "fold and accumulate passes after the trace kernel"),
                "launch": {"blocks": plan.blocks, "lds_bytes_per_block": plan.lds_bytes, "batch_items": plan.batch_items,
                           "batch_min": plan.batch_min},
                "collective": collective,
            },
            "value_note": "throughput of a stream of frames, %d in flight, images left in HBM; the latency of one frame "
                          "is ms_per_frame_one_at_a_time.  With several in flight the trace kernels of successive frames "
                          "run side by side for a while (kernels_in_stream, from the committed kernel trace of this "
                          "command): that is how a step can be shorter than the trace kernel launched alone "
                          "(roofline.kernel_ms)" % lanes,
            "kernels_in_stream": overlap_profile(key) if lanes == 2 and not args.equal_priorities else None,
            "ms_per_frame_one_at_a_time": round(single_ms, 3),
            "ms_per_frame_one_at_a_time_with_d2h": round(single_d2h_ms, 3),
            "ms_per_frame_with_d2h": round(stream_d2h_ms, 3) if stream_d2h_ms is not None else None,
            "mrays_per_s": round(stats["ray_queries"] / (elapsed / args.steps) / 1e6, 1),
            "ray_queries_per_sample": round(stats["ray_queries"] / max(stats["samples"], 1), 3),
            "mean_radiance": round(mean_radiance, 5),
            "roofline": roofline,
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
