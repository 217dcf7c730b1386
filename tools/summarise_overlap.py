"""Condenses the kernel trace of tools/overlap_trace.sh into profiles/<tag>_overlap.txt: one line per trace-kernel
launch of the timed region with its hardware queue, start and end -- and into the entry of profiles/overlap.json
that bench.py quotes as `kernels_in_stream`: over the launches that ran beside their successor, the average
duration of a trace kernel, the average distance between successive starts and the average time two successive
kernels ran side by side.

python tools/summarise_overlap.py <tag>
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "overlap_" + tag)
rows = [r for r in csv.DictReader(open(os.path.join(src, "ov_kernel_trace.csv"))) if "trace_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(os.path.join(src, "stdout.log")) as f:
    line = [l for l in f if l.startswith('{"metric"')][-1]
out = os.path.join(ROOT, "profiles", tag + "_overlap.txt")
with open(out, "w") as f:
    f.write("rocprofv3 --kernel-trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline (tools/overlap_trace.sh):\n"
            "the trace_kernel launches in start order.  Frames alternate between two streams of different priority, which\n"
            "HIP maps to two hardware queues; a frame's kernel starts while the previous frame's is still running, so the\n"
            "period between starts is shorter than a kernel.  The launches after them are the one-at-a-time frames (one\n"
            "queue, back to back), the stream of downloaded frames through the job API (the library's two streams) and the\n"
            "roofline timing of the same run.\n\n")
    f.write("queue  start ms   end ms     duration ms  start - previous start ms\n")
    t0 = int(rows[0]["Start_Timestamp"])
    prev = None
    for r in rows:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        f.write("%5s  %9.3f  %9.3f  %10.3f  %s\n" % (r.get("Queue_Id", "?"), s, e, e - s,
                                                      "" if prev is None else "%9.3f" % (s - prev)))
        prev = s
    f.write("\nbench line of the same run:\n" + line)
print(open(out).read())

# the launches that ran beside their successor (the stream of frames; the launches after it run one at a time)
spans = [(int(r["Start_Timestamp"]) / 1e6, int(r["End_Timestamp"]) / 1e6) for r in rows]
median = sorted(e - s for s, e in spans)[len(spans) // 2]
# (not the first launches of the process, which include loading the code object)
# ... and only the first such run: further down the job API's stream of downloaded frames overlaps its kernels too
pairs = []
for a, b in zip(spans, spans[1:]):
    if b[0] < a[1]:
        if max(a[1] - a[0], b[1] - b[0]) < 1.3 * median:
            pairs.append((a, b))
    elif pairs:
        break
if pairs:
    bench = json.loads(line)
    cfg = bench["config"]["workload"]
    key = "demo1_1920x1080_s16_g1"  # the default command of tools/overlap_trace.sh
    path = os.path.join(ROOT, "profiles", "overlap.json")
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[key] = {
        "kernel_ms": round(sum(a[1] - a[0] for a, _ in pairs) / len(pairs), 3),
        "start_to_start_ms": round(sum(b[0] - a[0] for a, b in pairs) / len(pairs), 3),
        "side_by_side_ms": round(sum(a[1] - b[0] for a, b in pairs) / len(pairs), 3),
        "pairs": len(pairs),
        "workload": cfg,
        "source": "profiles/%s_overlap.txt" % tag,
    }
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
    print(data[key])
