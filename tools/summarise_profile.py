"""Turns the raw rocprofv3 output of tools/profile.sh (gpurun_out/prof_<tag>/) into the small files kept under
profiles/: the kernel-trace statistics, the bench line of the same run, a per-kernel counter summary and the
HBM traffic per trace launch that bench.py reports as roofline.traffic.

python tools/summarise_profile.py <tag>
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
workload = ["1", "1920", "1080", "16", "10000"]
if os.path.exists(os.path.join(src, "workload.txt")):
    workload = open(os.path.join(src, "workload.txt")).read().split()
demo, w, h, s_row = (int(x) for x in workload[:4])
grid = len(workload) > 5

shutil.copy(os.path.join(src, "bench", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
with open(os.path.join(src, "bench_stdout.log")) as f:
    lines = [l for l in f if l.startswith('{"metric"')]
with open(os.path.join(dst, tag + "_bench_line.json"), "w") as f:
    f.write(lines[-1])

totals = {}
with open(os.path.join(dst, tag + "_pmc_summary.txt"), "w") as out:
    out.write("rocprofv3 --pmc passes over `python3 tools/render_once.py %s` (one frame: demo, width, height, samples per\n"
              "row, repeat, spheres of demo 3, flags), collected by tools/profile.sh; values summed over the dispatches of\n"
              "each kernel in the run (one trace launch).  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them\n"
              "(FETCH_SIZE uncorrected).\n\n" % " ".join(workload[:4] + ["1", workload[4], "2" if grid else "0"]))
    for name in ("sq1", "sq2", "sq3", "sq4", "fetch", "write"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        if not os.path.exists(os.path.join(src, name, name + "_counter_collection.csv")):
            continue  # (profiles taken before the pass existed)
        with open(os.path.join(src, name, name + "_counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_rocclr_\w+)", r["Kernel_Name"])
                agg[m.group(1) if m else r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
        for kernel, counters in agg.items():
            if "rocclr" in kernel:
                continue
            out.write("[%s] %s\n" % (name, kernel))
            for c, x in sorted(counters.items()):
                out.write("    %-28s %.0f\n" % (c, x))
                if "trace_kernel" in kernel:
                    totals[(name, c)] = x
                    totals.setdefault(c, x)  # the first pass that holds a counter names it (SQ_INSTS_VALU is in three)

traffic_path = os.path.join(dst, "traffic.json")
traffic = {}
if os.path.exists(traffic_path):
    with open(traffic_path) as f:
        traffic = json.load(f)
key = "demo%d_%dx%d_s%d_g1" % (demo, w, h, s_row) + ("_n%s" % workload[4] if demo == 3 else "") + ("_grid" if grid else "")
traffic[key] = {
    # gfx950: FETCH_SIZE counts half the bytes of wide reads; WRITE_SIZE is exact for 16-byte stores
    "trace_kernel_bytes_per_launch": int((2.0 * totals["FETCH_SIZE"] + totals["WRITE_SIZE"]) * 1024),
    "fetch_size_kib": totals["FETCH_SIZE"],
    "write_size_kib": totals["WRITE_SIZE"],
    "valu_wave_instructions": totals.get("SQ_INSTS_VALU"),
    "source": "profiles/%s_pmc_summary.txt" % tag,
}
if ("sq3", "SQ_INSTS_VALU_FMA_F32") in totals:
    # FP32 arithmetic among the vector instructions: an FMA is two operations per lane
    traffic[key]["fp32_wave_operations"] = (2.0 * totals["SQ_INSTS_VALU_FMA_F32"] + totals["SQ_INSTS_VALU_ADD_F32"] +
                                            totals["SQ_INSTS_VALU_MUL_F32"])
    traffic[key]["valu_wave_instructions_in_that_pass"] = totals[("sq3", "SQ_INSTS_VALU")]
if ("sq4", "SQ_THREAD_CYCLES_VALU") in totals:
    # lanes switched on per vector instruction, as rocprofv3 defines VALUUtilization: thread-cycles over (cycles
    # spent on vector instructions x 64 lanes)
    cycles = totals.get(("sq4", "SQ_ACTIVE_INST_VALU"))
    traffic[key]["valu_lane_utilisation"] = round(totals[("sq4", "SQ_THREAD_CYCLES_VALU")] / (64.0 * cycles), 4) if cycles else None
with open(traffic_path, "w") as f:
    json.dump(traffic, f, indent=1)
print("profiles/%s_* written; trace kernel: %s" % (tag, {k: totals[k] for k in sorted(k for k in totals if isinstance(k, str))}))
print(traffic[key])
