# Vector, scalar and LDS instruction counts of the headline trace launch with the pixel sums taken inside the kernel
# (RT_HIP_KERNEL_SUMS) and without: bash tools/pmc_ab.sh (on the GPU box, from the repository root).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in fused unfused; do
  if [ $v = fused ]; then export RT_HIP_KERNEL_SUMS=1; else unset RT_HIP_KERNEL_SUMS; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/$v -o $v -- python3 $ROOT/tools/render_once.py 1 1920 1080 16 1 10000 0 > $OUT/$v.log 2>&1 || { echo pass $v failed; tail -5 $OUT/$v.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for v in ("fused","unfused"):
    tot = collections.defaultdict(float)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    print(v, {k: "%.4g" % x for k, x in sorted(tot.items())})
PY
