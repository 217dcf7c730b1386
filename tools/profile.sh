#!/bin/bash
# Collects the profiles kept under profiles/: one kernel-trace pass over the benchmark and separate
# counter passes over one 1080p 16x16 frame.  Run on the GPU box from the repository root:
#   bash tools/profile.sh <tag>
# Results land in gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- \
  python3 "$ROOT/bench.py" --steps 5 --warmup 2 --sync-reduce > "$OUT/bench_stdout.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/bench_stdout.log"; exit 1; }

ONE="python3 $ROOT/tools/render_once.py 1 1920 1080 16"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $ONE > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo "profiles written to $OUT"
