#!/bin/bash
# Kernel trace of the default benchmark (two frames in flight): shows the trace kernels of successive frames on
# two hardware queues, each starting before its predecessor has ended.  Run on the GPU box from the repository
# root; tools/summarise_overlap.py turns the result into profiles/<tag>_overlap.txt.
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/overlap_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o ov -- \
  python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/stdout.log" 2>&1 || { tail -5 "$OUT/stdout.log"; exit 1; }
grep '^{"metric"' "$OUT/stdout.log" | tail -1 | cut -c1-200
