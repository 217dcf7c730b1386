#!/bin/bash
# Two builds of the library through bench.py, taking turns: tools/ab_bench.sh <lib dir A> <lib dir B> <rounds> -- <bench.py arguments>
a=$1; b=$2; rounds=$3; shift 4
for r in $(seq $rounds); do
  for d in $a $b; do
    RT_AMD_LIB_DIR=$PWD/$d timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['ms_per_step'], 'ms per step,', d['ms_per_frame_one_at_a_time'], 'one at a time, kernel', d['roofline']['kernel_ms'])" || exit 1
  done
done
