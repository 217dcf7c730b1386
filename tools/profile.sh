#!/bin/bash
# Collects the profiles kept under profiles/ for one workload: a kernel-trace pass over the benchmark (one frame
# at a time, so that every trace launch in the trace runs alone) and separate counter passes over one frame.
# Run on the GPU box from the repository root:
#   bash tools/profile.sh <tag> [demo width height samples [spheres [grid]]]     (default: 1 1920 1080 16)
# Results land in gpurun_out/prof_<tag>/; tools/summarise_profile.py condenses them into profiles/.
set -u
TAG=${1:-r2}
DEMO=${2:-1}; W=${3:-1920}; H=${4:-1080}; S=${5:-16}; SPHERES=${6:-10000}; GRID=${7:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
echo "$DEMO $W $H $S $SPHERES $GRID" > "$OUT/workload.txt"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--steps 3 --warmup 1 --sync-reduce --no-cpu-baseline --demo $DEMO --width $W --height $H --samples $S --spheres $SPHERES"
[ -n "$GRID" ] && BENCH_ARGS="$BENCH_ARGS --grid"

timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- \
  python3 "$ROOT/bench.py" $BENCH_ARGS > "$OUT/bench_stdout.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/bench_stdout.log"; exit 1; }

FLAGS=0; [ -n "$GRID" ] && FLAGS=2
ONE="python3 $ROOT/tools/render_once.py $DEMO $W $H $S 1 $SPHERES $FLAGS"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $ONE > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
# what the vector instructions are (wave instructions by class)
pass sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU
# how many of their lanes were switched on: rocprofv3's VALUUtilization, SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
pass sq4 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo "profiles written to $OUT"
