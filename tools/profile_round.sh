#!/bin/bash
# All profiles of a round in one go (on the GPU box, from the repository root): kernel trace + counter passes for the
# workloads of the baseline, and the kernel trace of the default benchmark (two frames in flight).
#   bash tools/profile_round.sh <round tag, e.g. r3> [part]     part 1: headline, overlap, the two 1024x768 frames;
#                                                                part 2: the 10 001-sphere frame (tiles, grid), glass 4K
# then, here: python tools/summarise_profile.py <tag>g ... ; python tools/summarise_overlap.py <tag>
set -u
R=${1:-r3}; PART=${2:-1}
if [ "$PART" = 1 ]; then
  bash tools/profile.sh ${R}g 1 1920 1080 16 || exit 1
  bash tools/overlap_trace.sh $R || exit 1
  bash tools/profile.sh ${R}l 1 1024 768 2 || exit 1
  bash tools/profile.sh ${R}i 2 1024 768 10 || exit 1
else
  bash tools/profile.sh ${R}h 3 1920 1080 8 10000 || exit 1
  bash tools/profile.sh ${R}k 3 1920 1080 8 10000 grid || exit 1
  bash tools/profile.sh ${R}j 2 3840 2160 32 || exit 1
fi
