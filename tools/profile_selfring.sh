#!/bin/bash
# Kernel trace of one slab of the 8-GPU strong-scaling case (32 x 256 x 256) through a 1-rank RCCL ring:
# gpurun_out/<tag>_trace/ ; print the timeline with tools/timeline.py.
#   bash tools/profile_selfring.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_trace -o t -- python3 bench.py --size 32 256 256 --selfring 1 --steps 60 --warmup 10 --cpu-baseline 0 --timing-period 1000 "$@" > $out/${tag}_trace.log 2>&1
ls $out/${tag}_trace
