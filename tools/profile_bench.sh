#!/bin/bash
# Profile the default bench line on the GPU box: kernel trace + the two PMC
# passes the guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit in one
# pass; never combined with tracing domains other than --kernel-trace).
#   bash tools/profile_bench.sh <tag> [bench args...]
# Writes gpurun_out/<tag>_{bench.json,stats,fetch,write}/ ; summarise with
# tools/pmc_summary.py.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
python3 bench.py "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- python3 bench.py "$@" --cpu-baseline 0 > $out/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -o f -- python3 bench.py "$@" --cpu-baseline 0 --steps 20 --warmup 2 > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -o w -- python3 bench.py "$@" --cpu-baseline 0 --steps 20 --warmup 2 > $out/${tag}_write.log 2>&1
ls $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write
