#!/bin/bash
# One rocprofv3 --pmc pass per counter group (never combined with tracing domains other than --kernel-trace).
#   bash tools/profile_pmc.sh <tag> "<group1 counters>;<group2 counters>;..." [bench args...]
# Writes gpurun_out/<tag>_pmc<k>/ ; summarise with tools/pmc_table.py <tag> <kernel-substring>.
set -e
tag=$1; groups=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
k=0
IFS=';' read -ra G <<< "$groups"
for g in "${G[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/${tag}_pmc$k -o p -- python3 bench.py "$@" --cpu-baseline 0 > $out/${tag}_pmc$k.log 2>&1
  k=$((k+1))
done
ls -d $out/${tag}_pmc*
