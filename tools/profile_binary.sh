#!/bin/bash
# The two-distribution step at 256^3: settings compared, then rocprofv3 kernel stats and the FETCH / WRITE PMC passes
# of the default (separate passes, program straight after --)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for rep in 1 2; do
  for t in "blocked=0" "blocked=1,nt_store=0" "blocked=1,nt_store=1" "blocked=1,nt_store=1,xcd_group=8"; do
    python3 tools/bench_binary.py --tune "$t"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_binary_stats -o s -- python3 tools/bench_binary.py > $out/r03_binary_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/r03_binary_fetch -o f -- python3 tools/bench_binary.py --steps 15 > $out/r03_binary_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/r03_binary_write -o w -- python3 tools/bench_binary.py --steps 15 > $out/r03_binary_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_binary_soa_stats -o s -- python3 tools/bench_binary.py --tune blocked=0 > $out/r03_binary_soa_stats.log 2>&1
ls $out/r03_binary_stats $out/r03_binary_fetch $out/r03_binary_write
