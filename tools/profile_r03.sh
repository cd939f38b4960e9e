#!/bin/bash
# Round-3 evidence: the default bench line, config 4 and the halo mode, each with rocprofv3 kernel stats and the two
# PMC passes (tools/profile_bench.sh); summaries go to profiles/ with tools/pmc_summary.py
set -e
bash tools/profile_bench.sh r03_default --steps 200 --warmup 20
bash tools/profile_bench.sh r03_cfg4 --config 4 --steps 200 --warmup 20
bash tools/profile_bench.sh r03_halo --mode fused_halo --hydro 1 --steps 100 --warmup 10
