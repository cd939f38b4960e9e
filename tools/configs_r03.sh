#!/bin/bash
# The bench lines of the BASELINE configurations on one MI355X, one call (profiles/r03_configs.jsonl):
#   1: D3Q19 BGK 64^3   2: D3Q19 M10 256^3 (the default line)   3 per GPU: a 32 x 256 x 256 slab through a 1-rank RCCL ring
#   4: D3Q19 + symmetric free energy 128^3 (one kernel)          5 per GPU: D3Q27 M10 64 x 512 x 256
# plus the halo mode, the every-step hydro traffic and the reference-equivalent three stages
run() { python bench.py "$@" --cpu-baseline 0 2>/dev/null | grep '^{' | tail -1; }
run --config 1 --steps 2000 --warmup 50
run --steps 200 --warmup 20
run --size 32 256 256 --selfring 1 --steps 400 --warmup 20
run --config 4 --steps 300 --warmup 20
run --config 5 --steps 200 --warmup 20
run --mode fused_halo --hydro 1 --steps 100 --warmup 10
run --mode fused_halo --hydro lazy --steps 100 --warmup 10
run --hydro 1 --steps 100 --warmup 10
run --mode eager --hydro 1 --steps 40 --warmup 5
run --fe symmetric --nhalo 2 --hydro 1 --size 256 256 256 --steps 100 --warmup 10
