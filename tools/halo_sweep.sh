#!/bin/bash
# The SoA -> SoA pull kernel of LBMI_MODE_FUSED_HALO (walls, colloids, every demoted run), D3Q19 M10 256^3, hydro
# arrays read and written by every collision: launch tuning (round 2 swept only the blocked order)
run() { python bench.py "$@" --steps 100 --warmup 10 --cpu-baseline 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'])"; }
echo -n "fused (blocked) hydro 1: "; run --mode fused --hydro 1
echo -n "fused_soa hydro 1: "; run --mode fused_soa --hydro 1
echo -n "fused_halo hydro 1 (defaults): "; run --mode fused_halo --hydro 1
echo -n "fused_halo hydro lazy (defaults): "; run --mode fused_halo --hydro lazy
for nt in 0 2; do for cap in 32768 49152 65536 98304; do for grp in 8 32 128; do
  echo -n "fused_halo hydro 1 nt_store=$nt lds_cap=$cap xcd_group=$grp: "; run --mode fused_halo --hydro 1 --tune nt_store=$nt,lds_cap=$cap,xcd_group=$grp
done; done; done
