set -e
for cfg in "7 1 phi" "7 1 grad" "27 2 phi" "27 2 grad" "27 3 grad" "7 4 phi"; do
  set -- $cfg
  python bench.py --size 256 256 256 --nhalo 2 --fe symmetric --fe-grad $1 --fe-order $2 --fe-route $3 --steps 50 --warmup 5 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/fe_sweep.jsonl
done
