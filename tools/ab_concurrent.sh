set -e
for rep in 1 2; do
for c in 0 1; do
  python bench.py --tune x_concurrent=$c --size 32 256 256 --selfring 1 --steps 300 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_conc.jsonl
  python bench.py --tune x_concurrent=$c --size 64 256 256 --selfring 1 --steps 300 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_conc.jsonl
done
done
