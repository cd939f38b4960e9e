set -e
for tp in 1 4 16; do
  python bench.py --timing-period $tp --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_timing.jsonl
  python bench.py --timing-period $tp --size 32 256 256 --selfring 1 --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_timing.jsonl
  python bench.py --timing-period $tp --size 32 256 256 --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_timing.jsonl
done
