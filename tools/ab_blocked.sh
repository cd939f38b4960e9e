set -e
for rep in 1 2; do
for m in fused_soa fused; do
  python bench.py --mode $m --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_blocked.jsonl
  python bench.py --mode $m --hydro 0 --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_blocked.jsonl
done
done
python bench.py --mode fused --nvel 27 --size 64 512 256 --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_blocked.jsonl
python bench.py --mode fused_soa --nvel 27 --size 64 512 256 --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | tail -1 >> gpurun_out/ab_blocked.jsonl
