cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in eager inplace; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/probe_$m -o s -- python3 bench.py --mode $m --hydro 1 --steps 40 --warmup 5 --cpu-baseline 0 > gpurun_out/probe_$m.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/probe_$m/**/s_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("$m", r["Name"][:80], r["Calls"], r["AverageNs"])
PY
done
