#!/bin/bash
# Dev aid: rocprofv3 kernel statistics of bench.py (3 steps, kernels + delivery only) under a tuning string.
#   gpurun --timeout 900 -- 'bash tools/run_r06_stats.sh tag "" ["seed_blocks_per_cu=3" ...]'
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for tune in "$@"; do
  i=$((i+1))
  export EMA_TUNING=$tune
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats$i" -o run -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-sam-leg > "$out/bench_stats$i.json" 2> "$out/bench_stats$i.err"
  echo "stats$i ($tune): rc=$?"
  f=$(find "$out/stats$i" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("   %-60s calls %5s  avg %10.3f ms  total %9.1f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
find "$out" -name "*kernel_trace.csv" -size +8M -delete
