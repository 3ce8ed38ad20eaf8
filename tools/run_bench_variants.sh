#!/bin/bash
# Dev aid: the bench line under several environments / flags in one GPU-box call (genome, index and batches are built by the first).
#   gpurun --timeout 2400 -- 'bash tools/run_bench_variants.sh r02d "ENV=.. --flags" ...'     (each argument: VAR=val ... then bench flags)
tag=${1:-bv}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i + 1))
  envs=(); flags=()
  for w in $v; do if [[ $w == *=* && $w != --* ]]; then envs+=("$w"); else flags+=("$w"); fi; done
  echo "=== variant $i: $v"
  env "${envs[@]}" timeout 1500 python3 "$root/bench.py" --no-cpu-baseline "${flags[@]}" > "$out/b$i.json" 2> "$out/b$i.err"; echo "rc=$?"
  python3 - "$out/b$i.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d["roofline"]
    print("value", d["value"], "ms/step", d["ms_per_step"], "boundary", (d["boundary"] or {}).get("value"), "resident", (d["engine_resident"] or {}).get("value"))
    print("kernels timed", r["all_kernels_ms"], "isolated", r.get("all_kernels_ms_isolated"))
except Exception as e:
    print("no line:", e)
PY
done
