#!/bin/bash
# Dev aid: bench.py (20 steps) over lean seeding budgets (extends per read before the read goes to the full-capacity tier), one box.
#   gpurun --timeout 1500 -- 'bash tools/run_r06_budget.sh tag 0 2048 8192 16384'      (0 = the engine's default, 4096)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  EMA_LEAN_SEED_EXTENDS=$b timeout 600 python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-sam-leg > "$out/bench_budget_$b.json" 2> "$out/bench_budget_$b.err"
  echo "budget $b rc=$?"
  python3 - "$out/bench_budget_$b.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = d.get("roofline") or {}
    print("   value %.0f  ms_per_step %.2f  isolated: %s" % (d["value"], d["ms_per_step"], rf.get("all_kernels_ms_isolated")))
    print("   in the steps:", rf.get("all_kernels_ms"), " redone pairs:", d["bucket_stats"].get("redone_pairs"))
except Exception as e:
    print("   no line:", e)
PY
done
