#!/bin/bash
# Dev aid: the seeding tests, then bench.py (20 steps, kernels + delivery only) over a set of tuning strings on ONE GPU box, so that the numbers compare.
#   gpurun --timeout 2400 -- 'bash tools/run_r06_ab.sh tag [--tests "tests/test_gpu_seed.py ..."] "" seed_blocks_per_cu=3 seed_park=32 ...'   ("" = no tuning)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
if [[ $1 == --tests ]]; then
  ( cd "$root" && timeout 1500 python3 -m pytest $2 -x -q -m gpu > "$out/pytest.log" 2>&1 ); echo "pytest rc=$?"; tail -5 "$out/pytest.log"
  shift 2
fi
i=0
for tune in "$@"; do
  i=$((i+1))
  name=${tune:-product}; name=${name//[=,]/_}_$i
  EMA_TUNING=$tune timeout 600 python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-sam-leg > "$out/bench_$name.json" 2> "$out/bench_$name.err"
  echo "$name rc=$?"
  python3 - "$out/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    er = d.get("engine_resident") or {}
    rf = d.get("roofline") or {}
    print("   value %.0f  ms_per_step %.2f  engine_resident ms %.2f  isolated: %s" % (d["value"], d["ms_per_step"], er.get("ms_per_step", 0), rf.get("all_kernels_ms_isolated")))
    print("   in the steps:", rf.get("all_kernels_ms"), " redone:", d["config"].get("pairs_redone_by_full_tier"))
except Exception as e:
    print("   no line:", e)
PY
done
