#!/bin/bash
# Dev aid: bench.py (20 steps, kernels + delivery only) over a set of engine builds on ONE GPU box, so that the numbers compare.
#   gpurun --timeout 2400 -- 'bash tools/run_r05_ab.sh tag "" chainlds tune:seed_split3=0 ...'      ("" = the product build libema_engine.so; tune:k=v = it with EMA_TUNING)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  name=${v:-product}
  lib=libema_engine${v:+_$v}.so
  tune=
  if [[ $v == tune:* ]]; then tune=${v#tune:}; lib=libema_engine.so; name=${tune//[=,]/_}; fi
  name=${name}_$i      # "tune:seed_split3=0": the product build with a tuning string
  EMA_TUNING=$tune EMA_ENGINE_LIB=$lib timeout 600 python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-sam-leg > "$out/bench_$name.json" 2> "$out/bench_$name.err"
  echo "$name rc=$?"
  python3 - "$out/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    er = d.get("engine_resident") or {}
    iso = (d.get("roofline") or {}).get("isolated") or {}
    print("   value %.0f  ms_per_step %.2f  engine_resident ms %.2f  isolated: %s" % (d["value"], d["ms_per_step"], er.get("ms_per_step", 0), (d.get("roofline") or {}).get("all_kernels_ms_isolated")))
    print("   ms between steps reaching the sink:", d["config"]["method"].get("ms_between_steps_reaching_the_sink"), " host:", {k: v for k, v in (d.get("host") or {}).items() if isinstance(v, (int, float))})
except Exception as e:
    print("   no line:", e)
PY
done
