#!/bin/bash
# Dev aid: A/B of engine knobs with the PRODUCT builds at the default scale, one GPU-box call: bench.py once per variant (the
# first run builds genome, index and batches; the others find them in the bench workdir), then value, the steady state, one pass
# alone and the isolated K1..K4 times of each.
#   gpurun --timeout 2400 -- 'bash tools/run_ab.sh r03ab "" "EMA_HEAVY_CHAINS=8" "EMA_HEAVY_CHAINS=2"'
tag=${1:-ab}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i + 1))
  echo "=== variant $i: ${v:-(defaults)}"
  env $v timeout 900 python3 "$root/bench.py" --steps ${AB_STEPS:-10} --no-cpu-baseline --no-sam-leg > "$out/v$i.json" 2> "$out/v$i.err"; echo "rc=$?"
  python3 - "$out/v$i.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("value %.3f M  resident %.3f M (%.1f ms)  single pass %.1f ms  isolated %s  in the timed steps %s" % (
        d["value"] / 1e6, d["engine_resident"]["value"] / 1e6, d["engine_resident"]["ms_per_step"], d["engine_resident"]["single_pass"]["ms"],
        r["all_kernels_ms_isolated"], r["all_kernels_ms"]))
except Exception as e:
    print("no line:", e)
PY
  grep "full-capacity tier" "$out/v$i.err" | tail -1
done
