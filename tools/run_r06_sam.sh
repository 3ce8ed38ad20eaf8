#!/bin/bash
# Dev aid: the 500-bucket SAM leg of bench.py (configs[2]'s shape) under a set of tuning strings, on one box.
#   gpurun --timeout 1800 -- 'bash tools/run_r06_sam.sh tag "" "stream_readers=2" ...'
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for tune in "$@"; do
  i=$((i+1))
  name=${tune:-product}; name=${name//[=,]/_}_$i
  EMA_TUNING=$tune timeout 900 python3 "$root/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --spot-check 2000 > "$out/sam_$name.json" 2> "$out/sam_$name.err"
  echo "$name rc=$?"
  python3 - "$out/sam_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    l = d["bucket_files_to_sam"]; p = l["without_density_optimiser"]
    print("   -d: %.0f pairs/s, cpu-s/M %.3f %s" % (l["value"], l["host_cpu_seconds_per_million_pairs"], l["host_cpu_seconds_per_million_pairs_by_stage"]))
    print("   no -d: %.0f pairs/s, cpu-s/M %.3f; stage seconds %s" % (p["value"], p["host_cpu_seconds_per_million_pairs"], p["stage_seconds"]))
except Exception as e:
    print("   no line:", e)
PY
done
