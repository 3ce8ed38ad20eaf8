#!/bin/bash
# Dev aid, one GPU-box call: builds the bench workload (2 batches), then K1's diagnostics on it -- tick statistics under a few knobs,
# the per-launch durations of the series from a rocprofv3 kernel trace, and the oracle's extends-per-read distribution.
#   gpurun --timeout 1500 -- 'bash tools/run_k1.sh r04b "" "EMA_SEED_TAIL=0"'
tag=${1:-k1}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-sam-leg > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
tail -3 "$out/bench.err"
i=0
for v in "$@"; do
  i=$((i + 1))
  echo "=== variant $i: ${v:-(defaults)}"
  env $v timeout 600 python3 "$root/tools/gpu_k1_profile.py" 2>&1 | grep -E "isolated|flagged"
  env $v EMA_PHASE_PROFILE=1 timeout 600 python3 "$root/tools/gpu_k1_profile.py" 2>&1 | grep -E "^K1:|serial pass" | tail -3
done
echo "=== kernel trace (defaults)"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 "$root/tools/gpu_k1_profile.py" > "$out/trace.log" 2>&1
python3 - "$out/trace" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "seed" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print(f, len(rows), "seed launches")
    for r in rows[-12:]:
        print("  %-40s %8.3f ms  grid %s" % (r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "")))
PY
echo "=== lean capacity flags"
for c in "" "64 64 256" "96 96 384" "48 96 384" "96 48 192"; do timeout 600 python3 "$root/tools/gpu_capdist.py" $c 2>&1 | tail -8; done
echo "=== extends per read (oracle, CPU)"
timeout 900 python3 "$root/tools/gpu_k1_profile.py" dist 20000
