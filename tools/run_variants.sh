#!/bin/bash
# Dev aid: A/B of engine knobs on one bench batch at the default scale (one GPU-box call): builds genome, index and two batches
# through bench.py once, then runs tools/gpu_readlog.py (isolated K1..K4 times of one pass + K2b's per-read log) per variant.
#   gpurun --timeout 2400 -- 'bash tools/run_variants.sh r02c "EMA_SEED_BLOCKS=2" "EMA_STREAMS=4" ...'
tag=${1:-var}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
timeout 1200 python3 "$root/bench.py" --steps 2 --warmup 0 --batches 2 --no-cpu-baseline --no-extras > "$out/bench0.json" 2> "$out/bench0.err"; echo "bench0: rc=$?"; tail -3 "$out/bench0.err"
i=0
for v in "$@"; do
  i=$((i + 1))
  echo "=== variant $i: $v"
  env $v EMA_PHASE_PROFILE=2 timeout 600 python3 "$root/tools/gpu_readlog.py" "${tag}_v$i" > "$out/v$i.txt" 2>&1; echo "rc=$?"
  grep -E "isolated|handed over|^K2 phase|^K1:|\[ *(40|200)," "$out/v$i.txt" | tail -8
done
