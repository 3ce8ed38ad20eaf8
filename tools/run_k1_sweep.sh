#!/bin/bash
# Dev aid, one GPU-box call: the bench workload once, then K1's isolated time (tools/gpu_k1_profile.py, product build) per environment.
#   gpurun --timeout 1500 -- 'bash tools/run_k1_sweep.sh "" "EMA_SEED_ORDER=8,12"'
root=${GRAFT_REPO_ROOT:-$(pwd)}
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-sam-leg > /tmp/sweep_bench.json 2> /tmp/sweep_bench.err; echo "bench rc=$?"
for v in "$@"; do
  echo "=== ${v:-(defaults)}"
  env $v timeout 600 python3 "$root/tools/gpu_k1_profile.py" 2>&1 | grep -E "isolated" | tail -1
done
