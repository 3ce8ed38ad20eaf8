#!/bin/bash
# Dev aid: the default bench command under a few engine knobs, one GPU box, one index build (the first run builds it).
#   gpurun --timeout 1500 -- 'bash tools/run_knobs.sh tag'
tag=${1:-knobs}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
ulimit -c 0
run() {      # name, env..., -- bench args
	name=$1; shift
	envs=()
	while [ "$1" != "--" ]; do envs+=("$1"); shift; done
	shift
	env "${envs[@]}" timeout 900 python3 "$root/bench.py" --steps 5 --no-cpu-baseline "$@" > "$out/$name.json" 2> "$out/$name.err"
	python3 - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:28s} {d['value'] / 1e6:6.3f} M pairs/s  {d['ms_per_step']:7.1f} ms/step  iso: {r['all_kernels_ms_isolated']}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run default X=1 --
run seed_blocks3 EMA_SEED_BLOCKS_PER_CU=3 --
run streams4 X=1 -- --streams 4
