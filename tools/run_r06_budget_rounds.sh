#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r06l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 600 python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sam-leg > $out/bench_$name.json 2> $out/bench_$name.err
  python3 - $out/bench_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=d["roofline"]
print(sys.argv[2], "ms_per_step %.2f value %.0f" % (d["ms_per_step"], d["value"]), "iso", rf.get("all_kernels_ms_isolated"), "redone", d["bucket_stats"].get("redone_pairs"))
PY
}
run base EMA_X=1
run b8192_r3 EMA_LEAN_SEED_EXTENDS=8192 EMA_TUNING=seed_rounds=3
run b8192_r3_p16 EMA_LEAN_SEED_EXTENDS=8192 EMA_TUNING=seed_rounds=3,seed_park=16
run b6144 EMA_LEAN_SEED_EXTENDS=6144
run b8192_r4_p32 EMA_LEAN_SEED_EXTENDS=8192 EMA_TUNING=seed_rounds=4,seed_park=32
