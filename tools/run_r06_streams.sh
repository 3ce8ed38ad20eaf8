#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r06k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for st in 3 2 4 3; do
  timeout 600 python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sam-leg --streams $st > $out/bench_streams_$st.json 2> $out/bench_streams_$st.err
  python3 - $out/bench_streams_$st.json $st <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=d["roofline"]
print("streams", sys.argv[2], "ms_per_step %.2f value %.0f" % (d["ms_per_step"], d["value"]), rf.get("all_kernels_ms"))
PY
done
