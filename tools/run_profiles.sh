#!/bin/bash
# Dev aid: the bench line, the rocprofv3 kernel statistics and the FETCH_SIZE / WRITE_SIZE passes of the default bench
# command, in one go on a GPU box (the index built by the first run is reused by the others).  Usage:
#   gpurun --timeout 1500 -- 'bash tools/run_profiles.sh r01g'
# Results land in gpurun_out/<tag>/; copy the summaries into profiles/ (see profiles/README.md).
tag=${1:-run}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 "$root/bench.py" > "$out/bench_plain.json" 2> "$out/bench_plain.err"
echo "plain: rc=$?"; cut -c1-300 "$out/bench_plain.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o runc -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
echo "stats: rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o runc -- python3 "$root/bench.py" --steps 2 --warmup 0 --no-cpu-baseline > "$out/bench_fetch.json" 2> "$out/bench_fetch.err"
echo "fetch: rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o runc -- python3 "$root/bench.py" --steps 2 --warmup 0 --no-cpu-baseline > "$out/bench_write.json" 2> "$out/bench_write.err"
echo "write: rc=$?"
# the per-dispatch traces are large; keep the statistics and the counter tables
find "$out" -name "*kernel_trace.csv" -size +8M -delete
ls -la "$out" "$out"/*/ 2>/dev/null | head -40
