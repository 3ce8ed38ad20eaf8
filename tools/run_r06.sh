#!/bin/bash
# Dev aid: one GPU-box call of round 6 -- GPU test suite, the random-gather microbenchmark, the bench line, the oracle's
# seeding profile and K2b's per-read log at the default scale, rocprofv3 kernel statistics and the PMC passes.  Usage:
#   gpurun --timeout 2700 -- 'bash tools/run_r06.sh r06a [tests] [bench] [prof] [pmc]'
# Results land in gpurun_out/<tag>/; copy the summaries into profiles/ (see profiles/README.md).
tag=${1:-r06}; shift
what="${*:-tests bench prof pmc}"
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
ulimit -c 0
export EMA_VERBOSE=1
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras"
if [[ $what == *tests* ]]; then
  (cd "$root" && timeout 1800 python3 -m pytest tests -m gpu -x -q) > "$out/pytest.log" 2>&1; echo "pytest: rc=$?"; tail -5 "$out/pytest.log"
fi
if [[ $what == *bench* ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_mb "$root/tools/gather_microbench.hip" 2> /dev/null && timeout 300 /tmp/gather_mb > "$out/gather_mb.txt" 2>&1; echo "gather: rc=$?"
  EMA_INDEX_PROF=1 timeout 1500 python3 "$root/bench.py" > "$out/bench.json" 2> "$out/bench.err"; echo "bench: rc=$?"; cut -c1-600 "$out/bench.json"; tail -25 "$out/bench.err"
  timeout 300 python3 "$root/tools/cpu_seed_profile.py" > "$out/seed_profile.txt" 2>&1; echo "seedprof: rc=$?"
  EMA_PHASE_PROFILE=2 timeout 600 python3 "$root/tools/gpu_readlog.py" "$tag" > "$out/readlog.txt" 2>&1; echo "readlog: rc=$?"; tail -40 "$out/readlog.txt"
  timeout 600 python3 "$root/tools/gpu_k2_profile.py" > "$out/k2_profile.txt" 2>&1; echo "k2 profile: rc=$?"; cat "$out/k2_profile.txt"
  timeout 1500 python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_20steps.json" 2> "$out/bench_20steps.err"; echo "bench 20 steps: rc=$?"; cut -c1-300 "$out/bench_20steps.json"
  EMA_PHASE_PROFILE=1 timeout 600 python3 "$root/tools/gpu_k1_profile.py" > "$out/k1_profile.txt" 2>&1; echo "k1 profile: rc=$?"; grep -E "^K1:|isolated" "$out/k1_profile.txt" | tail -3
  timeout 600 python3 "$root/tools/gpu_capdist.py" > "$out/capdist.txt" 2>&1; cat "$out/capdist.txt"
fi
if [[ $what == *sam* ]]; then
  EMA_SAM_REPEAT=4 timeout 900 python3 "$root/tools/gpu_sam_rate.py" 25 200000 > "$out/sam_rate.txt" 2>&1; echo "sam: rc=$?"; tail -12 "$out/sam_rate.txt"
fi
if [[ $what == *cpuscale* ]]; then
  timeout 600 python3 "$root/tools/cpu_scaling.py" 3000 > "$out/cpu_scaling.txt" 2>&1; echo "cpuscale: rc=$?"; cat "$out/cpu_scaling.txt"
fi
if [[ $what == *prof* ]]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o runc -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"; echo "stats: rc=$?"
fi
if [[ $what == *pmc* ]]; then
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM" \
             "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc$i" -o runc -- python3 "$root/bench.py" --steps 2 --warmup 0 $B > "$out/bench_pmc$i.json" 2> "$out/bench_pmc$i.err"
    echo "pmc$i ($set): rc=$?"
  done
  python3 "$root/tools/pmc_summary.py" --csv "$out"/pmc* > "$out/pmc_summary.csv" 2> /dev/null
fi
if [[ $what == *rebench* ]]; then      # the bench line once more, now with K1's traffic and K2's instruction count from this call's PMC passes
  cp "$out/pmc_summary.csv" "$root/profiles/r06_pmc_grch38scale.csv"
  python3 "$root/tools/kernel_hash.py" > "$root/profiles/r06_pmc_grch38scale.csv.srchash"; cp "$root/profiles/r06_pmc_grch38scale.csv.srchash" "$out/"
  timeout 1500 python3 "$root/bench.py" > "$out/bench_with_pmc.json" 2> "$out/bench_with_pmc.err"; echo "rebench: rc=$?"; cut -c1-300 "$out/bench_with_pmc.json"
fi
# the per-dispatch traces are large; keep the statistics and the counter tables
find "$out" -name "*kernel_trace.csv" -size +8M -delete
find "$out" -name "*counter_collection.csv" -size +16M -delete
du -sh "$out"; ls "$out"
