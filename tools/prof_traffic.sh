#!/bin/bash
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only) + a --stats trace of
# the headline bench.  Run ON the GPU box:  bash tools/prof_traffic.sh <tag>
set -u
TAG="$1"; shift
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $BENCH > "$OUT/stats.log" 2>&1 || tail -3 "$OUT/stats.log"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pass1" -- python $BENCH > "$OUT/pass1.log" 2>&1 || tail -3 "$OUT/pass1.log"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pass2" -- python $BENCH > "$OUT/pass2.log" 2>&1 || tail -3 "$OUT/pass2.log"
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_summary.txt"
cp "$OUT"/stats/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
grep -E "^==|FETCH|WRITE" "$OUT/pmc_summary.txt"
head -8 "$OUT/kernel_stats.csv"
grep '"metric"' "$OUT/stats.log" | tail -1 | cut -c1-400
