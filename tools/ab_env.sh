#!/bin/bash
# A/B of environment settings (and library builds) on the headline bench (run ON the GPU box):
#   bash tools/ab_env.sh <out-file-under-gpurun_out> "VAR=a VAR2=b" "VAR=c" ...      each setting runs REPS (default 2) times, interleaved
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
: > "$OUT"
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for setting in "$@"; do
  line=$(env $setting python "$GRAFT_REPO_ROOT/bench.py" --steps 60 --warmup 10 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']['per_frame_ms']
print('ms/frame %.4f closest %.4f any %.4f shade %.4f first %.4f build_s %.4f | %s' % (d['ms_per_step'], r['trace_closest'], r['trace_any'], r['shade'], d['config'].get('first_frame_ms', 0), d['config'].get('bvh_build_s', 0), ' '.join(__import__('re').findall(r'BVH2 \d+ nodes depth \d+|over \d+ triangle records|\d+ mirror nodes', d['config']['workload']))))")
  echo "$setting  $line" | tee -a "$OUT"
done
done
