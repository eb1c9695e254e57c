#!/bin/bash
# A/B of library builds on the headline bench (run ON the GPU box): bash tools/ab.sh <out-file-under-gpurun_out> lib1.so lib2.so ...
# each lib (under optix_raytracer_amd/lib/) is run `REPS` times (default 2), interleaved, so box drift shows
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
: > "$OUT"
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for lib in "$@"; do
  [ -f "$GRAFT_REPO_ROOT/optix_raytracer_amd/lib/$lib" ] || { echo "$lib missing" | tee -a "$OUT"; continue; }
  line=$(MIPT_LIBRARY=$lib python "$GRAFT_REPO_ROOT/bench.py" --steps 60 --warmup 10 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']['per_frame_ms']
print('ms/frame %.4f closest %.4f any %.4f shade %.4f first %.4f' % (d['ms_per_step'], r['trace_closest'], r['trace_any'], r['shade'], d['config'].get('first_frame_ms', 0)))")
  echo "$lib  $line" | tee -a "$OUT"
done
done
