#!/bin/bash
# bench.py's headline frame under one environment knob at a time (run ON the GPU box): usage tools/sweep_env.sh VAR v1 v2 ...
# prints ms per frame and the per-kernel split for every value; the first line is the default build.
VAR=$1; shift
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|\"per_frame_ms\": {[^}]*}" | tr "\n" " "; echo; }
echo -n "default: "; one
for v in "$@"; do echo -n "$VAR=$v: "; env $VAR=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|\"per_frame_ms\": {[^}]*}" | tr "\n" " "; echo; done
