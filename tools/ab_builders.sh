#!/bin/bash
# The three builders (MIPT_BUILD = 0 host SAH / 1 device LBVH / 2 device SAH) on the scenes whose top levels need SAH (run ON the GPU box):
#   bash tools/ab_builders.sh <out-file-under-gpurun_out>
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
: > "$OUT"
for b in 0 1 2; do
  echo "== MIPT_BUILD=$b" | tee -a "$OUT"
  MIPT_BUILD=$b python "$GRAFT_REPO_ROOT/tools/bench_configs.py" C1 C2\ Cornell\ 1920x1080\ spp16\ \(RR\ depth\) C3 C5 S3 S2 2>/dev/null | tee -a "$OUT"
  MIPT_BUILD=$b python "$GRAFT_REPO_ROOT/tools/bench_pipelines.py" 2>/dev/null | grep -v "two-level\|flattened" | tee -a "$OUT"
done
