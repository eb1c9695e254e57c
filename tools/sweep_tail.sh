#!/bin/bash
# MIPT_TAIL_PATHS sweep on the long passes (run ON the GPU box): bash tools/sweep_tail.sh <out-file-under-gpurun_out>
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
: > "$OUT"
for v in 262144 524288 1048576 2097152; do
  echo "== MIPT_TAIL_PATHS=$v" | tee -a "$OUT"
  MIPT_TAIL_PATHS=$v python "$GRAFT_REPO_ROOT/tools/bench_configs.py" C2\ Cornell\ 1920x1080\ spp16\ \(RR\ depth\) C4 S1 S4 S6 S2 S3 2>/dev/null | tee -a "$OUT"
done
