#!/bin/bash
# builder 3 (host SAH with spatial splits) against the default on the other configurations and pipelines (run ON the GPU box)
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
: > "$OUT"
for b in 2 3; do
  echo "== MIPT_BUILD=$b" | tee -a "$OUT"
  MIPT_BUILD=$b python "$GRAFT_REPO_ROOT/tools/bench_configs.py" C1 C2\ Cornell\ 1920x1080\ spp16\ \(RR\ depth\) C3 C4 C5 S1 S3 S2 S8 2>/dev/null | tee -a "$OUT"
  MIPT_BUILD=$b python "$GRAFT_REPO_ROOT/tools/bench_pipelines.py" 2>/dev/null | grep -v "two-level\|flattened" | tee -a "$OUT"
done
