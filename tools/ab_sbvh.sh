#!/bin/bash
# builders 3 / 4 (host / device SAH with spatial splits) against builder 2 on the other configurations and pipelines (run ON the GPU box)
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
: > "$OUT"
for b in ${BUILDERS:-2 3 4}; do
  echo "== MIPT_BUILD=$b" | tee -a "$OUT"
  MIPT_BUILD=$b python "$GRAFT_REPO_ROOT/tools/bench_configs.py" C1 C2\ Cornell\ 1920x1080\ spp16\ \(RR\ depth\) C3 C4 C5 S1 S3 S2 S8 2>/dev/null | tee -a "$OUT"
  MIPT_BUILD=$b python "$GRAFT_REPO_ROOT/tools/bench_pipelines.py" 2>/dev/null | grep -v "two-level\|flattened" | tee -a "$OUT"
done
