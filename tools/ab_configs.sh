#!/bin/bash
# A/B of library builds on the long-tail configurations and the shard probe (run ON the GPU box):
#   bash tools/ab_configs.sh <out-file-under-gpurun_out> lib1.so lib2.so ...     (CONFIGS="S1 S8 ..." picks the rows)
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
: > "$OUT"
for lib in "$@"; do
  echo "== $lib" | tee -a "$OUT"
  MIPT_LIBRARY=$lib python "$GRAFT_REPO_ROOT/tools/bench_configs.py" ${CONFIGS:-C1 C4 S1 S4 S5 S6 S8 S9 S10} 2>/dev/null | tee -a "$OUT"
  MIPT_LIBRARY=$lib PROBE_WORLDS=${PROBE_WORLDS:-4,8} python "$GRAFT_REPO_ROOT/tools/shard_probe.py" 2>/dev/null | tee -a "$OUT"
done
