#!/bin/bash
# one SQ-counter pass (fast) over the headline bench; usage: tools/prof_sq.sh <outdir> [bench args]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d "$OUT/pass1" -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $* > "$OUT/pass1.log" 2>&1 || tail -3 "$OUT/pass1.log"
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" | grep -A9 -E "k_trace_(stream|simple|persistent)<[02], false"
