#!/bin/bash
# PMC passes over the headline bench (run ON the GPU box through gpurun).  Counters are collected in
# their own runs with --kernel-trace only (no --stats / sys-trace: MI355X_MICROARCH.md, gpurun rules).
# (TA_* / TD_* counter sets abort rocprofv3 on this pool -- signal 6 -- and are left out.)
# usage: tools/prof_pmc.sh <out-subdir-under-gpurun_out> [extra bench.py args]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- python $BENCH > "$OUT/pass$i.log" 2>&1 \
    || { echo "pass $i ($set) failed"; tail -3 "$OUT/pass$i.log"; }
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_VMEM_TA_ADDR_FIFO_FULL
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
GRBM_GUI_ACTIVE
SETS
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
