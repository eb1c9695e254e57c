#!/bin/bash
# PMC passes for the vector-L1 / texture-addresser question (run ON the GPU box through gpurun).
# Round 1 asked for whole TA_* / TD_* sets in one pass; rocprofv3 refused them with "error code 38: Request exceeds the
# capabilities of the hardware to collect" (more counters of one block than it has slots) and aborted before any kernel
# was dispatched.  Here every pass asks for at most TWO counters of a block.  --kernel-trace only (no --stats, no sys-trace).
# usage: tools/prof_l1.sh <out-subdir-under-gpurun_out> [extra bench.py args]
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
    || { echo "pass $i ($set) failed"; grep -m2 -i "error\|abort" "$OUT/pass$i.log"; }
  echo "pass $i done: $set"
done <<'SETS'
TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum
TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_COALESCED_READ_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_BUSY_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE
SETS
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -A60 "== k_trace_pool<0\|== k_trace_static<2" "$OUT/summary.txt" | grep -v "^==.*k_trace_stream" | head -150
