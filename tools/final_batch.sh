#!/bin/bash
# The round's closing measurements in ONE gpurun call (run ON the GPU box): bash tools/final_batch.sh <tag>   -> files gpurun_out/<tag>_*
T="$1"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cd "$R"
python -m pytest tests -m gpu -q > "$O/${T}_tests.log" 2>&1; tail -3 "$O/${T}_tests.log"
python bench.py > "$O/${T}_bench_n1.json" 2> "$O/${T}_bench_n1.err"; cut -c1-300 "$O/${T}_bench_n1.json"
python tools/bench_configs.py 2>/dev/null > "$O/${T}_bench_configs.txt"; echo configs done
python tools/bench_pipelines.py 2>/dev/null > "$O/${T}_bench_pipelines.txt"; echo pipelines done
python tools/bench_instances.py 2>/dev/null > "$O/${T}_bench_instances.txt"; echo instances done
python tools/bench_inst_pt.py 2>/dev/null > "$O/${T}_bench_inst_pt.txt"; echo inst_pt done
python tools/shard_probe.py 2>/dev/null > "$O/${T}_shard_probe.txt"; echo shard done
python tools/build_probe.py 1000000 0:2,1:2,2:2,4:2 2>/dev/null > "$O/${T}_build_probe.txt"; echo build done
python tools/soak.py 100 > "$O/${T}_soak.txt" 2>&1; tail -2 "$O/${T}_soak.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${T}_kstats" -- python3 "$R/bench.py" --no-cpu-baseline > "$O/${T}_kstats.log" 2>&1; echo kstats done
cd "$R" && bash tools/prof_l1.sh "${T}_pmc" > "$O/${T}_pmc.log" 2>&1; tail -3 "$O/${T}_pmc.log"
