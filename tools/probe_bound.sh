#!/bin/bash
# Which resource binds the traversal kernels?  Runs the headline bench on libmipt builds that carry n extra 16-byte loads
# (probe_vmem<n>.so) or n extra VALU instructions (probe_valu<n>.so) per traversal step (csrc/Makefile `probe-%`) and prints
# the per-kernel times: a resource with slack absorbs the extra instructions, the binding one charges them in full.
# Run ON the GPU box:  bash tools/probe_bound.sh <out-file-under-gpurun_out>
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
: > "$OUT"
for lib in libmipt.so probe_valu24.so probe_valu48.so probe_vmem1.so probe_vmem2.so libmipt.so; do
  [ -f "$GRAFT_REPO_ROOT/optix_raytracer_amd/lib/$lib" ] || continue
  line=$(MIPT_LIBRARY=$lib python "$GRAFT_REPO_ROOT/bench.py" --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']['per_frame_ms']
print('ms/frame %.4f closest %.4f any %.4f shade %.4f' % (d['ms_per_step'], r['trace_closest'], r['trace_any'], r['shade']))")
  echo "$lib  $line" | tee -a "$OUT"
done
