#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of a few headline frames -> the launches of three consecutive frames with their gaps.
usage (on the GPU box): rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/frame_timeline.py run ; python tools/frame_timeline.py show <dir>"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import torch
    from optix_raytracer_amd import pathtracer as PT, scenes
    scene = scenes.random_soup(1_000_000)
    st = PT.make_state(scene, 1920, 1080, 1, 0, max_depth=1)
    ob = PT.OutputBuffer(1920, 1080, st.device)
    st.params.frame_buffer = ob.map()
    for k in range(40):
        st.params.subframe_index = k
        st.context.launch(st.params, 1920, 1080, 1, stream=st.stream.cuda_stream)
    torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "mipt::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-9:]                                   # the last three frames: three launches each
    t0, prev_end = int(rows[0]["Start_Timestamp"]), None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("mipt::", "").replace("void ", "").split("(")[0][:34]
        print("%-34s start %8.1f us  duration %6.1f us  gap before %5.1f us" % (n, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3))
        prev_end = e
    print("three frames: %.1f us from the first launch's start to the last launch's end" % ((prev_end - t0) / 1e3))
