#!/usr/bin/env python3
"""Differential fuzz of the Whitted any-hit kernels: random stacks of MASK / BLEND panes (tilted, coincident twins) over a tessellated
ground, node formats 0 / 1 / 2 in turn, every frame against the oracle bit for bit (the helper of tests/test_whitted.py).
round 3: seeds 10-33, no failure."""
import os, sys, pathlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_whitted as TW
import _gltf_fixture as FX
bad = 0
for seed in range(10, 34):
    fmt = seed % 3
    os.environ["MIPT_NODE_FORMAT"] = str(fmt)
    tmp = pathlib.Path(tempfile.mkdtemp())
    try:
        TW._alpha_frames_match_the_oracle(FX.write_alpha(tmp, stack=20 + seed % 17, tess=12 + seed % 11, jitter=seed), 112, 80, 2)
        print("seed %d format %d ok" % (seed, fmt), flush=True)
    except AssertionError as e:
        bad += 1
        print("seed %d format %d FAILED: %s" % (seed, fmt, str(e)[:200]), flush=True)
print("failures:", bad)
