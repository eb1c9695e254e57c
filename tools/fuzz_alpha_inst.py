#!/usr/bin/env python3
"""Differential fuzz of the any-hit kernels over TWO-LEVEL structures: instanced scenes of random seeds and sizes with BLEND / MASK
materials, every frame against the oracle's two-level render bit for bit (the test of tests/test_instances.py, other seeds).
round 3: seeds 40-51 (30-166 instances, pool and one-ray-per-thread forms), no failure; round 4: every other seed with the meshes built by
builder 4 (spatial splits: a candidate met through two references counts once), no failure."""
import os, sys, pathlib, tempfile
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import test_instances as TI
bad = 0
for seed in range(40, 52):
    n = 30 + 17 * (seed % 9)
    builder = 4 if seed % 2 else None
    mp = pytest.MonkeyPatch()
    try:
        TI.test_any_hit_programs_inside_instances_match_the_oracle(pathlib.Path(tempfile.mkdtemp()), mp, None if seed % 4 else 2, seed, n, builder)
        print("seed %d, %d instances%s ok" % (seed, n, ", builder 4" if builder else ""), flush=True)
    except AssertionError as e:
        bad += 1
        print("seed %d, %d instances FAILED: %s" % (seed, n, str(e)[:200]), flush=True)
    finally:
        mp.undo()
print("failures:", bad)
