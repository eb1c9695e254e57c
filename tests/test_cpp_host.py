"""The C++ host sample (host/mipt_pathtracer.cpp), the counterpart of SDK/optixPathTracer's main() in
file-output mode: same flags, same state functions, PPM written bottom row last (sutil::saveImage)."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "mipt_pathtracer")


def _build():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)


def test_no_gpu_is_reported_like_the_reference_reports_errors(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _build()
    r = subprocess.run([EXE, "--file", str(tmp_path / "x.ppm"), "--dim=16x16"], capture_output=True, text=True)
    assert r.returncode == 1 and "Caught exception" in r.stderr and "MIPT_ERROR_NO_DEVICE" in r.stderr
    r = subprocess.run([EXE, "--bogus"], capture_output=True, text=True)      # unknown flag -> usage, exit 0 (:971-975)
    assert r.returncode == 0 and "Unknown option" in r.stderr


def _read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"255"
        data = np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)
    return data


@pytest.mark.gpu
@pytest.mark.parametrize("args,scene_fn,w,h,spp,depth,subframes", [
    (["--scene", "cornell"], "cornell_box", 96, 64, 2, 0, 1),
    (["--scene", "cornell", "--subframes", "3"], "cornell_box", 64, 48, 1, 0, 3),
    (["--scene", "soup", "--tris", "50000", "--max-depth", "2"], "soup", 160, 90, 1, 2, 1),
])
def test_cpp_sample_writes_the_oracle_frame(tmp_path, args, scene_fn, w, h, spp, depth, subframes):
    from optix_raytracer_amd import scenes
    _build()
    out = tmp_path / "frame.ppm"
    r = subprocess.run([EXE, "--file", str(out), "--dim=%dx%d" % (w, h), "-s", str(spp)] + args,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    img = _read_ppm(out)
    scene = scenes.cornell_box() if scene_fn == "cornell_box" else scenes.random_soup(50000)
    osc = O.Scene(scene).build_bvh()          # the oracle's own BVH: results do not depend on the tree
    acc = None
    for sub in range(subframes):
        acc, frame, _ = osc.render(scenes.launch_desc(scene, w, h, spp, subframe=sub), max_depth=depth, accum=acc, threads=4)
    assert np.array_equal(img, frame[::-1, :, :3])       # file rows are top to bottom
