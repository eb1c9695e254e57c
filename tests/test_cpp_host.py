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


MV_EXE = os.path.join(ROOT, "host", "mipt_meshviewer")
MG_EXE = os.path.join(ROOT, "host", "mipt_multigpu")


def _build_all():
    _build()


@pytest.mark.gpu
def test_zero_copy_output_buffer_gives_the_same_file(tmp_path):
    """sutil::HIPOutputBuffer<uchar4> in ZERO_COPY mode (CUDAOutputBuffer.h:203-216: pinned, mapped host memory the launch writes
    directly) against the device-memory mode"""
    _build_all()
    outs = []
    for extra in ([], ["--zero-copy"]):
        out = tmp_path / ("f%d.ppm" % len(outs))
        r = subprocess.run([EXE, "--file", str(out), "--dim=120x80", "-s", "2", "--scene", "cornell", "--subframes", "2"] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(_read_ppm(out))
    assert np.array_equal(outs[0], outs[1]) and outs[0].max() > 100


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["--two-level", "--flattened", "auto"])
def test_cpp_meshviewer_writes_the_frame_of_the_python_mirror(tmp_path, mode):
    """host/mipt_meshviewer.cpp (= optixMeshViewer's main() on sutil::Scene / loadScene of include/mipt_sutil.hpp) against
    meshviewer.render, which the oracle checks (tests/test_whitted.py, tests/test_instances.py)"""
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import scenes
    _build_all()
    path = scenes.write_instanced_gltf(tmp_path / "instanced.gltf", n_instances=40, seed=2)
    out = tmp_path / "mv.ppm"
    w, h = 200, 120
    # ("auto": no flag -- sutil::Scene's AccelMode::AUTO flattens a scene of several instances and a few thousand triangles)
    r = subprocess.run([MV_EXE, "--model", str(path), "--file", str(out), "--dim=%dx%d" % (w, h), "--subframes", "2", "--zero-copy"] + ([mode] if mode != "auto" else []),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert ("two-level structure" if mode == "--two-level" else "flattened structure") in r.stderr
    img = _read_ppm(out)
    st, frame = MV.render(path, w, h, subframes=2, two_level=(mode == "--two-level"))
    MV.cleanup(st)
    assert np.array_equal(img, frame[::-1, :, :3])
    r = subprocess.run([MV_EXE, "--model", str(tmp_path / "missing.gltf"), "--file", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "Caught exception" in r.stderr and "loadScene" in r.stderr


@pytest.mark.gpu
def test_cpp_multigpu_sample_writes_the_oracle_frame(tmp_path):
    """host/mipt_multigpu.cpp (= optixMultiGPU's main(): one host thread, a state per device, mipt_group_launch / allgather /
    sync) on the devices of this box; the assembled frame equals the oracle's single-process frame"""
    from optix_raytracer_amd import scenes
    _build_all()
    out = tmp_path / "mg.ppm"
    w, h, spp = 128, 96, 2
    r = subprocess.run([MG_EXE, "--file", str(out), "--dim=%dx%d" % (w, h), "-s", str(spp), "--subframes", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "device(s), exchange through" in r.stderr
    img = _read_ppm(out)
    scene = scenes.cornell_box()
    osc = O.Scene(scene).build_bvh()
    acc = None
    for sub in range(2):
        acc, frame, _ = osc.render(scenes.launch_desc(scene, w, h, spp, subframe=sub), accum=acc, threads=4)
    assert np.array_equal(img, frame[::-1, :, :3])
