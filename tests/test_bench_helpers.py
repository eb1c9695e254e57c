"""CPU-only checks of bench.py's bookkeeping: the committed PMC profile is accepted only for the kernel source it was measured on, and
the ceilings are throughput ceilings (the data-return unit's stall cycles are not counted as work)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_source_id_covers_the_kernels_and_their_headers():
    """bench.kernels_blob_sha(): sha1 over git's blob ids of kernels.hip and every header it includes -- a change in pool_trace.h is a
    change of the kernels just as one in kernels.hip"""
    import hashlib
    import re
    import subprocess
    b = _bench()
    csrc = os.path.join(ROOT, "optix_raytracer_amd", "csrc")
    h = hashlib.sha1()
    for name in b.KERNEL_SOURCES:
        data = open(os.path.join(csrc, name), "rb").read()
        blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
        if os.path.isdir(os.path.join(ROOT, ".git")):
            assert blob == subprocess.run(["git", "-C", ROOT, "hash-object", os.path.join(csrc, name)], capture_output=True, text=True).stdout.strip()
        h.update(blob.encode())
    assert b.kernels_blob_sha() == h.hexdigest()
    # every local header kernels.hip pulls in (transitively) is on the list
    seen, todo = set(), ["kernels.hip"]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'#include "([^"/]+)"', open(os.path.join(csrc, f)).read()):
            todo.append(inc)
    assert seen <= set(b.KERNEL_SOURCES), seen - set(b.KERNEL_SOURCES)


def test_profile_of_other_kernel_source_is_refused(tmp_path, monkeypatch):
    b = _bench()
    prof = {"commit": "x", "source": "t", "kernels_hip_blob": "0" * 40, "kernels": {"k_trace_pool<0, 24, 1, true>": {"FETCH_SIZE": 1.0}}}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "pmc_kernels.json").write_text(json.dumps(prof))
    (tmp_path / "optix_raytracer_amd" / "csrc").mkdir(parents=True)
    for name in b.KERNEL_SOURCES:
        (tmp_path / "optix_raytracer_amd" / "csrc" / name).write_bytes(b"// other source\n")
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    c, src = b.pmc_counters("k_trace_pool<0, 24, 1, true>")
    assert c is None and "refused" in src and "other kernel sources" in src["refused"]
    prof["kernels_hip_blob"] = b.kernels_blob_sha()
    (tmp_path / "profiles" / "pmc_kernels.json").write_text(json.dumps(prof))
    c, src = b.pmc_counters("k_trace_pool<0, 24, 1, true>")
    assert c == {"FETCH_SIZE": 1.0} and "refused" not in src
    c, src = b.pmc_counters("k_trace_pool<0, 24, 1>")                     # no matching by name stem any more
    assert c is None and "no kernel named" in src["refused"]


def test_ceilings_are_throughput_ceilings():
    b = _bench()
    t = 0.35e-3
    cyc = t * b.CLOCK_HZ * b.N_CUS
    counters = {"FETCH_SIZE": 90000.0, "WRITE_SIZE": 100000.0, "TD_TD_BUSY_sum": 0.9 * cyc, "TD_TC_STALL_sum": 0.5 * cyc, "TA_TA_BUSY_sum": 0.6 * cyc,
                "TA_BUFFER_READ_WAVEFRONTS_sum": 4.48e6, "SQ_THREAD_CYCLES_VALU": 0.53 * 64 * 1e6, "SQ_ACTIVE_INST_VALU": 1e6, "SQ_INSTS_VALU": 2.1e8}
    c = b.ceilings(counters, t, 5e9)
    assert abs(c["td"]["frac"] - 0.4) < 1e-3 and abs(c["td"]["busy_incl_stall"] - 0.9) < 1e-3 and abs(c["td"]["stall_share_of_busy"] - 0.5556) < 1e-3
    assert abs(c["hbm"]["frac"] - (2 * 90000 + 100000) * 1024 / t / 1e9 / 8000.0) < 1e-3
    assert 0.6 < c["vmem"]["frac"] < 0.8 and 0.9 < c["vmem"]["frac_if_every_lane_diverges"] < 1.1
    assert all(v["frac"] <= 1.05 for v in c.values())
