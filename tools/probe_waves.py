#!/usr/bin/env python3
"""Per-wave records of k_trace_pool on the headline frame's primary rays (probe build `make -C optix_raytracer_amd/csrc probe-waves
PROBE_FLAGS=-DMIPT_PROBE_WAVES=1`, run with MIPT_LIBRARY=probe_waves.so): lifetime, bulk cycles, bulk steps, pool rounds of every
wave (= 64-ray tile), and when it started / ended.  Answers: what is a wave's life made of, and how well do the waves pack."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import _lib as L, pathtracer as PT, scenes

w, h = 1920, 1080
scene = scenes.random_soup(1_000_000)
st = PT.make_state(scene, w, h, 1, 0, max_depth=1)
ctx = st.context
launch = scenes.launch_desc(scene, w, h, 1)
eye, U, V, W = (np.asarray(launch[k], np.float32) for k in ("eye", "U", "V", "W"))
ty, tx = np.meshgrid(np.arange(h // 4), np.arange(w // 16), indexing="ij")
iy, ix = np.meshgrid(np.arange(4), np.arange(16), indexing="ij")
px = (tx[..., None, None] * 16 + ix).reshape(-1)
py = (ty[..., None, None] * 4 + iy).reshape(-1)
d = (2 * ((px + 0.5) / w) - 1)[:, None] * U + (2 * ((py + 0.5) / h) - 1)[:, None] * V + W
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((px.size, 8), np.float32)
rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = eye, 0.01, d, 1e16
n = rays.shape[0]
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
waves = (n + 63) // 64
rec = torch.zeros((2 * 40000, 4), dtype=torch.int32, device="cuda")
FRAME = "--frame" in sys.argv        # the primary launch of whole frames (with the cost-ordered tiles of MIPT_OPT_TILE_ORDER) instead
if FRAME:
    ob = PT.OutputBuffer(w, h, st.device)
    for _ in range(4):
        PT.launchSubframe(ob, st)
    os.environ["MIPT_PROBE_WAVES"] = str(rec.data_ptr())
    rec_any = torch.zeros((2 * 40000, 4), dtype=torch.int32, device="cuda")
    os.environ["MIPT_PROBE_WAVES_ANY"] = str(rec_any.data_ptr())
    ctx.set_option(L.OPT_PROFILE, 2)
    PT.launchSubframe(ob, st)
    print("whole frame, MIPT_OPT_TILE_ORDER = %d" % ctx.get_option(L.OPT_TILE_ORDER))
else:
    for _ in range(3):
        ctx.trace_closest(st.gas_handle, d_rays.data_ptr(), n, d_hits.data_ptr())
    ctx.sync()
    os.environ["MIPT_PROBE_WAVES"] = str(rec.data_ptr())
    ctx.set_option(L.OPT_PROFILE, 2)
    ctx.trace_closest(st.gas_handle, d_rays.data_ptr(), n, d_hits.data_ptr())
tm = ctx.launch_timings()
r = rec.cpu().numpy().view(np.uint32)
info = ctx.accel_info(st.gas_handle)
g = 32768 if waves > 32768 else ((waves + 7) // 8) * 8
life, bulk, steps, rounds = (r[:waves, k].astype(np.float64) for k in range(4))
t0 = r[g:g + waves, 0].astype(np.uint64) | (r[g:g + waves, 1].astype(np.uint64) << np.uint64(32))
t1 = r[g:g + waves, 2].astype(np.uint64) | (r[g:g + waves, 3].astype(np.uint64) << np.uint64(32))
ok = life > 0
# (the absolute stamps are per-XCD counters that are not synchronised with each other: only durations inside one wave are used;
#  a tick is a shader cycle)
kernel_cycles = tm["ms_trace_closest"] * 1e-3 * 2.4e9
print("node format %d (%d mirror nodes), kernel %.4f ms = %.0f k cycles at 2.4 GHz; %d waves" % (
    info["node_format"], info["mirror_nodes"], tm["ms_trace_closest"], kernel_cycles / 1e3, ok.sum()))
print("wave lifetime [k cycles]: mean %.1f, median %.1f, p90 %.1f, p99 %.1f, max %.1f; sum of lifetimes / (kernel x 256 CUs) = %.1f waves resident per CU on average" % (
    life[ok].mean() / 1e3, np.median(life[ok]) / 1e3, np.quantile(life[ok], 0.9) / 1e3, np.quantile(life[ok], 0.99) / 1e3, life[ok].max() / 1e3,
    life[ok].sum() / kernel_cycles / 256))
print("bulk share of the lifetime %.2f; bulk steps per wave mean %.1f max %d; pool rounds mean %.1f max %d" % (
    bulk[ok].sum() / life[ok].sum(), steps[ok].mean(), steps[ok].max(), rounds[ok].mean(), rounds[ok].max()))
cost = steps[ok] + rounds[ok]
print("tile cost (bulk steps + pool rounds): mean %.1f, median %.0f, p75 %.0f, p90 %.0f, p99 %.0f; share of the tiles per class of 6 units, heaviest (>= 90) first: %s" % (
    cost.mean(), np.median(cost), np.quantile(cost, 0.75), np.quantile(cost, 0.9), np.quantile(cost, 0.99),
    " ".join("%.3f" % ((np.minimum(cost // 6, 15) == k).mean()) for k in range(15, -1, -1))))
A = np.stack([np.ones(ok.sum()), steps[ok], rounds[ok]], 1)
coef, *_ = np.linalg.lstsq(A, life[ok], rcond=None)
print("least squares: lifetime = %.0f cycles + %.0f per bulk step + %.0f per pool round" % (coef[0], coef[1], coef[2]))
busy = steps[ok] > 0
print("waves with work: %d; cycles per bulk step (bulk cycles / steps) mean %.0f; cycles per pool round mean %.0f" % (
    busy.sum(), (bulk[ok][busy] / steps[ok][busy]).mean(), ((life[ok] - bulk[ok])[rounds[ok] > 0] / rounds[ok][rounds[ok] > 0]).mean()))
# ---- the timeline, from the device-wide 100 MHz counter (s_memrealtime)
s0 = float(t0[ok].min())
span = float(t1[ok].max()) - s0
print("first start -> last end: %.1f us" % (span / 100.0))
edges = np.linspace(0.0, span, 21)
a0, a1 = t0[ok].astype(np.float64) - s0, t1[ok].astype(np.float64) - s0
occ = [(np.clip(np.minimum(a1, b) - np.maximum(a0, a), 0, None).sum() / (b - a)) / 256.0 for a, b in zip(edges[:-1], edges[1:])]
print("waves resident per CU in 20 time slices:", " ".join("%.1f" % x for x in occ))
started = [((a0 >= a) & (a0 < b)).sum() for a, b in zip(edges[:-1], edges[1:])]
print("waves started per slice:", " ".join("%d" % x for x in started))
heavy = life[ok] > np.quantile(life[ok], 0.99)
print("the heaviest 1 %% of the waves: start at %.2f +- %.2f of the span, end at %.2f +- %.2f; the last 5 %% of the span holds the ends of %d waves, their mean lifetime %.0f k cycles" % (
    (a0[heavy] / span).mean(), (a0[heavy] / span).std(), (a1[heavy] / span).mean(), (a1[heavy] / span).std(), (a1 > 0.95 * span).sum(), life[ok][a1 > 0.95 * span].mean() / 1e3))
if FRAME:
    # ---- the any-hit kernel (k_trace_static<2>): a record per wave = chunk of rays
    q = rec_any.cpu().numpy().view(np.uint32)
    lifeA, raysA, entA = (q[:40000, k].astype(np.float64) for k in range(3))
    rndA, deadA = (q[:40000, 3] & 0xffff).astype(np.float64), (q[:40000, 3] >> 16).astype(np.float64)
    okA = lifeA > 0
    b0 = q[40000:, 0].astype(np.uint64) | (q[40000:, 1].astype(np.uint64) << np.uint64(32))
    b1 = q[40000:, 2].astype(np.uint64) | (q[40000:, 3].astype(np.uint64) << np.uint64(32))
    kc = tm["ms_trace_any"] * 1e-3 * 2.4e9
    print("\nany-hit kernel %.4f ms = %.0f k cycles; %d waves with %d rays" % (tm["ms_trace_any"], kc / 1e3, okA.sum(), raysA[okA].sum()))
    print("wave lifetime [k cycles]: mean %.1f, median %.1f, p90 %.1f, p99 %.1f, max %.1f; %.1f waves resident per CU on average" % (
        lifeA[okA].mean() / 1e3, np.median(lifeA[okA]) / 1e3, np.quantile(lifeA[okA], 0.9) / 1e3, np.quantile(lifeA[okA], 0.99) / 1e3,
        lifeA[okA].max() / 1e3, lifeA[okA].sum() / kc / 256))
    print("pool rounds per wave: mean %.1f, p90 %.0f, max %d; entries per round %.1f of 64; entries per ray %.1f; cycles per round %.0f" % (
        rndA[okA].mean(), np.quantile(rndA[okA], 0.9), rndA[okA].max(), entA[okA].sum() / rndA[okA].sum(), entA[okA].sum() / raysA[okA].sum(),
        lifeA[okA].sum() / rndA[okA].sum()))
    print("popped entries whose ray was already found occluded: %.1f %% of all popped entries" % (100.0 * deadA[okA].sum() / entA[okA].sum()))
    s0 = float(b0[okA].min()); spanA = float(b1[okA].max()) - s0
    c0, c1 = b0[okA].astype(np.float64) - s0, b1[okA].astype(np.float64) - s0
    edges = np.linspace(0.0, spanA, 21)
    print("first start -> last end: %.1f us" % (spanA / 100.0))
    print("waves resident per CU in 20 time slices:", " ".join("%.1f" % ((np.clip(np.minimum(c1, b) - np.maximum(c0, a), 0, None).sum() / (b - a)) / 256.0) for a, b in zip(edges[:-1], edges[1:])))
    print("waves started per slice:", " ".join("%d" % ((c0 >= a) & (c0 < b)).sum() for a, b in zip(edges[:-1], edges[1:])))
    sys.exit(0)                      # the slot -> tile map of an ordered launch is not the static one below
tile_of_block = np.arange(waves)
k, xcd = tile_of_block // 8, tile_of_block % 8
tile = (k // 256 * 8 + xcd) * 256 + k % 256
rows = tile // (w // 16)
print("mean lifetime [k cycles] by image band (tile rows 0..269 in 9 bands):", " ".join("%.0f" % (life[ok][(rows[ok] >= b * 30) & (rows[ok] < b * 30 + 30)].mean() / 1e3) for b in range(9)))
