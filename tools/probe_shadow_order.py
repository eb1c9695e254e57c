#!/usr/bin/env python3
"""Experiment: does the ORDER of the shadow-ray queue matter to the any-hit kernel?  Builds the headline frame's shadow rays
(oracle-free: primary hits from mipt_trace_closest, NEE rays towards the light's centre), then times mipt_trace_any on them in
(a) path order (what k_shade emits), (b) origin-Morton order, (c) random order, (d) sorted by (direction octant, Morton)."""
import os
import sys
import time

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
# primary rays in the kernel's path order: 16x4-pixel tiles
ty, tx = np.meshgrid(np.arange(h // 4), np.arange(w // 16), indexing="ij")
iy, ix = np.meshgrid(np.arange(4), np.arange(16), indexing="ij")
px = (tx[..., None, None] * 16 + ix).reshape(-1)
py = (ty[..., None, None] * 4 + iy).reshape(-1)
d = (2 * ((px + 0.5) / w) - 1)[:, None] * U + (2 * ((py + 0.5) / h) - 1)[:, None] * V + W
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((px.size, 8), np.float32)
rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = eye, 0.01, d, 1e16
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.zeros((rays.shape[0], 4), dtype=torch.float32, device="cuda")
ctx.trace_closest(st.gas_handle, d_rays.data_ptr(), rays.shape[0], d_hits.data_ptr())
ctx.sync()
hits = d_hits.cpu().numpy()
hit = hits[:, 0] > 0
P = rays[hit, :3] + hits[hit, 0:1] * rays[hit, 4:7]
lt = scene["light"]
rng = np.random.default_rng(1)
lp = np.asarray(lt["corner"]) + rng.random((P.shape[0], 1), np.float32) * np.asarray(lt["v1"]) + rng.random((P.shape[0], 1), np.float32) * np.asarray(lt["v2"])
toL = (lp - P).astype(np.float32)
dist = np.linalg.norm(toL, axis=1)
sh = np.zeros((P.shape[0], 8), np.float32)
sh[:, :3], sh[:, 3], sh[:, 4:7], sh[:, 7] = P, 0.01, toL / dist[:, None], dist - 0.01
sh = sh[toL[:, 1] > 0][:420_000]                       # the light is above: roughly the rays NEE would cast
n = sh.shape[0]


def morton(p):
    q = np.clip(((p + 0.25) / 1.5 * 1024).astype(np.int64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


orders = {"path order": np.arange(n), "origin Morton": np.argsort(morton(sh[:, :3]), kind="stable"), "random": rng.permutation(n),
          "tile-local Morton (512 rays)": np.concatenate([i + np.argsort(morton(sh[i:i + 512, :3]), kind="stable") for i in range(0, n, 512)]),
          "end-point Morton": np.argsort(morton(sh[:, :3] + sh[:, 4:7] * sh[:, 7:8] * 0.5), kind="stable")}
d_occ = torch.zeros(n, dtype=torch.int32, device="cuda")
ref = None
for name, o in orders.items():
    dr = torch.from_numpy(np.ascontiguousarray(sh[o])).cuda()
    for _ in range(3):
        ctx.trace_any(st.gas_handle, dr.data_ptr(), n, d_occ.data_ptr())
    ctx.sync()
    ctx.set_option(L.OPT_PROFILE, 2)
    for _ in range(20):
        ctx.trace_any(st.gas_handle, dr.data_ptr(), n, d_occ.data_ptr())
    tm = ctx.launch_timings()
    ctx.set_option(L.OPT_PROFILE, 0)
    occ = d_occ.cpu().numpy()
    back = np.empty_like(occ); back[o] = occ
    if ref is None:
        ref = back
    print("%-30s %d rays  %.4f ms  occluded %.3f  same flags %s" % (name, n, tm["ms_trace_any"] / 20, occ.mean(), bool(np.array_equal(back, ref))), flush=True)

# ---- what would the IDEAL order be worth?  (probe build: make -C optix_raytracer_amd/csrc probe-waves PROBE_FLAGS=-DMIPT_PROBE_WAVES=1,
# MIPT_LIBRARY=.../probe_waves.so)  Pool rounds of every 32-ray chunk in path order, then the chunks heaviest first.
if "probe_waves" in os.environ.get("MIPT_LIBRARY", ""):
    rec = torch.zeros((2 * 40000, 4), dtype=torch.int32, device="cuda")
    dr = torch.from_numpy(np.ascontiguousarray(sh)).cuda()
    os.environ["MIPT_PROBE_WAVES_ANY"] = str(rec.data_ptr())
    ctx.trace_any(st.gas_handle, dr.data_ptr(), n, d_occ.data_ptr())
    ctx.sync()
    del os.environ["MIPT_PROBE_WAVES_ANY"]
    q = rec.cpu().numpy().view(np.uint32)
    nch = (n + 31) // 32
    life, nr, ent, rnd = (q[:nch, k].astype(np.float64) for k in range(4))
    occ0 = d_occ.cpu().numpy()
    occ_frac = np.add.reduceat(occ0.astype(np.float64), np.arange(0, n, 32)) / np.maximum(nr, 1)
    print("chunks %d; rounds per chunk mean %.1f p90 %.0f max %d; corr(rounds, unoccluded fraction) = %.2f; corr(rounds, lifetime) = %.2f" % (
        nch, rnd.mean(), np.quantile(rnd, 0.9), rnd.max(), np.corrcoef(rnd, 1 - occ_frac)[0, 1], np.corrcoef(rnd, life)[0, 1]))
    lag = np.corrcoef(rnd[:-1], rnd[1:])[0, 1]
    print("corr(rounds of chunk i, chunk i+1) = %.2f, (i, i+4) = %.2f, (i, i+30: one tile row of 1920 px further) = %.2f" % (
        lag, np.corrcoef(rnd[:-4], rnd[4:])[0, 1], np.corrcoef(rnd[:-30], rnd[30:])[0, 1]))
    def chunk_order(key):
        o = np.argsort(-key, kind="stable")
        return np.concatenate([np.arange(c * 32, min(c * 32 + 32, n)) for c in o])
    cls16 = np.minimum(rnd // 4, 15)
    trial = {"chunks by rounds, heaviest first": chunk_order(rnd), "chunks by 16 classes of 4 rounds, stable": chunk_order(cls16),
             "chunks by unoccluded fraction": chunk_order(1 - occ_frac), "chunks lightest first": chunk_order(-rnd)}
    for name, o in trial.items():
        dr = torch.from_numpy(np.ascontiguousarray(sh[o])).cuda()
        for _ in range(3):
            ctx.trace_any(st.gas_handle, dr.data_ptr(), n, d_occ.data_ptr())
        ctx.sync()
        ctx.set_option(L.OPT_PROFILE, 2)
        for _ in range(20):
            ctx.trace_any(st.gas_handle, dr.data_ptr(), n, d_occ.data_ptr())
        tm = ctx.launch_timings()
        ctx.set_option(L.OPT_PROFILE, 0)
        print("%-45s %.4f ms" % (name, tm["ms_trace_any"] / 20), flush=True)
    sys.exit(0)

# ---- occluder sharing: how often does the triangle that occludes ray i-k also occlude ray i?  (closest occluders from
# mipt_trace_closest on the shadow segments, path order; the any-hit kernel would find SOME occluder, this is an estimate)
d_sh = torch.from_numpy(np.ascontiguousarray(sh)).cuda()
d_h = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
ctx.trace_closest(st.gas_handle, d_sh.data_ptr(), n, d_h.data_ptr())
ctx.sync()
hp = d_h.cpu().numpy()
prim = hp[:, 1].view(np.uint32)
V = np.asarray(scene["vertices"], np.float32).reshape(-1, 3, 4)[:, :, :3]


def hits(ray_idx, tri_idx):
    o, dd, tmin, tmax = sh[ray_idx, :3], sh[ray_idx, 4:7], sh[ray_idx, 3], sh[ray_idx, 7]
    v0 = V[tri_idx, 0]; e1 = V[tri_idx, 1] - v0; e2 = V[tri_idx, 2] - v0
    p = np.cross(dd, e2); det = (e1 * p).sum(1)
    ok = det != 0
    inv = 1.0 / np.where(ok, det, 1)
    tv = o - v0
    u = (tv * p).sum(1) * inv
    q = np.cross(tv, e1)
    v = (dd * q).sum(1) * inv
    t = (e2 * q).sum(1) * inv
    return ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax)


occ = prim != 0xffffffff
print("occluded (closest-hit form): %.3f" % occ.mean())
idx = np.arange(n)
for k in (1, 2, 4, 8, 16, 32):
    src = idx - k
    valid = (src >= 0) & occ & (prim[np.maximum(src, 0)] != 0xffffffff) & ((idx // 32) == (np.maximum(src, 0) // 32))
    h = hits(idx[valid], prim[src[valid]])
    print("lag %2d (same 32-ray chunk): candidate available for %.3f of the rays, of those %.3f are occluded by it" % (k, valid.mean(), h.mean()))
# best case inside a 32-ray chunk: any earlier ray's occluder
any_hit = np.zeros(n, bool)
for k in range(1, 32):
    src = idx - k
    valid = (src >= 0) & occ & (prim[np.maximum(src, 0)] != 0xffffffff) & ((idx // 32) == (np.maximum(src, 0) // 32)) & ~any_hit
    vi = idx[valid]
    any_hit[vi] |= hits(vi, prim[src[valid]])
print("occluded by the occluder of ANY earlier ray of its 32-ray chunk: %.3f of all rays" % any_hit.mean())
