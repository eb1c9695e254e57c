"""Writes a small synthetic glTF 2.0 scene (and its .glb twin) that exercises what sutil::loadScene handles:
u16 / u32 / no indices, a strided vertex view, a data: URI buffer and an external .bin, nested nodes with
matrix + TRS, two instances of one mesh, a perspective camera node, a non-triangle primitive that must be
ignored, materials with and without factors.  Also returns the expected world-space triangles computed with
float32 numpy in the operation order of SDK/sutil/Matrix.h."""
import base64
import json
import struct

import numpy as np

f32 = np.float32


def m4_mul(a, b):
    out = np.zeros((4, 4), f32)
    for i in range(4):
        for j in range(4):
            s = f32(0)
            for k in range(4):
                s = f32(s + f32(a[i, k] * b[k, j]))
            out[i, j] = s
    return out


def m4_apply(a, v):
    out = np.zeros(4, f32)
    for i in range(4):
        s = f32(0)
        for j in range(4):
            s = f32(s + f32(a[i, j] * v[j]))
        out[i] = s
    return out


def trs(t=None, r=None, s=None, matrix=None):
    T = np.eye(4, dtype=f32)
    if t is not None:
        T[:3, 3] = np.asarray(t, f32)
    R = np.eye(4, dtype=f32)
    if r is not None:
        qx, qy, qz, qw = [f32(x) for x in r]
        two = f32(2)
        R[0, 0] = f32(f32(f32(1) - f32(f32(two * qy) * qy)) - f32(f32(two * qz) * qz)); R[0, 1] = f32(f32(f32(two * qx) * qy) - f32(f32(two * qz) * qw)); R[0, 2] = f32(f32(f32(two * qx) * qz) + f32(f32(two * qy) * qw))
        R[1, 0] = f32(f32(f32(two * qx) * qy) + f32(f32(two * qz) * qw)); R[1, 1] = f32(f32(f32(1) - f32(f32(two * qx) * qx)) - f32(f32(two * qz) * qz)); R[1, 2] = f32(f32(f32(two * qy) * qz) - f32(f32(two * qx) * qw))
        R[2, 0] = f32(f32(f32(two * qx) * qz) - f32(f32(two * qy) * qw)); R[2, 1] = f32(f32(f32(two * qy) * qz) + f32(f32(two * qx) * qw)); R[2, 2] = f32(f32(f32(1) - f32(f32(two * qx) * qx)) - f32(f32(two * qy) * qy))
    S = np.eye(4, dtype=f32)
    if s is not None:
        S[0, 0], S[1, 1], S[2, 2] = [f32(x) for x in s]
    M = np.eye(4, dtype=f32)
    if matrix is not None:
        M = np.asarray(matrix, f32).reshape(4, 4).T.copy()      # glTF stores column-major
    return M, T, R, S


def node_xform(parent, t=None, r=None, s=None, matrix=None):
    M, T, R, S = trs(t, r, s, matrix)
    return m4_mul(m4_mul(m4_mul(m4_mul(parent, M), T), R), S)


def write(tmpdir, n_grid=6):
    rng = np.random.default_rng(11)
    # mesh 0, primitive 0: a bumpy grid, u16 indices, positions interleaved with normals (stride 24)
    g = n_grid
    xs, zs = np.meshgrid(np.linspace(-1, 1, g), np.linspace(-1, 1, g))
    pos0 = np.stack([xs.ravel(), 0.15 * rng.standard_normal(g * g), zs.ravel()], 1).astype(f32)
    inter = np.zeros((g * g, 6), f32)
    inter[:, :3] = pos0
    nrm = np.random.default_rng(5).standard_normal((g * g, 3)) * 0.25 + (0, 1, 0)      # smooth-ish, not unit length
    inter[:, 3:] = nrm.astype(f32)
    idx0 = []
    for j in range(g - 1):
        for i in range(g - 1):
            a = j * g + i
            idx0 += [a, a + 1, a + g, a + 1, a + g + 1, a + g]
    idx0 = np.asarray(idx0, np.uint16)
    # mesh 0, primitive 1: 4 loose triangles, no indices
    pos1 = (rng.uniform(-1, 1, (12, 3)) * (1, 0.5, 1) + (0, 1.0, 0)).astype(f32)
    # mesh 1: a box-ish blob, u32 indices
    pos2 = rng.uniform(-0.4, 0.4, (20, 3)).astype(f32)
    idx2 = rng.integers(0, 20, (14, 3)).astype(np.uint32)
    idx2 = idx2[(idx2[:, 0] != idx2[:, 1]) & (idx2[:, 1] != idx2[:, 2]) & (idx2[:, 0] != idx2[:, 2])]
    # a line primitive that must be ignored
    idx_lines = np.asarray([0, 1, 1, 2], np.uint16)

    bin0 = inter.tobytes() + idx0.tobytes()
    bin0 += b"\0" * (-len(bin0) % 4)
    off_pos1 = len(bin0)
    bin0 += pos1.tobytes()
    off_lines = len(bin0)
    bin0 += idx_lines.tobytes()
    bin1 = pos2.tobytes() + idx2.tobytes()

    def mm(a):
        return [float(x) for x in a.min(0)], [float(x) for x in a.max(0)]

    root_matrix = [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 0.1, 0.2, -0.3, 1]        # column-major: scale 0.5 + translate
    q = np.asarray([0.1, 0.7, -0.2, 0.68], np.float64)
    q /= np.linalg.norm(q)
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(bin0), "uri": "fixture0.bin"},
                    {"byteLength": len(bin1), "uri": "data:application/octet-stream;base64," + base64.b64encode(bin1).decode()}],
        "bufferViews": [
            {"buffer": 0, "byteOffset": 0, "byteLength": inter.nbytes, "byteStride": 24},
            {"buffer": 0, "byteOffset": inter.nbytes, "byteLength": idx0.nbytes},
            {"buffer": 0, "byteOffset": off_pos1, "byteLength": pos1.nbytes},
            {"buffer": 1, "byteOffset": 0, "byteLength": pos2.nbytes},
            {"buffer": 1, "byteOffset": pos2.nbytes, "byteLength": idx2.nbytes},
            {"buffer": 0, "byteOffset": off_lines, "byteLength": idx_lines.nbytes},
        ],
        "accessors": [
            {"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": g * g, "type": "VEC3", "min": mm(pos0)[0], "max": mm(pos0)[1]},
            {"bufferView": 1, "componentType": 5123, "count": int(idx0.size), "type": "SCALAR"},
            {"bufferView": 2, "componentType": 5126, "count": 12, "type": "VEC3", "min": mm(pos1)[0], "max": mm(pos1)[1]},
            {"bufferView": 3, "componentType": 5126, "count": 20, "type": "VEC3", "min": mm(pos2)[0], "max": mm(pos2)[1]},
            {"bufferView": 4, "componentType": 5125, "count": int(idx2.size), "type": "SCALAR"},
            {"bufferView": 5, "componentType": 5123, "count": 4, "type": "SCALAR"},
            {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": g * g, "type": "VEC3"},
        ],
        "materials": [
            {"name": "red", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.1, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.5}},
            {"name": "defaults", "emissiveFactor": [0.0, 0.0, 0.0], "alphaMode": "MASK", "alphaCutoff": 0.25, "doubleSided": True},
        ],
        "meshes": [
            {"name": "ground", "primitives": [
                {"attributes": {"POSITION": 0, "NORMAL": 6}, "indices": 1, "material": 0, "mode": 4},
                {"attributes": {"POSITION": 2}, "material": 1},
                {"attributes": {"POSITION": 0}, "indices": 5, "mode": 1}]},
            {"name": "blob", "primitives": [{"attributes": {"POSITION": 3}, "indices": 4}]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7, "aspectRatio": 1.25, "znear": 0.1}}],
        "nodes": [
            {"children": [1, 3], "matrix": root_matrix},
            {"mesh": 0, "translation": [0.0, -0.5, 0.0], "scale": [2.0, 1.0, 2.0], "children": [2]},
            {"mesh": 1, "rotation": [float(x) for x in q], "translation": [0.3, 1.2, 0.1]},
            {"mesh": 1, "translation": [-1.0, 0.8, 0.5], "scale": [1.5, 1.5, 1.5]},
            {"camera": 0, "translation": [0.2, 1.5, 4.0]},
        ],
        "scenes": [{"nodes": [0, 4]}], "scene": 0,
    }
    gltf = tmpdir / "fixture.gltf"
    gltf.write_text(json.dumps(doc))
    (tmpdir / "fixture0.bin").write_bytes(bin0)
    # GLB twin: one embedded BIN chunk holding both buffers' bytes would change buffer indices, so the twin keeps
    # buffer 1 as a data URI and embeds buffer 0
    doc_glb = json.loads(json.dumps(doc))
    del doc_glb["buffers"][0]["uri"]
    js = json.dumps(doc_glb).encode()
    js += b" " * (-len(js) % 4)
    glb = tmpdir / "fixture.glb"
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bin0), 0x004E4942) + bin0
    glb.write_bytes(b"glTF" + struct.pack("<II", 2, 12 + len(body)) + body)

    # ---- expected flattening, instance order = node traversal order (roots in index order, depth first)
    I = np.eye(4, dtype=f32)
    x0 = node_xform(I, matrix=root_matrix)
    x1 = node_xform(x0, t=[0.0, -0.5, 0.0], s=[2.0, 1.0, 2.0])
    x2 = node_xform(x1, t=[0.3, 1.2, 0.1], r=[float(x) for x in q])
    x3 = node_xform(x0, t=[-1.0, 0.8, 0.5], s=[1.5, 1.5, 1.5])
    x4 = node_xform(I, t=[0.2, 1.5, 4.0])
    tris, mats = [], []

    def emit(xf, pos, idx, mat):
        ids = np.arange(pos.shape[0]) if idx is None else idx.reshape(-1)
        for v in ids[: (len(ids) // 3) * 3]:
            w = m4_apply(xf, np.array([pos[v, 0], pos[v, 1], pos[v, 2], 1.0], f32))
            tris.append((w[0], w[1], w[2], f32(0)))
        mats.extend([mat] * (len(ids) // 3))
    emit(x1, pos0, idx0, 0); emit(x1, pos1, None, 1)
    emit(x2, pos2, idx2, 0xFFFFFFFF)
    emit(x3, pos2, idx2, 0xFFFFFFFF)
    eye = m4_apply(x4, np.array([0, 0, 0, 1], f32))[:3]
    up = m4_apply(x4, np.array([0, 1, 0, 0], f32))[:3]
    # world-space normals of primitive 0 (the only one with a NORMAL accessor): M^-T n, float64 reference
    it = np.linalg.inv(x1.astype(np.float64)[:3, :3]).T
    n_grid_tris = idx0.size // 3
    normals = np.zeros((len(tris), 4), np.float64)
    normals[: n_grid_tris * 3, :3] = inter[idx0.reshape(-1), 3:].astype(np.float64) @ it.T
    normals[: n_grid_tris * 3, 3] = 1.0
    expect = {"normals": normals, "vertices": np.asarray(tris, f32), "mat_index": np.asarray(mats, np.uint32), "eye": eye, "up": up,
              "fovY": f32(f32(0.7) * f32(180.0) / f32(np.pi)), "aspect": f32(1.25), "n_instances": 3, "n_meshes": 2}
    return gltf, glb, expect


def write_textured(tmpdir, seed=3):
    """A small scene that exercises the textured Whitted path: TEXCOORD_0/1, COLOR_0, base-colour / metallic-roughness /
    normal / emissive textures, KHR_texture_transform, wrap / clamp / mirror / nearest samplers, an 8-bit RGBA, a palette
    and a 16-bit PNG.  Returns the .gltf path."""
    import _png_fixture as PF
    rng = np.random.default_rng(seed)
    # ---- images
    yy, xx = np.mgrid[0:16, 0:16]
    checker = np.zeros((16, 16, 4), int)
    checker[..., 0] = np.where((xx // 4 + yy // 4) % 2, 230, 40)
    checker[..., 1] = xx * 16
    checker[..., 2] = yy * 16
    checker[..., 3] = 255
    png_rgba, _ = PF.encode(checker, 6, 8)
    pal = rng.integers(30, 256, (8, 3))
    png_pal, _ = PF.encode(rng.integers(0, 8, (5, 7, 1)), 3, 4, palette=pal)
    nm = np.zeros((8, 8, 3), int)
    nm[..., 0] = 32768 + (20000 * np.sin(xx[:8, :8] * 0.9)).astype(int)
    nm[..., 1] = 32768 + (20000 * np.cos(yy[:8, :8] * 0.7)).astype(int)
    nm[..., 2] = 60000
    png_nm16, _ = PF.encode(nm, 2, 16)
    mr = np.zeros((4, 4, 3), int)
    mr[..., 1] = rng.integers(60, 256, (4, 4))
    mr[..., 2] = rng.integers(0, 256, (4, 4))
    png_mr, _ = PF.encode(mr, 2, 8)
    (tmpdir / "nm16.png").write_bytes(png_nm16)
    uri = lambda b: "data:image/png;base64," + base64.b64encode(b).decode()          # noqa: E731
    # ---- geometry: a ground grid (uv0 + uv1), a tilted quad with vertex colours, a small emissive quad
    g = 5
    gx, gz = np.meshgrid(np.linspace(-2, 2, g), np.linspace(-2, 2, g))
    pos0 = np.stack([gx.ravel(), 0.1 * np.sin(gx.ravel() * 2) * np.cos(gz.ravel()), gz.ravel()], 1).astype(f32)
    uv0 = np.stack([(gx.ravel() + 2) / 4 * 2.5 - 0.3, (gz.ravel() + 2) / 4 * 1.7 + 0.2], 1).astype(f32)      # beyond [0,1]: wrap modes matter
    uv1 = np.stack([(gz.ravel() + 2) / 4, (gx.ravel() + 2) / 4], 1).astype(f32)
    nrm0 = np.tile(np.asarray([0, 1, 0], f32), (g * g, 1))
    idx0 = []
    for j in range(g - 1):
        for i in range(g - 1):
            a = j * g + i
            idx0 += [a, a + g, a + 1, a + 1, a + g, a + g + 1]
    idx0 = np.asarray(idx0, np.uint16)
    pos1 = np.asarray([[-1, 0.3, -1], [-1, 1.6, -1.3], [0.6, 0.3, -1.2], [0.6, 0.3, -1.2], [-1, 1.6, -1.3], [0.6, 1.7, -1.5]], f32)
    col1 = rng.uniform(0.2, 1.0, (6, 4)).astype(f32)
    pos2 = np.asarray([[0.8, 0.5, 0.2], [1.6, 0.5, 0.2], [0.8, 1.3, 0.0], [1.6, 0.5, 0.2], [1.6, 1.3, 0.0], [0.8, 1.3, 0.0]], f32)
    uv2 = np.asarray([[0, 1], [1, 1], [0, 0], [1, 1], [1, 0], [0, 0]], f32)
    parts = [pos0, uv0, uv1, nrm0, idx0, pos1, col1, pos2, uv2]
    offs, blob = [], b""
    for p in parts:
        blob += b"\0" * (-len(blob) % 4)
        offs.append(len(blob))
        blob += p.tobytes()
    views = [{"buffer": 0, "byteOffset": o, "byteLength": p.nbytes} for o, p in zip(offs, parts)]

    def mm(a):
        return [float(x) for x in a.min(0)], [float(x) for x in a.max(0)]
    acc = [
        {"bufferView": 0, "componentType": 5126, "count": g * g, "type": "VEC3", "min": mm(pos0)[0], "max": mm(pos0)[1]},
        {"bufferView": 1, "componentType": 5126, "count": g * g, "type": "VEC2"},
        {"bufferView": 2, "componentType": 5126, "count": g * g, "type": "VEC2"},
        {"bufferView": 3, "componentType": 5126, "count": g * g, "type": "VEC3"},
        {"bufferView": 4, "componentType": 5123, "count": int(idx0.size), "type": "SCALAR"},
        {"bufferView": 5, "componentType": 5126, "count": 6, "type": "VEC3", "min": mm(pos1)[0], "max": mm(pos1)[1]},
        {"bufferView": 6, "componentType": 5126, "count": 6, "type": "VEC4"},
        {"bufferView": 7, "componentType": 5126, "count": 6, "type": "VEC3", "min": mm(pos2)[0], "max": mm(pos2)[1]},
        {"bufferView": 8, "componentType": 5126, "count": 6, "type": "VEC2"},
    ]
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
        "bufferViews": views, "accessors": acc,
        "images": [{"uri": uri(png_rgba)}, {"uri": uri(png_pal)}, {"uri": "nm16.png"}, {"uri": uri(png_mr)}],
        "samplers": [{"wrapS": 10497, "wrapT": 33648}, {"wrapS": 33071, "wrapT": 33071, "minFilter": 9728}, {}],
        "textures": [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}, {"source": 2, "sampler": 2}, {"source": 3}, {"source": 0}],
        "materials": [
            {"name": "ground", "pbrMetallicRoughness": {
                "baseColorFactor": [0.9, 0.8, 0.7, 1.0], "metallicFactor": 0.9, "roughnessFactor": 0.8,
                "baseColorTexture": {"index": 0, "extensions": {"KHR_texture_transform": {"offset": [0.1, 0.05], "rotation": 0.4, "scale": [1.5, 0.75]}}},
                "metallicRoughnessTexture": {"index": 3, "texCoord": 1}},
             "normalTexture": {"index": 2, "texCoord": 1, "extensions": {"KHR_texture_transform": {"rotation": -0.7}}}},
            {"name": "painted", "pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1], "metallicFactor": 0.1, "roughnessFactor": 0.6}},
            {"name": "lamp", "pbrMetallicRoughness": {"baseColorTexture": {"index": 1}, "metallicFactor": 0.0, "roughnessFactor": 0.5},
             "emissiveFactor": [0.6, 0.5, 0.4], "emissiveTexture": {"index": 4}},
        ],
        "meshes": [
            {"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "TEXCOORD_1": 2, "NORMAL": 3}, "indices": 4, "material": 0}]},
            {"primitives": [{"attributes": {"POSITION": 5, "COLOR_0": 6}, "material": 1},
                            {"attributes": {"POSITION": 7, "TEXCOORD_0": 8}, "material": 2}]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "aspectRatio": 1.5, "znear": 0.1}}],
        "nodes": [{"mesh": 0}, {"mesh": 1, "rotation": [0.0, 0.2588190451, 0.0, 0.9659258263], "translation": [0.1, 0.0, 0.3]},
                  {"camera": 0, "translation": [0.4, 2.2, 4.5], "rotation": [-0.1736481777, 0.0, 0.0, 0.984807753]}],
        "scenes": [{"nodes": [0, 1, 2]}], "scene": 0,
    }
    path = tmpdir / "textured.gltf"
    path.write_text(json.dumps(doc))
    return path


def write_alpha(tmpdir, seed=4, stack=0, tess=1, jitter=None):
    """Alpha MASK / BLEND materials, single- and double-sided: a ground, a masked fence, two stacked blend panes (textured
    alpha), an untextured blend pane, and two opaque panes seen from behind (one single sided -> culled for camera rays).
    stack = n: n more horizontal glass panes above the ground, every fifth one TWICE (coincident surfaces: candidates at equal t) --
    occlusion rays from below cross up to n + n // 5 attenuating surfaces.  tess = k: the ground as k x k quads (a tree deep enough for
    the entry-pool form of the any-hit kernels, which shallow trees do not take).  jitter = s: the stacked panes at random heights, sizes and
    tilts (seed s), a third of them coincident twins of the pane before, materials drawn at random."""
    import _png_fixture as PF
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:8, 0:8]
    fence = np.zeros((8, 8, 4), int)
    fence[..., :3] = rng.integers(80, 256, (8, 8, 3))
    fence[..., 3] = np.where((xx + yy) % 3 == 0, 0, np.where((xx * yy) % 5 == 0, 100, 255))
    glass = np.zeros((4, 4, 4), int)
    glass[..., :3] = rng.integers(0, 256, (4, 4, 3))
    glass[..., 3] = rng.integers(60, 220, (4, 4))
    uri = lambda b: "data:image/png;base64," + base64.b64encode(b).decode()          # noqa: E731
    quad_uv = np.asarray([[0, 0], [1, 0], [1, 1], [0, 0], [1, 1], [0, 1]], f32)

    def quad(p0, du, dv):
        p0, du, dv = [np.asarray(x, f32) for x in (p0, du, dv)]
        return np.stack([p0, p0 + du, p0 + du + dv, p0, p0 + du + dv, p0 + dv]).astype(f32)
    quads = [
        quad((-3, 0, 3), (6, 0, 0), (0, 0, -6)),            # 0 ground (+y)
        quad((-1.5, 0, 0.5), (1.4, 0, 0), (0, 1.6, 0)),     # 1 fence, faces +z
        quad((0.2, 0.1, 0.9), (1.3, 0, 0), (0, 1.4, 0)),    # 2 pane A (blend, textured)
        quad((0.4, 0.2, 0.3), (1.3, 0, 0), (0, 1.4, 0)),    # 3 pane B behind A
        quad((-0.6, 0.9, 1.4), (0.9, 0, 0), (0, 0.8, 0)),   # 4 untextured blend pane
        quad((-2.6, 0.1, -0.5), (0, 0, 1.0), (0, 1.2, 0)),  # 5 opaque single sided, winding faces -x: seen from behind
        quad((1.9, 0.1, -0.5), (0, 1.2, 0), (0, 0, 1.0)),   # 6 opaque double sided, faces -x as well from the camera? (+x normal flipped)
    ]
    mats_of = [0, 1, 2, 2, 3, 4, 5]
    if tess > 1:
        quads[0:1] = []
        mats_of[0:1] = []
        c = 6.0 / tess
        for i in range(tess):
            for j in range(tess):
                quads.append(quad((-3 + c * i, 0, 3 - c * j), (c, 0, 0), (0, 0, -c))); mats_of.append(0)
    jr = np.random.default_rng(jitter) if jitter is not None else None
    pane = None
    for k in range(stack):
        if jr is None:
            pane = quad((0.9, 0.25 + 0.06 * k, 0.2), (1.8, 0, 0), (0, 0, -2.2))
            for _ in range(2 if k % 5 == 4 else 1):
                quads.append(pane); mats_of.append(2 if k % 3 else 1)          # glass, every third one the masked fence texture
        else:
            if pane is None or jr.random() > 0.33:                            # else: a coincident twin of the pane before (another material, perhaps)
                pane = quad((0.3 + jr.random(), 0.2 + 1.6 * jr.random(), 0.6 * jr.random()), (1.0 + jr.random(), 0.3 * (jr.random() - 0.5), 0), (0, 0.3 * (jr.random() - 0.5), -1.0 - 1.5 * jr.random()))
            quads.append(pane); mats_of.append(int(jr.integers(1, 4)))          # fence (MASK), glass (BLEND, textured), film (BLEND, untextured)
    parts, views, accs, prims = [], [], [], []
    blob = b""
    for k, qv in enumerate(quads):
        for arr, typ in ((qv, "VEC3"), (quad_uv, "VEC2")):
            blob += b"\0" * (-len(blob) % 4)
            views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
            a = {"bufferView": len(views) - 1, "componentType": 5126, "count": 6, "type": typ}
            if typ == "VEC3":
                a["min"] = [float(x) for x in arr.min(0)]; a["max"] = [float(x) for x in arr.max(0)]
            accs.append(a)
            blob += arr.tobytes()
        prims.append({"attributes": {"POSITION": 2 * k, "TEXCOORD_0": 2 * k + 1}, "material": mats_of[k]})
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
        "bufferViews": views, "accessors": accs,
        "images": [{"uri": uri(PF.encode(fence, 6, 8)[0])}, {"uri": uri(PF.encode(glass, 6, 8)[0])}],
        "samplers": [{"minFilter": 9728}, {}],
        "textures": [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}],
        "materials": [
            {"name": "ground", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.8, 0.75, 1], "metallicFactor": 0.0, "roughnessFactor": 0.9}},
            {"name": "fence", "alphaMode": "MASK", "alphaCutoff": 0.5, "doubleSided": True,
             "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0, "roughnessFactor": 0.7}},
            {"name": "glass", "alphaMode": "BLEND", "doubleSided": True,
             "pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 0.9], "baseColorTexture": {"index": 1}, "metallicFactor": 0.2, "roughnessFactor": 0.4}},
            {"name": "film", "alphaMode": "BLEND", "doubleSided": True,
             "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.3, 0.2, 0.5], "metallicFactor": 0.0, "roughnessFactor": 0.8}},
            {"name": "wall1", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.7, 0.3, 1], "metallicFactor": 0.0, "roughnessFactor": 0.8}},
            {"name": "wall2", "doubleSided": True, "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.3, 0.8, 1], "metallicFactor": 0.0, "roughnessFactor": 0.8}},
        ],
        "meshes": [{"primitives": prims}],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.9, "aspectRatio": 1.5, "znear": 0.1}}],
        "nodes": [{"mesh": 0}, {"camera": 0, "translation": [0.0, 1.6, 5.0], "rotation": [-0.1305261922, 0.0, 0.0, 0.9914448614]}],
        "scenes": [{"nodes": [0, 1]}], "scene": 0,
    }
    path = tmpdir / ("alpha.gltf" if not stack and tess == 1 else "alpha_stack%d_%d_%s.gltf" % (stack, tess, jitter))
    path.write_text(json.dumps(doc))
    return path
