"""Scene data of the path-tracer sample and the benchmark scene.

cornell_box(): the hard-coded scene of SDK/optixPathTracer/optixPathTracer.cpp:144-314 (32 triangles,
4 materials), kept here as quads (v0,v1,v2,v3 -> triangles (v0,v1,v2),(v0,v2,v3), the order the
reference lists them in), light :435-439, camera :538-541, background :852.

random_soup(): SURVEY.md 8(d) "C4" -- n small random triangles in the unit cube, one Lambert material.
"""
import numpy as np

from . import api

_W, _H, _D = 556.0, 548.8, 559.2

# (material, quad corners)
_CORNELL_QUADS = [
    (0, [(0, 0, 0), (0, 0, _D), (_W, 0, _D), (_W, 0, 0)]),                       # floor
    (0, [(0, _H, 0), (_W, _H, 0), (_W, _H, _D), (0, _H, _D)]),                   # ceiling
    (0, [(0, 0, _D), (0, _H, _D), (_W, _H, _D), (_W, 0, _D)]),                   # back wall
    (1, [(0, 0, 0), (0, _H, 0), (0, _H, _D), (0, 0, _D)]),                       # right wall (green)
    (2, [(_W, 0, 0), (_W, 0, _D), (_W, _H, _D), (_W, _H, 0)]),                   # left wall (red)
    # short block
    (0, [(130, 165, 65), (82, 165, 225), (242, 165, 274), (290, 165, 114)]),
    (0, [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)]),
    (0, [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)]),
    (0, [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)]),
    (0, [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)]),
    # tall block
    (0, [(423, 330, 247), (265, 330, 296), (314, 330, 455), (472, 330, 406)]),
    (0, [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)]),
    (0, [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)]),
    (0, [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)]),
    (0, [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)]),
    # ceiling light
    (3, [(343, 548.6, 227), (213, 548.6, 227), (213, 548.6, 332), (343, 548.6, 332)]),
]

_CORNELL_EMISSION = [(0, 0, 0), (0, 0, 0), (0, 0, 0), (15, 15, 5)]
_CORNELL_DIFFUSE = [(0.80, 0.80, 0.80), (0.05, 0.80, 0.05), (0.80, 0.05, 0.05), (0.50, 0.00, 0.00)]


def _light(corner, v1, v2, emission):
    corner, v1, v2 = (np.asarray(x, np.float32) for x in (corner, v1, v2))
    # normalize(cross(v1, v2)) in float32, operation order of SDK/sutil/vec_math.h:533,545
    f = np.float32
    c = np.array([f(f(v1[1] * v2[2]) - f(v1[2] * v2[1])), f(f(v1[2] * v2[0]) - f(v1[0] * v2[2])),
                  f(f(v1[0] * v2[1]) - f(v1[1] * v2[0]))], np.float32)
    d = f(f(f(c[0] * c[0]) + f(c[1] * c[1])) + f(c[2] * c[2]))
    n = (c * np.float32(np.float32(1.0) / np.sqrt(d, dtype=np.float32))).astype(np.float32)
    return {"corner": corner, "v1": v1, "v2": v2, "normal": n, "emission": np.asarray(emission, np.float32)}


def cornell_box():
    verts, mats = [], []
    for m, q in _CORNELL_QUADS:
        for tri in ((0, 1, 2), (0, 2, 3)):
            for k in tri:
                verts.append((q[k][0], q[k][1], q[k][2], 0.0))
            mats.append(m)
    return {
        "name": "cornell",
        "vertices": np.asarray(verts, np.float32),
        "mat_index": np.asarray(mats, np.uint32),
        "emission": np.asarray(_CORNELL_EMISSION, np.float32),
        "diffuse": np.asarray(_CORNELL_DIFFUSE, np.float32),
        "bg": np.zeros(4, np.float32),
        "light": _light((343.0, 548.5, 227.0), (0.0, 0.0, 105.0), (-130.0, 0.0, 0.0), (15.0, 15.0, 5.0)),
        "camera": {"eye": (278.0, 273.0, -900.0), "lookat": (278.0, 273.0, 330.0), "up": (0.0, 1.0, 0.0), "fovY": 35.0},
    }


def random_soup(n_tris=1_000_000, seed_tag=0x5EED, edge_scale=0.01):
    return {
        "name": "soup%d" % n_tris,
        "vertices": api.random_triangles(n_tris, seed_tag, edge_scale),
        "mat_index": np.zeros(n_tris, np.uint32),
        "emission": np.zeros((1, 3), np.float32),
        "diffuse": np.asarray([(0.8, 0.8, 0.8)], np.float32),
        "bg": np.zeros(4, np.float32),
        # SURVEY.md 8(d) lists v1 = (0,0,.5), v2 = (.5,0,0) and states normal = (0,-1,0); normalize(v1 x v2)
        # of those is (0,+1,0) (facing away from the cube: no shadow ray would ever be cast), so the
        # two edges are swapped here -- same parallelogram, normal (0,-1,0) as the survey intends.
        "light": _light((0.25, 1.5, 0.25), (0.5, 0.0, 0.0), (0.0, 0.0, 0.5), (15.0, 15.0, 5.0)),
        "camera": {"eye": (0.5, 0.5, -1.5), "lookat": (0.5, 0.5, 0.5), "up": (0.0, 1.0, 0.0), "fovY": 35.0},
    }


def launch_desc(scene, width, height, spp=1, subframe=0):
    """Host-side launch description (what initLaunchParams + handleCameraUpdate put into Params)."""
    from .camera import Camera
    c = scene["camera"]
    cam = Camera(c["eye"], c["lookat"], c["up"], c["fovY"], 1.0)
    cam.setAspectRatio(np.float32(width) / np.float32(height))
    U, V, W = cam.UVWFrame()
    return {"width": width, "height": height, "spp": spp, "subframe": subframe,
            "eye": np.asarray(c["eye"], np.float32), "U": U, "V": V, "W": W, "light": scene["light"]}


def from_gltf(filename):
    """BASELINE config 3/5 scenes: a glTF file through the sutil::Scene-compatible loader.  The reference has no
    path tracer over sutil::Scene (only the Whitted PBR programs), so the shading here is build-defined
    (SURVEY.md 8d): Lambert with diffuse = pbr.base_color.rgb (textures ignored for now), no emission, camera
    from Scene::camera(), one downward parallelogram light of side 0.5*maxExtent centred 1.0*maxExtent above the
    scene aabb centre, emission (15,15,5)."""
    sc = api.Scene(filename)
    info = sc.info()
    verts, mat = sc.flatten()
    mats = sc.materials()
    diffuse = np.asarray([m["base_color"][:3] for m in mats] + [(0.8, 0.8, 0.8)], np.float32)   # last = "no material"
    mat = np.where(mat == 0xFFFFFFFF, len(mats), mat).astype(np.uint32)
    lo, hi = info["aabb_min"], info["aabb_max"]
    f = np.float32
    centre = ((lo + hi) * f(0.5)).astype(np.float32)
    ext = f(max(hi - lo))
    side = f(0.5) * ext
    corner = np.array([centre[0] - side * f(0.5), centre[1] + ext, centre[2] - side * f(0.5)], np.float32)
    cam = sc.camera()
    return {
        "name": "gltf:" + str(filename).rsplit("/", 1)[-1],
        "vertices": verts, "mat_index": mat,
        "emission": np.zeros((diffuse.shape[0], 3), np.float32), "diffuse": diffuse,
        "bg": np.zeros(4, np.float32),
        "light": _light(corner, (side, 0.0, 0.0), (0.0, 0.0, side), (15.0, 15.0, 5.0)),
        "camera": {"eye": cam["eye"], "lookat": cam["lookat"], "up": cam["up"], "fovY": cam["fovY"]},
        "info": info,
    }


def write_rooms_gltf(path, n_boxes=20_000, seed=5):
    """BASELINE config 5 asks for a Sponza-class glTF (~250 k triangles); the tree holds none, so this writes a
    deterministic stand-in that still goes through the glTF loader: a hall of 4 x 1.5 x 4 units with
    `n_boxes` axis-aligned boxes of random size on a jittered lattice (12 triangles each, u32 indices, one mesh,
    six materials), a floor, and a perspective camera node looking down the hall.  Returns the path."""
    import base64
    import json
    rng = np.random.default_rng(seed)
    g = int(np.ceil(n_boxes ** (1.0 / 3.0)))
    ijk = np.stack(np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 3)[:n_boxes]
    cell = np.asarray([4.0, 1.5, 4.0]) / g
    centre = (ijk + 0.5 + rng.uniform(-0.3, 0.3, ijk.shape)) * cell
    half = rng.uniform(0.08, 0.3, ijk.shape) * cell
    corners = np.asarray([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64)
    pos = (centre[:, None, :] + corners[None] * half[:, None, :]).reshape(-1, 3).astype(np.float32)
    faces = np.asarray([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]],
                       np.uint32)
    floor = np.asarray([[0, -0.01, 0], [4, -0.01, 0], [4, -0.01, 4], [0, -0.01, 4]], np.float32)
    prims, blob = [], b""
    views, accs = [], []

    def add(arr, target_type, comp):
        nonlocal blob
        blob += b"\0" * (-len(blob) % 4)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
        a = {"bufferView": len(views) - 1, "componentType": comp, "count": int(arr.shape[0] if arr.ndim > 1 else arr.size), "type": target_type}
        if target_type == "VEC3":
            a["min"] = [float(x) for x in arr.min(0)]; a["max"] = [float(x) for x in arr.max(0)]
        accs.append(a)
        blob += arr.tobytes()
        return len(accs) - 1
    n_mat = 6
    owner = rng.integers(0, n_mat, n_boxes)
    for m in range(n_mat):
        sel = np.nonzero(owner == m)[0]
        if sel.size == 0:
            continue
        idx = (faces[None] + (sel[:, None, None] * 8).astype(np.uint32)).reshape(-1)
        prims.append({"attributes": {"POSITION": None}, "indices": add(idx.astype(np.uint32), "SCALAR", 5125), "material": m})
    p_acc = add(pos, "VEC3", 5126)
    for p in prims:
        p["attributes"]["POSITION"] = p_acc
    prims.append({"attributes": {"POSITION": add(floor, "VEC3", 5126)}, "indices": add(np.asarray([0, 2, 1, 0, 3, 2], np.uint32), "SCALAR", 5125), "material": 0})
    (path.parent / (path.stem + ".bin")).write_bytes(blob)
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(blob), "uri": path.stem + ".bin"}], "bufferViews": views, "accessors": accs,
           "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [float(x) for x in rng.uniform(0.3, 0.9, 3)] + [1.0], "metallicFactor": 0.0}}
                         for _ in range(n_mat)],
           "meshes": [{"primitives": prims}],
           "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "aspectRatio": 16 / 9, "znear": 0.05}}],
           "nodes": [{"mesh": 0}, {"camera": 0, "translation": [2.0, 0.75, 5.2]}],
           "scenes": [{"nodes": [0, 1]}], "scene": 0}
    path.write_text(json.dumps(doc))
    return path


def write_duck_like_gltf(path, seed=3):
    """BASELINE config 3 names SDK/data/Duck/Duck.gltf, which does not travel to the GPU box.  This writes a deterministic
    stand-in with the properties of that file that matter to the path: ONE mesh with ONE primitive of
    4 212 triangles, uint16 indices, positions in centimetres (body, head and bill of a duck-like blob, about 170 units
    across), a root node with scale 0.01 whose children are the mesh node and a perspective camera node (yfov 0.6606 rad,
    aspectRatio 1.5) placed like Duck's camera, and one dielectric material (metallicFactor 0).  Returns the path."""
    import base64
    import json
    rng = np.random.default_rng(seed)

    def blob(centre, radii, slices, stacks, bump):
        """closed lat/long surface: 2 * slices * (stacks - 1) triangles, (stacks + 1) * slices + ... vertices"""
        th = np.linspace(0.0, np.pi, stacks + 1)[1:-1]                       # interior rings
        ph = np.linspace(0.0, 2.0 * np.pi, slices, endpoint=False)
        ring = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.cos(th), np.ones_like(ph)), np.outer(np.sin(th), np.sin(ph))], -1).reshape(-1, 3)
        pts = np.concatenate([[[0.0, 1.0, 0.0]], ring, [[0.0, -1.0, 0.0]]])
        pts = pts * (1.0 + bump * rng.standard_normal((pts.shape[0], 1))) * np.asarray(radii) + np.asarray(centre)
        tri = []
        n_r = stacks - 1
        for s in range(slices):
            s1 = (s + 1) % slices
            tri.append([0, 1 + s1, 1 + s])
            for r in range(n_r - 1):
                a, b, c, d = 1 + r * slices + s, 1 + r * slices + s1, 1 + (r + 1) * slices + s, 1 + (r + 1) * slices + s1
                tri += [[a, b, d], [a, d, c]]
            tri.append([1 + (n_r - 1) * slices + s, 1 + (n_r - 1) * slices + s1, 1 + n_r * slices])
        return pts.astype(np.float32), np.asarray(tri, np.uint32)
    parts = [blob((0.0, 60.0, 0.0), (70.0, 45.0, 50.0), 54, 28, 0.004),      # body   2 * 54 * 27 = 2916 triangles
             blob((55.0, 125.0, 0.0), (30.0, 28.0, 27.0), 36, 15, 0.004),    # head   2 * 36 * 14 = 1008
             blob((92.0, 118.0, 0.0), (18.0, 6.0, 11.0), 24, 7, 0.0)]        # bill   2 * 24 * 6  =  288
    pos, idx, base = [], [], 0
    for p, t in parts:
        pos.append(p); idx.append(t + base); base += p.shape[0]
    pos = np.concatenate(pos); idx = np.concatenate(idx).astype(np.uint16).reshape(-1)
    assert idx.size == 3 * 4212 and pos.shape[0] < 65536
    bin0 = pos.tobytes() + idx.tobytes()
    # camera: placed like Duck's (eye (4.00, 4.63, -4.31) after the 0.01 root scale), rotated about an oblique
    # axis so that its transformed up vector is not axis aligned
    q = np.asarray([np.sin(0.2) * 0.6, 0.0, np.sin(0.2) * 0.8, np.cos(0.2)])
    doc = {"asset": {"version": "2.0"},
           "buffers": [{"byteLength": len(bin0), "uri": "data:application/octet-stream;base64," + base64.b64encode(bin0).decode()}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": pos.nbytes}, {"buffer": 0, "byteOffset": pos.nbytes, "byteLength": idx.nbytes}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": int(pos.shape[0]), "type": "VEC3",
                          "min": [float(x) for x in pos.min(0)], "max": [float(x) for x in pos.max(0)]},
                         {"bufferView": 1, "componentType": 5123, "count": int(idx.size), "type": "SCALAR"}],
           "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.85, 0.65, 0.12, 1.0], "metallicFactor": 0.0}}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]}],
           "cameras": [{"type": "perspective", "perspective": {"yfov": 0.660593, "aspectRatio": 1.5, "znear": 1.0, "zfar": 10000.0}}],
           "nodes": [{"children": [1, 2], "scale": [0.01, 0.01, 0.01]}, {"mesh": 0},
                     {"camera": 0, "translation": [400.113, 463.264, -431.078], "rotation": [float(x) for x in q]}],
           "scenes": [{"nodes": [0]}], "scene": 0}
    path.write_text(json.dumps(doc))
    return path


def write_instanced_gltf(path, n_instances=150, seed=9, alpha_textures=None, blob_mode="BLEND"):
    """An instanced scene for the two-level (instance) acceleration structure: TWO meshes -- a bumpy blob of 960 triangles
    (u16 indices) and a box of 12 -- placed `n_instances` times by nodes with random translation / rotation / NON-uniform
    scale (some nested under a scaled parent, so matrices are general affine), over a floor quad; three opaque materials, a
    camera node.  Every node with a mesh becomes one sutil::Scene instance (Scene.cpp:192-205).  Returns the path.
    alpha_textures = (png bytes, png bytes): the blob becomes a BLEND material and the box a double-sided MASK material over
    those RGBA images (both meshes get TEXCOORD_0), so the Whitted any-hit programs run inside the instances.
    blob_mode = "MASK": the blob is cut out by the second image as well (foliage-like: rays end at the first solid texel);
    "OPAQUE": the blob stays opaque and only the boxes are cut out (a mostly opaque scene with one alpha-tested material)."""
    import base64
    import json
    rng = np.random.default_rng(seed)
    st, sl = 17, 30                                                          # 2 * 30 * 16 = 960 triangles
    th = np.linspace(0.0, np.pi, st + 1)[1:-1]
    ph = np.linspace(0.0, 2.0 * np.pi, sl, endpoint=False)
    ring = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.cos(th), np.ones_like(ph)), np.outer(np.sin(th), np.sin(ph))], -1).reshape(-1, 3)
    pts = np.concatenate([[[0.0, 1.0, 0.0]], ring, [[0.0, -1.0, 0.0]]])
    pts = (pts * (1.0 + 0.08 * rng.standard_normal((pts.shape[0], 1))) * (0.5, 0.35, 0.4)).astype(np.float32)
    tri = []
    for s in range(sl):
        s1 = (s + 1) % sl
        tri.append([0, 1 + s1, 1 + s])
        for r in range(st - 2):
            a, b, c, d = 1 + r * sl + s, 1 + r * sl + s1, 1 + (r + 1) * sl + s, 1 + (r + 1) * sl + s1
            tri += [[a, b, d], [a, d, c]]
        tri.append([1 + (st - 2) * sl + s, 1 + (st - 2) * sl + s1, 1 + (st - 1) * sl])
    idx_blob = np.asarray(tri, np.uint16).reshape(-1)
    box = np.asarray([[x, y, z] for x in (-0.3, 0.3) for y in (-0.3, 0.3) for z in (-0.3, 0.3)], np.float32)
    idx_box = np.asarray([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]],
                         np.uint16).reshape(-1)
    floor = np.asarray([[-9, -0.6, -9], [9, -0.6, -9], [9, -0.6, 9], [-9, -0.6, 9]], np.float32)
    idx_floor = np.asarray([0, 2, 1, 0, 3, 2], np.uint16)
    blob, views, accs = b"", [], []

    def add(arr, typ, comp):
        nonlocal blob
        blob += b"\0" * (-len(blob) % 4)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
        a = {"bufferView": len(views) - 1, "componentType": comp, "count": int(arr.shape[0] if arr.ndim > 1 else arr.size), "type": typ}
        if typ == "VEC3":
            a["min"] = [float(x) for x in arr.min(0)]; a["max"] = [float(x) for x in arr.max(0)]
        accs.append(a)
        blob += arr.tobytes()
        return len(accs) - 1
    a_blob, a_box = {"POSITION": add(pts, "VEC3", 5126)}, {"POSITION": add(box, "VEC3", 5126)}
    if alpha_textures is not None:
        uv_blob = np.stack([np.arctan2(pts[:, 2], pts[:, 0]) / (2 * np.pi) + 0.5, np.clip(pts[:, 1] / 0.9 + 0.5, 0, 1)], -1).astype(np.float32)
        uv_box = ((box[:, :2] + box[:, 2:3] * 0.5) / 0.6 + 0.5).astype(np.float32)
        a_blob["TEXCOORD_0"] = add(uv_blob, "VEC2", 5126)
        a_box["TEXCOORD_0"] = add(uv_box, "VEC2", 5126)
    meshes = [{"primitives": [{"attributes": a_blob, "indices": add(idx_blob, "SCALAR", 5123), "material": 0}]},
              {"primitives": [{"attributes": a_box, "indices": add(idx_box, "SCALAR", 5123), "material": 1}]},
              {"primitives": [{"attributes": {"POSITION": add(floor, "VEC3", 5126)}, "indices": add(idx_floor, "SCALAR", 5123), "material": 2}]}]
    nodes = [{"mesh": 2}]
    parents = []
    for i in range(n_instances):
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        n = {"mesh": int(i % 3 != 2), "translation": [float(x) for x in rng.uniform(-6, 6, 3) * (1, 0.35, 1) + (0, 1.2, 0)],
             "rotation": [float(x) for x in q], "scale": [float(x) for x in rng.uniform(0.4, 1.6, 3)]}
        if i % 10 == 0:                                                       # a scaled parent: its child's matrix is a general affine map
            n = {"children": [len(nodes) + 1], "scale": [1.0, float(rng.uniform(0.6, 1.4)), 1.0], "translation": [0.0, float(rng.uniform(0, 0.5)), 0.0]}
            nodes.append(n)
            parents.append(len(nodes) - 1)
            q2 = rng.standard_normal(4); q2 /= np.linalg.norm(q2)
            nodes.append({"mesh": 0, "translation": [float(x) for x in rng.uniform(-5, 5, 3) * (1, 0.3, 1) + (0, 1.5, 0)], "rotation": [float(x) for x in q2],
                          "scale": [float(x) for x in rng.uniform(0.5, 1.2, 3)]})
        else:
            nodes.append(n)
    child = {c for n in nodes for c in n.get("children", [])}
    nodes.append({"camera": 0, "translation": [0.0, 3.5, 13.0], "rotation": [float(np.sin(-0.12)), 0.0, 0.0, float(np.cos(-0.12))]})
    roots = [i for i in range(len(nodes)) if i not in child]
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
           "bufferViews": views, "accessors": accs,
           "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.85, 0.55, 0.2, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.6}},
                         {"pbrMetallicRoughness": {"baseColorFactor": [0.25, 0.5, 0.85, 1.0], "metallicFactor": 0.3, "roughnessFactor": 0.4}},
                         {"pbrMetallicRoughness": {"baseColorFactor": [0.7, 0.7, 0.7, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.9}}],
           "meshes": meshes, "cameras": [{"type": "perspective", "perspective": {"yfov": 0.75, "aspectRatio": 16 / 9, "znear": 0.05}}],
           "nodes": nodes, "scenes": [{"nodes": roots}], "scene": 0}
    if alpha_textures is not None:
        uri = lambda b: "data:image/png;base64," + base64.b64encode(b).decode()
        doc["images"] = [{"uri": uri(alpha_textures[0])}, {"uri": uri(alpha_textures[1])}]
        doc["samplers"] = [{"minFilter": 9728, "magFilter": 9728}, {}]
        doc["textures"] = [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}]
        if blob_mode in ("MASK", "OPAQUE"):
            doc["textures"].append({"source": 1, "sampler": 0})             # nearest texel: alpha is 0 or 1, a ray ends at the first solid one
            if blob_mode == "MASK":
                doc["materials"][0].update({"alphaMode": "MASK", "alphaCutoff": 0.5, "doubleSided": True})
                doc["materials"][0]["pbrMetallicRoughness"].update({"baseColorTexture": {"index": 2}})
        else:
            doc["materials"][0].update({"alphaMode": "BLEND", "doubleSided": True})
            doc["materials"][0]["pbrMetallicRoughness"].update({"baseColorFactor": [0.85, 0.55, 0.2, 0.8], "baseColorTexture": {"index": 0}})
        doc["materials"][1].update({"alphaMode": "MASK", "alphaCutoff": 0.5, "doubleSided": True})
        doc["materials"][1]["pbrMetallicRoughness"].update({"baseColorTexture": {"index": 2 if blob_mode in ("MASK", "OPAQUE") else 1}})
    path.write_text(json.dumps(doc))
    return path


def gltf_two_level(sc):
    """What Scene::buildMeshAccels / buildInstanceAccel consume, from an api.Scene: ([(object-space vertices4, material index)] per
    mesh, [(mesh index, 3x4 object->world matrix)] per instance).  The instances are in mipt_scene_flatten's order, so a hit's
    flattened primitive index addresses the arrays of from_gltf()."""
    n_mesh = sc.info()["num_meshes"]
    return [sc.mesh_triangles(m) for m in range(n_mesh)], sc.instances()
