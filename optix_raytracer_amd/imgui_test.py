"""Host side of SDK/imgui_test (the fork author's tracer) above the C ABI: the sample's Camera, light and
material classes, Params, the NBT mesh reader of TriangleGAS and the body of TracerWindow::run's loop, with
the reference's names (imgui_test/{camera.h, light.h, *_light.h, diffuse.h, optixTriangle.h, triangle_gas.cpp,
tracer_window.cpp}).  Display, ImGui panel and key handling are out of scope.

NBT: `awegsche/nbt` is not vendored (SDK/ext/CMakeLists.txt:13-18); the reader below implements the public
NBT format (big-endian tagged tree, optionally gzip) for the one shape load_nbt consumes: a root compound
of mesh compounds, each with TAG_Byte_Array "vertices" and "normals" holding little-endian float3s
(triangle_gas.cpp:16-75)."""
import ctypes as C
import gzip
import struct

import numpy as np
import torch

from . import _lib as L
from . import api

# ------------------------------------------------------------------------------------------------ NBT
TAG_END, TAG_BYTE, TAG_SHORT, TAG_INT, TAG_LONG, TAG_FLOAT, TAG_DOUBLE, TAG_BYTE_ARRAY, TAG_STRING, TAG_LIST, TAG_COMPOUND, \
    TAG_INT_ARRAY, TAG_LONG_ARRAY = range(13)


class _Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def take(self, n):
        if self.p + n > len(self.d):
            raise ValueError("NBT: unexpected end of data")
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def num(self, fmt):
        return struct.unpack(">" + fmt, self.take(struct.calcsize(fmt)))[0]

    def string(self):
        return self.take(self.num("H")).decode("utf-8", "replace")

    def payload(self, tag):
        if tag == TAG_BYTE: return self.num("b")
        if tag == TAG_SHORT: return self.num("h")
        if tag == TAG_INT: return self.num("i")
        if tag == TAG_LONG: return self.num("q")
        if tag == TAG_FLOAT: return self.num("f")
        if tag == TAG_DOUBLE: return self.num("d")
        if tag == TAG_BYTE_ARRAY: return bytes(self.take(self.num("i")))
        if tag == TAG_STRING: return self.string()
        if tag == TAG_LIST:
            t, n = self.num("b"), self.num("i")
            return [self.payload(t) for _ in range(max(n, 0))]
        if tag == TAG_COMPOUND:
            out = {}
            while True:
                t = self.num("b")
                if t == TAG_END:
                    return out
                name = self.string()
                out[name] = self.payload(t)
        if tag == TAG_INT_ARRAY: return [self.num("i") for _ in range(self.num("i"))]
        if tag == TAG_LONG_ARRAY: return [self.num("q") for _ in range(self.num("i"))]
        raise ValueError("NBT: unknown tag %d" % tag)


def read_nbt(filename):
    data = open(filename, "rb").read()
    if data[:2] == b"\x1f\x8b":
        data = gzip.decompress(data)
    r = _Reader(data)
    if r.num("b") != TAG_COMPOUND:
        raise ValueError("NBT: root is not a compound")
    r.string()
    return r.payload(TAG_COMPOUND)


def write_nbt(filename, meshes, compress=True):
    """meshes: {name: (vertices (n,3) f32, normals (n,3) f32)} -> the layout load_nbt reads"""
    def s(x):
        b = x.encode()
        return struct.pack(">H", len(b)) + b
    body = b""
    for name, (v, n) in meshes.items():
        inner = b""
        for key, arr in (("vertices", v), ("normals", n)):
            raw = np.ascontiguousarray(arr, "<f4").tobytes()
            inner += struct.pack(">b", TAG_BYTE_ARRAY) + s(key) + struct.pack(">i", len(raw)) + raw
        body += struct.pack(">b", TAG_COMPOUND) + s(name) + inner + struct.pack(">b", TAG_END)
    blob = struct.pack(">b", TAG_COMPOUND) + s("") + body + struct.pack(">b", TAG_END)
    open(filename, "wb").write(gzip.compress(blob) if compress else blob)


def load_nbt(filename):
    """triangle_gas.cpp:16-75: flat float3 vertices / normals (3 per triangle) and one material index (0) per vertex"""
    import os
    if not os.path.exists(filename):
        raise RuntimeError("can't find file")
    root = read_nbt(filename)
    verts, norms = [], []
    for mesh in root.values():
        v = np.frombuffer(mesh["vertices"], "<f4")
        n = np.frombuffer(mesh["normals"], "<f4")
        nv = len(mesh["vertices"]) // 12
        verts.append(v[:nv * 3].reshape(-1, 3))
        norms.append(n[:nv * 3].reshape(-1, 3))
    vertices = np.concatenate(verts).astype(np.float32) if verts else np.zeros((0, 3), np.float32)
    normals = np.concatenate(norms).astype(np.float32) if norms else np.zeros((0, 3), np.float32)
    return vertices, normals, np.zeros(vertices.shape[0], np.int32)


# ------------------------------------------------------------------------------------------------ device objects
class Camera:
    """imgui_test/camera.h: thin-lens / orthographic camera"""

    def __init__(self):
        self.c = L.MiptImguiCamera()
        self.c.eye = (C.c_float * 3)(0.0, 1.0, -2.0)
        self.c.lookat = (C.c_float * 3)(0.0, 0.0, 0.0)
        self.c.up = (C.c_float * 3)(0.0, 1.0, 0.000073)
        self.c.ortho, self.c.fov, self.c.fd, self.c.aperture, self.c.speed = 0, 45.0, 1.0, 0.0, 0.01

    def set_lookat(self, v): self.c.lookat = (C.c_float * 3)(*map(float, v))
    def set_up(self, v): self.c.up = (C.c_float * 3)(*map(float, v))
    def set_eye(self, v): self.c.eye = (C.c_float * 3)(*map(float, v))
    def set_fd(self, fd): L.load().mipt_imgui_camera_set_fd(C.byref(self.c), float(fd))
    def set_fov(self, v): self.c.fov = float(v)
    def set_aperture(self, v): self.c.aperture = float(v)
    def set_ortho(self, v): self.c.ortho = 1 if v else 0
    def compute_uvw(self): L.load().mipt_imgui_camera_compute_uvw(C.byref(self.c))
    def pack(self): return bytes(self.c)


def PointLight(position, lumi):
    l = L.MiptImguiLight()
    l.a = (C.c_float * 3)(*map(float, position)); l.b = (C.c_float * 4)(*map(float, lumi), 0.0); l.tag = 0
    return l


def DirectionalLight(direction, lumi, jitter):
    l = L.MiptImguiLight()
    l.a = (C.c_float * 3)(*map(float, direction)); l.b = (C.c_float * 4)(*map(float, lumi), float(jitter)); l.tag = 1
    return l


def VolumetricLight(position, radius, lumi):
    l = L.MiptImguiLight()
    l.a = (C.c_float * 3)(*map(float, position)); l.b = (C.c_float * 4)(float(radius), *map(float, lumi)); l.tag = 2
    return l


def _to_device(raw, device):
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


class TriangleGAS:
    """imgui_test/triangle_gas.cpp:170-241: vertices/normals on the device + the acceleration structure"""

    def __init__(self, context, device, vertices, normals, mat_indices):
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        self.mat_indices = np.ascontiguousarray(mat_indices, np.int32)
        self.d_vertices = torch.from_numpy(self.vertices).to(device)
        self.d_normals = torch.from_numpy(self.normals).to(device)
        self.handle = context.accel_build([{"vertices": self.d_vertices.data_ptr(), "num_vertices": self.vertices.shape[0],
                                            "stride": 12}])

    @classmethod
    def from_nbt(cls, context, device, filename):
        return cls(context, device, *load_nbt(filename))


class Params:
    """imgui_test/optixTriangle.h:42-108"""

    def __init__(self, device):
        self.p = L.MiptImguiParams()
        self.p.image_width, self.p.image_height, self.p.samples_per_frame = 400, 300, 4
        self.p.dt, self.p.dirty, self.p.tfactor = 0, 1, 1.0
        self.device = device
        self._keep = {}

    def _set(self, name, raw):
        t = _to_device(raw, self.device)
        self._keep[name] = t
        setattr(self.p, name, t.data_ptr())

    def set_camera(self, cam): self._set("camera", cam.pack())
    def set_lights(self, lights):
        self._set("lights", b"".join(bytes(x) for x in lights) or b"\0" * 44)
        self.p.nlights = len(lights)

    def set_materials(self, colors):
        a = np.ascontiguousarray(colors, np.float32).reshape(-1, 3)
        self._set("materials", a.tobytes())
        self.p.nmaterials = a.shape[0]

    def set_mat_indices(self, idx):
        a = np.ascontiguousarray(idx, np.int32)
        self._set("mat_indices", a.tobytes() or b"\0" * 4)
        self.p.nmat_indices = a.shape[0]

    def alloc_film(self):
        self.film = torch.zeros((self.p.image_height, self.p.image_width, 3), dtype=torch.float32, device=self.device)
        self.p.film = self.film.data_ptr()

    def frame_step(self): self.p.dt += self.p.samples_per_frame


def default_scene_objects():
    """the scene of imgui_test/main.cpp:236-261"""
    cam = Camera()
    cam.set_eye((0.0, 1.0, -10.0)); cam.set_up((0.0, 0.0000073, 1.0)); cam.set_lookat((0.0, 0.1, 0.0))
    cam.set_aperture(0.0); cam.set_fd(1.0); cam.set_fov(45.0); cam.compute_uvw()
    lights = [VolumetricLight((0.0, 2.0, 0.0), 0.1, (0.1, 0.08, 0.08)), VolumetricLight((2.0, 2.0, 0.0), 0.1, (0.1, 0.08, 0.08)),
              VolumetricLight((2.0, 2.0, 2.0), 0.1, (0.1, 0.08, 0.08)), DirectionalLight((-1.0, 1.0, -1.0), (0.1, 0.1, 0.1), 0.05)]
    mats = [(0.5, 0.5, 0.5)] + [(i / 5.0, j / 5.0, 0.2) for i in range(5) for j in range(5)] + [(0.5, 0.5, 0.5)] * 2
    return cam, lights, mats


def render_frame(context, params, width, height, image, stream=None, dirty=False):
    """one iteration of TracerWindow::run (tracer_window.cpp:80-105) without the GL parts"""
    params.p.dirty = 1 if dirty else 0
    params.p.image = image.data_ptr()
    params.p.image_width, params.p.image_height = width, height
    if dirty:
        params.p.dt = 0
    params.frame_step()
    rc = context._lib.mipt_imgui_launch(context._h, stream, C.byref(params.p), width, height, 1)
    context._check(rc)
    context.sync()
