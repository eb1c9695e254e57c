"""Thin object layer over the C ABI (include/mipt.h).  Errors raise MiptError, the
counterpart of sutil::Exception thrown by OPTIX_CHECK (SDK/sutil/Exception.h:82-112)."""
import ctypes as C

import numpy as np

from . import _lib as L


class MiptError(RuntimeError):
    def __init__(self, code, name, detail):
        super().__init__("%s (%d): %s" % (name, code, detail))
        self.code = code
        self.name = name


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_uvw(eye, lookat, up, fov_y, aspect, fod=1.0):
    """sutil::Camera::UVWFrame (SDK/sutil/Camera.cpp:34-46) evaluated by the library's host code."""
    U, V, W = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    L.load().mipt_camera_uvw(_f3(eye), _f3(lookat), _f3(up), fov_y, aspect, fod, U, V, W)
    return np.array(U, np.float32), np.array(V, np.float32), np.array(W, np.float32)


def random_triangles(n, seed_tag=0x5EED, edge_scale=0.01):
    """Synthetic scene of SURVEY.md 8(d): returns (n*3, 4) float32 vertices (float4 stride)."""
    out = np.empty((n * 3, 4), np.float32)
    rc = L.load().mipt_scene_random_triangles(n, seed_tag, edge_scale, out.ctypes.data)
    if rc:
        raise MiptError(rc, "mipt_scene_random_triangles", "failed")
    return out


def shard_num_samples(width, height, world):
    n = C.c_uint32()
    rc = L.load().mipt_shard_num_samples(width, height, world, C.byref(n))
    if rc:
        raise MiptError(rc, "MIPT_ERROR_INVALID_VALUE", "shard_num_samples")
    return n.value


def shard_pixel(width, height, world, rank, sample):
    x, y = C.c_int32(), C.c_int32()
    rc = L.load().mipt_shard_pixel(width, height, world, rank, sample, C.byref(x), C.byref(y))
    if rc:
        raise MiptError(rc, "MIPT_ERROR_INVALID_VALUE", "shard_pixel")
    return x.value, y.value


def host_bvh(vertices4, spatial=None):
    """Builds the BVH on the host only (no GPU): returns (nodes (n,16) f32 view of 64-B nodes,
    tris (m,12) f32 view of 48-B triangles, info dict).  vertices4: (n*3, 4) float32.  spatial: None = the object-split
    builder; a float = the builder with spatial splits (MIPT_OPT_BUILD = 3) and that many extra references per triangle."""
    lib = L.load()
    v = np.ascontiguousarray(np.asarray(vertices4, np.float32).reshape(-1, 4)[:, :3]).reshape(-1, 9)
    h = C.c_void_p()
    if spatial is None:
        rc = lib.mipt_host_bvh_create(v.ctypes.data, v.shape[0], C.byref(h))
    else:
        rc = lib.mipt_host_bvh_create_spatial(v.ctypes.data, v.shape[0], float(spatial), C.byref(h))
    if rc:
        raise MiptError(rc, lib.mipt_error_name(rc).decode(), "mipt_host_bvh_create")
    try:
        info = L.MiptAccelInfo()
        lib.mipt_host_bvh_info(h, C.byref(info))
        nodes = np.zeros((info.num_nodes, 16), np.float32)
        tris = np.zeros((max(info.num_triangles, 1), 12), np.float32)
        lib.mipt_host_bvh_copy(h, nodes.ctypes.data, tris.ctypes.data)
    finally:
        lib.mipt_host_bvh_destroy(h)
    return nodes, tris[:info.num_triangles], {k: getattr(info, k) for k, _ in info._fields_}


def _texture_info(t):
    return {"texture": t.texture, "texcoord": t.texcoord, "offset": tuple(t.offset), "rotation": tuple(t.rotation),
            "scale": tuple(t.scale)}


class Scene:
    """Host scene loaded from glTF: the data side of sutil::Scene (SDK/sutil/Scene.h:53-159) behind
    sutil::loadScene (Scene.cpp:267).  Raises MiptError on a malformed file, like the reference throws."""

    def __init__(self, filename):
        self._lib = L.load()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self._lib.mipt_scene_load(str(filename).encode(), C.byref(self._h), err, 512)
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), err.value.decode() or "mipt_scene_load")

    def __del__(self):
        try:
            if self._h:
                self._lib.mipt_scene_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), "mipt_scene_*")

    def info(self):
        i = L.MiptSceneInfo()
        self._lib.mipt_scene_info(self._h, C.byref(i))
        d = {k: getattr(i, k) for k, _ in i._fields_}
        d["aabb_min"] = np.array(i.aabb_min, np.float32)
        d["aabb_max"] = np.array(i.aabb_max, np.float32)
        return d

    def camera(self):
        """Scene::camera(): dict(eye, lookat, up, fovY, aspect)"""
        e, l, u = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
        f, a = C.c_float(), C.c_float()
        self._lib.mipt_scene_camera(self._h, e, l, u, C.byref(f), C.byref(a))
        return {"eye": np.array(e, np.float32), "lookat": np.array(l, np.float32), "up": np.array(u, np.float32),
                "fovY": f.value, "aspect": a.value}

    def materials(self):
        out = []
        for k in range(self.info()["num_materials"]):
            m = L.MiptMaterialData()
            self._lib.mipt_scene_material(self._h, k, C.byref(m))
            out.append({"base_color": np.array(m.base_color, np.float32), "metallic": m.metallic, "roughness": m.roughness,
                        "emissive_factor": np.array(m.emissive_factor, np.float32), "alpha_mode": m.alpha_mode,
                        "alpha_cutoff": m.alpha_cutoff, "double_sided": bool(m.double_sided),
                        "base_color_texture": m.base_color_tex.texture,
                        **{k: _texture_info(getattr(m, k)) for k in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emissive_tex")}})
        return out

    def flatten(self):
        """(n*3, 4) float32 world-space vertices, (n,) uint32 material indices"""
        n = self.info()["num_triangles"]
        v = np.zeros((max(n, 1) * 3, 4), np.float32)
        m = np.zeros(max(n, 1), np.uint32)
        rc = self._lib.mipt_scene_flatten(self._h, v.ctypes.data, m.ctypes.data)
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), "mipt_scene_flatten")
        return v[:n * 3], m[:n]

    def mesh_triangles(self, mesh):
        """object-space triangles of mesh `mesh`: (n*3, 4) float32 vertices, (n,) uint32 material index"""
        n = C.c_uint32()
        self._check(self._lib.mipt_scene_mesh_num_triangles(self._h, mesh, C.byref(n)))
        v = np.zeros((n.value * 3, 4), np.float32)
        m = np.zeros(max(n.value, 1), np.uint32)
        self._check(self._lib.mipt_scene_mesh_triangles(self._h, mesh, v.ctypes.data, m.ctypes.data))
        return v, m[:n.value]

    def mesh_shading(self, mesh):
        """the per-mesh OBJECT-space shading attributes (mipt_whitted_set_geometry_instanced), in mesh_triangles' order:
        dict(normals (n*3, 4) {n, 1} or zeros, texcoords [(uv (n*3, 2), present (n,))] * 2, colors (rgba (n*3, 4), present (n,)))"""
        n = C.c_uint32()
        self._check(self._lib.mipt_scene_mesh_num_triangles(self._h, mesh, C.byref(n)))
        k = max(n.value, 1)
        nrm = np.zeros((k * 3, 4), np.float32)
        self._check(self._lib.mipt_scene_mesh_normals(self._h, mesh, nrm.ctypes.data))
        tcs = []
        for j in range(2):
            uv, has = np.zeros((k * 3, 2), np.float32), np.zeros(k, np.uint8)
            self._check(self._lib.mipt_scene_mesh_texcoords(self._h, mesh, j, uv.ctypes.data, has.ctypes.data))
            tcs.append((uv[:n.value * 3], has[:n.value]))
        col, has = np.ones((k * 3, 4), np.float32), np.zeros(k, np.uint8)
        self._check(self._lib.mipt_scene_mesh_colors(self._h, mesh, col.ctypes.data, has.ctypes.data))
        return {"normals": nrm[:n.value * 3], "texcoords": tcs, "colors": (col[:n.value * 3], has[:n.value])}

    def instances(self):
        """[(mesh index, (3, 4) float32 row-major object->world matrix)] in Scene::instances() order"""
        out = []
        for i in range(self.info()["num_instances"]):
            mesh, xf = C.c_uint32(), (C.c_float * 12)()
            self._check(self._lib.mipt_scene_instance(self._h, i, C.byref(mesh), xf))
            out.append((mesh.value, np.array(xf, np.float32).reshape(3, 4)))
        return out

    def flatten_normals(self):
        """(n*3, 4) float32 {world-space normal, 1}; zero rows where a primitive has no NORMAL accessor"""
        n = self.info()["num_triangles"]
        v = np.zeros((max(n, 1) * 3, 4), np.float32)
        rc = self._lib.mipt_scene_flatten_normals(self._h, v.ctypes.data)
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), "mipt_scene_flatten_normals")
        return v[:n * 3]

    def textures(self):
        """list of dicts: width, height, bits, address_s/t (0 wrap, 1 clamp, 2 mirror), filter (0 point, 1 linear), image,
        texels = (h, w, 4) uint8 / uint16 array (a copy) or None when the image could not be decoded"""
        out = []
        for k in range(self.info()["num_textures"]):
            t = L.MiptTextureData()
            self._lib.mipt_scene_texture(self._h, k, C.byref(t))
            d = {f: getattr(t, f) for f in ("width", "height", "bits", "address_s", "address_t", "filter", "image")}
            d["texels"] = None
            if t.texels and t.width:
                dt = np.uint16 if t.bits == 16 else np.uint8
                n = t.width * t.height * 4
                buf = (C.c_uint16 * n if t.bits == 16 else C.c_uint8 * n).from_address(t.texels)
                d["texels"] = np.frombuffer(buf, dt).reshape(t.height, t.width, 4).copy()
            out.append(d)
        return out

    def flatten_texcoords(self, tex_set=0):
        """(n*3, 2) float32 TEXCOORD_<set> per corner, (n,) uint8 presence per triangle"""
        n = self.info()["num_triangles"]
        uv = np.zeros((max(n, 1) * 3, 2), np.float32)
        present = np.zeros(max(n, 1), np.uint8)
        rc = self._lib.mipt_scene_flatten_texcoords(self._h, tex_set, uv.ctypes.data, present.ctypes.data)
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), "mipt_scene_flatten_texcoords")
        return uv[:n * 3], present[:n]

    def flatten_colors(self):
        """(n*3, 4) float32 COLOR_0 per corner (1 where absent), (n,) uint8 presence per triangle"""
        n = self.info()["num_triangles"]
        col = np.ones((max(n, 1) * 3, 4), np.float32)
        present = np.zeros(max(n, 1), np.uint8)
        rc = self._lib.mipt_scene_flatten_colors(self._h, col.ctypes.data, present.ctypes.data)
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), "mipt_scene_flatten_colors")
        return col[:n * 3], present[:n]

    def texture_table(self):
        """ctypes array of MiptTextureData whose texel pointers point into the scene (valid while it lives)"""
        n = self.info()["num_textures"]
        tab = (L.MiptTextureData * max(n, 1))()
        for k in range(n):
            self._lib.mipt_scene_texture(self._h, k, C.byref(tab[k]))
        return tab, n

    def material_table(self):
        """ctypes array of MiptMaterialData: the scene's materials + one record with the defaults of `MaterialData()`
        (cuda/MaterialData.h:44-51: base colour 1, metallic 1, roughness 1, no textures), which is what Scene::createSBT gives a
        primitive without a material (Scene.cpp:1423-1428)"""
        n = self.info()["num_materials"]
        tab = (L.MiptMaterialData * (n + 1))()
        for k in range(n):
            self._lib.mipt_scene_material(self._h, k, C.byref(tab[k]))
        d = tab[n]
        d.base_color = (C.c_float * 4)(1.0, 1.0, 1.0, 1.0)
        d.metallic, d.roughness = 1.0, 1.0
        for name in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emissive_tex"):
            ti = getattr(d, name)
            ti.texture, ti.texcoord = -1, 0
            ti.rotation = (C.c_float * 2)(0.0, 1.0)
            ti.scale = (C.c_float * 2)(1.0, 1.0)
        return tab


class Context:
    """A device context (createContext, optixPathTracer.cpp:555-573)."""

    def __init__(self, device=0, log=None, log_level=0):
        self._lib = L.load()
        self._h = C.c_void_p()
        self._log_py = log

        def _cb(level, tag, msg, _):
            if self._log_py:
                self._log_py(level, tag.decode(), msg.decode())
        self._cb = L.LOG_CALLBACK(_cb)
        rc = self._lib.mipt_context_create(device, self._cb, None, log_level, C.byref(self._h))
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), "mipt_context_create(device=%d)" % device)
        self.device = device

    @classmethod
    def _borrow(cls, handle, device):
        """A Context object over a MiptContext that someone else owns (a member of a Group): destroy() does nothing."""
        self = cls.__new__(cls)
        self._lib = L.load()
        self._h = C.c_void_p(handle)
        self._log_py, self._cb, self.device, self._borrowed = None, None, device, True
        return self

    def _check(self, rc):
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(),
                            self._lib.mipt_last_error_string(self._h).decode())

    def destroy(self):
        if self._h and not getattr(self, "_borrowed", False):
            self._lib.mipt_context_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # ---- options
    def set_option(self, opt, value):
        self._check(self._lib.mipt_set_option(self._h, opt, int(value)))

    def get_option(self, opt):
        v = C.c_uint64()
        self._check(self._lib.mipt_get_option(self._h, opt, C.byref(v)))
        return v.value

    # ---- accel
    def _build_inputs(self, inputs):
        arr = (L.MiptBuildInput * len(inputs))()
        keep = []
        for i, d in enumerate(inputs):
            bi = arr[i]

            def ptr(x):
                if x is None:
                    return None
                if isinstance(x, np.ndarray):
                    x = np.ascontiguousarray(x)
                    keep.append(x)
                    return x.ctypes.data
                return int(x)
            bi.vertices = ptr(d["vertices"])
            bi.num_vertices = int(d["num_vertices"])
            bi.vertex_stride_bytes = int(d.get("stride", 0))
            bi.indices = ptr(d.get("indices"))
            bi.index_format = int(d.get("index_format", 0))
            bi.index_stride_bytes = int(d.get("index_stride", 0))
            bi.num_triangles = int(d.get("num_triangles", 0))
            bi.sbt_index_offsets = ptr(d.get("sbt_index_offsets"))
            fl = d.get("flags")
            if fl is not None:
                fl = np.ascontiguousarray(fl, dtype=np.uint32)
                keep.append(fl)
                bi.flags, bi.num_sbt_records = fl.ctypes.data, fl.shape[0]
            t = d.get("transform")
            if t is not None:
                t = np.ascontiguousarray(t, dtype=np.float32).reshape(12)
                keep.append(t)
                bi.transform3x4 = t.ctypes.data
        return arr, keep

    def accel_build(self, inputs):
        """inputs: list of dicts {vertices(ptr|ndarray), num_vertices, stride, indices, index_format,
        num_triangles, sbt_index_offsets, transform, flags (one MiptGeometryFlags word per SBT record)}"""
        arr, keep = self._build_inputs(inputs)
        h = C.c_ulonglong()
        self._check(self._lib.mipt_accel_build(self._h, arr, len(inputs), C.byref(h)))
        return h.value

    def accel_update(self, handle, inputs):
        """refit of `handle` to moved vertices (same inputs as accel_build otherwise)"""
        arr, keep = self._build_inputs(inputs)
        self._check(self._lib.mipt_accel_update(self._h, handle, arr, len(inputs)))

    def accel_build_instances(self, instances):
        """instances: [(gas_handle, (3, 4) object->world matrix)] -> handle of the two-level structure (OptixInstance records with
        instanceId = index, visibilityMask 1, as Scene::buildInstanceAccel fills them, Scene.cpp:1143-1156)"""
        arr = self._instance_array(instances)
        h = C.c_ulonglong()
        self._check(self._lib.mipt_accel_build_instances(self._h, arr, len(instances), C.byref(h)))
        return h.value

    def accel_update_instances(self, handle, instances):
        arr = self._instance_array(instances)
        self._check(self._lib.mipt_accel_update_instances(self._h, handle, arr, len(instances)))

    @staticmethod
    def _instance_array(instances):
        arr = (L.MiptInstance * len(instances))()
        for i, (gas, xf) in enumerate(instances):
            arr[i].transform = (C.c_float * 12)(*np.asarray(xf, np.float32).reshape(-1)[:12].tolist())
            arr[i].instanceId, arr[i].sbtOffset, arr[i].visibilityMask, arr[i].flags = i, 0, 1, 0
            arr[i].traversableHandle = int(gas)
        return arr

    def accel_export_instances(self, handle):
        """(top-level nodes (n,16) f32, order (k,) u32, inv (k,12) f32, prim_base (k,) u32, gas handles (k,) u64)"""
        info = self.accel_info(handle)
        k = info["num_instances"]
        nodes = np.zeros((info["num_nodes"], 16), np.float32)
        order, base = np.zeros(k, np.uint32), np.zeros(k, np.uint32)
        inv, gas = np.zeros((k, 12), np.float32), np.zeros(k, np.uint64)
        self._check(self._lib.mipt_accel_export_instances(self._h, handle, nodes.ctypes.data, order.ctypes.data, inv.ctypes.data,
                                                         base.ctypes.data, gas.ctypes.data))
        return nodes, order, inv, base, gas

    def accel_destroy(self, handle):
        self._check(self._lib.mipt_accel_destroy(self._h, handle))

    def accel_info(self, handle):
        info = L.MiptAccelInfo()
        self._check(self._lib.mipt_accel_info(self._h, handle, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}

    def accel_export(self, handle):
        info = self.accel_info(handle)
        nodes = np.zeros((info["num_nodes"], 16), np.float32)
        tris = np.zeros((max(info["num_triangles"], 1), 12), np.float32)
        self._check(self._lib.mipt_accel_export(self._h, handle, nodes.ctypes.data, tris.ctypes.data))
        return nodes, tris[:info["num_triangles"]]

    # ---- SBT
    def sbt_set_hitgroups(self, records):
        """records: list of (emission3, diffuse3, device_vertices_ptr)"""
        arr = (L.MiptHitGroupData * max(len(records), 1))()
        for i, (em, df, vp) in enumerate(records):
            arr[i].emission_color = _f3(em)
            arr[i].diffuse_color = _f3(df)
            arr[i].vertices = int(vp) if vp else None
        self._check(self._lib.mipt_sbt_set_hitgroups(self._h, arr, len(records)))

    def sbt_set_miss(self, bg):
        m = L.MiptMissData()
        m.bg_color = (C.c_float * 4)(*[float(x) for x in bg])
        self._check(self._lib.mipt_sbt_set_miss(self._h, C.byref(m)))

    # ---- launch
    def launch(self, params, width, height, depth=1, stream=None):
        self._check(self._lib.mipt_launch(self._h, stream, C.byref(params), width, height, depth))

    def launch_shard(self, params, width, height, rank, world, compact_accum_ptr, stream=None):
        self._check(self._lib.mipt_launch_shard(self._h, stream, C.byref(params), width, height, rank, world,
                                                compact_accum_ptr))

    def unpermute(self, gathered_ptr, world, width, height, accum_ptr, frame_ptr, stream=None):
        self._check(self._lib.mipt_unpermute(self._h, stream, gathered_ptr, world, width, height, accum_ptr, frame_ptr))

    def sync(self):
        self._check(self._lib.mipt_sync(self._h))

    def launch_stats(self):
        s = L.MiptLaunchStats()
        self._check(self._lib.mipt_get_launch_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_ if k != "reserved"}

    def launch_timings(self):
        t = L.MiptLaunchTimings()
        self._check(self._lib.mipt_get_launch_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    # ---- ray-buffer queries
    def trace_closest(self, handle, rays_ptr, n, hits_ptr, stream=None):
        self._check(self._lib.mipt_trace_closest(self._h, stream, handle, rays_ptr, n, hits_ptr))

    def trace_any(self, handle, rays_ptr, n, occluded_ptr, stream=None):
        self._check(self._lib.mipt_trace_any(self._h, stream, handle, rays_ptr, n, occluded_ptr))


class Group:
    """mipt_group_*: one host thread, n devices (SDK/optixMultiGPU's shape, optixMultiGPU.cpp:562-594).  contexts[i] is member
    i's Context (owned by the group): build the scene's acceleration structure and hit-group records on each of them."""

    def __init__(self, devices, log=None, log_level=0):
        self._lib = L.load()
        self._h = C.c_void_p()
        self._log_py = log

        def _cb(level, tag, msg, _):
            if self._log_py:
                self._log_py(level, tag.decode(), msg.decode())
        self._cb = L.LOG_CALLBACK(_cb)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.mipt_group_create(devs, len(devices), self._cb, None, log_level, C.byref(self._h))
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), "mipt_group_create(%s)" % list(devices))
        self.devices = [int(d) for d in devices]
        self.contexts, self.streams = [], []
        for i in range(len(devices)):
            ctx, dev, st = C.c_void_p(), C.c_int(), C.c_void_p()
            self._check(self._lib.mipt_group_member(self._h, i, C.byref(ctx), C.byref(dev), C.byref(st)))
            self.contexts.append(Context._borrow(ctx.value, dev.value))
            self.streams.append(st.value)

    def _check(self, rc):
        if rc:
            raise MiptError(rc, self._lib.mipt_error_name(rc).decode(), self._lib.mipt_group_last_error_string(self._h).decode())

    @property
    def uses_rccl(self):
        return bool(self._lib.mipt_group_uses_rccl(self._h))

    def launch(self, params_list, width, height):
        arr = (L.MiptParams * len(params_list))(*params_list)
        self._check(self._lib.mipt_group_launch(self._h, arr, width, height))

    def allgather(self, accum_ptrs, frame_ptrs=None):
        n = len(self.devices)
        acc = (C.c_void_p * n)(*[int(p) if p else None for p in accum_ptrs])
        frm = (C.c_void_p * n)(*[int(p) if p else None for p in frame_ptrs]) if frame_ptrs is not None else None
        self._check(self._lib.mipt_group_allgather(self._h, acc, frm))

    def sync(self):
        self._check(self._lib.mipt_group_sync(self._h))

    def destroy(self):
        if self._h:
            for c in self.contexts:
                c.destroy()
            self._lib.mipt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
