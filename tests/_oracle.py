"""ctypes binding of oracle/liboracle.so -- the CPU ORACLE (checker only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It never touches /root/reference at run time.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

MODE_BRUTE, MODE_BVH, MODE_INST = 0, 1, 2


class OrcLight(C.Structure):
    _fields_ = [("corner", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3),
                ("normal", C.c_float * 3), ("emission", C.c_float * 3)]


class OrcParams(C.Structure):
    _fields_ = [("subframe_index", C.c_uint32), ("accum_buffer", C.c_void_p), ("frame_buffer", C.c_void_p),
                ("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_launch", C.c_uint32),
                ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
                ("light", OrcLight), ("handle", C.c_uint64)]


class OrcStats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("max_depth_seen", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class OrcLightW(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("intensity", C.c_float), ("position", C.c_float * 3),
                ("falloff", C.c_int32)]


class OrcTexInfoW(C.Structure):
    _fields_ = [("texture", C.c_int32), ("texcoord", C.c_int32), ("offset", C.c_float * 2), ("rotation", C.c_float * 2),
                ("scale", C.c_float * 2)]


class OrcMaterialW(C.Structure):
    _fields_ = [("base_color", C.c_float * 4), ("metallic", C.c_float), ("roughness", C.c_float),
                ("emissive_factor", C.c_float * 3), ("alpha_cutoff", C.c_float), ("alpha_mode", C.c_int32),
                ("double_sided", C.c_int32), ("base_color_tex", OrcTexInfoW), ("metallic_roughness_tex", OrcTexInfoW),
                ("normal_tex", OrcTexInfoW), ("emissive_tex", OrcTexInfoW)]


class OrcTextureW(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bits", C.c_uint32), ("address_s", C.c_uint32),
                ("address_t", C.c_uint32), ("filter", C.c_uint32), ("image", C.c_int32), ("texels", C.c_void_p)]


class OrcWhittedFrame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("subframe_index", C.c_uint32),
                ("accum_buffer", C.c_void_p), ("frame_buffer", C.c_void_p),
                ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
                ("lights", C.POINTER(OrcLightW)), ("nlights", C.c_uint32), ("miss_color", C.c_float * 3),
                ("normals4", C.c_void_p), ("material_index", C.c_void_p),
                ("materials", C.POINTER(OrcMaterialW)), ("nmaterials", C.c_uint32),
                ("texcoords", C.c_void_p * 2), ("texcoord_present", C.c_void_p * 2),
                ("colors4", C.c_void_p), ("color_present", C.c_void_p),
                ("textures", C.POINTER(OrcTextureW)), ("ntextures", C.c_uint32)]


class OrcImguiCamera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3), ("ortho", C.c_uint8),
                ("pad_", C.c_uint8 * 3), ("fov", C.c_float), ("fd", C.c_float), ("aperture", C.c_float),
                ("speed", C.c_float), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3)]


class OrcImguiLight(C.Structure):
    _fields_ = [("a", C.c_float * 3), ("b", C.c_float * 4), ("c", C.c_float * 3), ("tag", C.c_int32)]


class OrcImguiFrame(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("samples_per_frame", C.c_uint32),
                ("dt", C.c_uint32), ("dirty", C.c_int32), ("camera", C.POINTER(OrcImguiCamera)),
                ("lights", C.POINTER(OrcImguiLight)), ("nlights", C.c_int32), ("materials", C.POINTER(C.c_float)),
                ("nmaterials", C.c_int32), ("mat_indices", C.POINTER(C.c_int32)), ("normals", C.POINTER(C.c_float)),
                ("film", C.POINTER(C.c_float)), ("image", C.POINTER(C.c_uint8))]


def _build():
    src = os.path.join(ORACLE_DIR, "pt_oracle.c")
    if os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ORACLE_DIR, "pt_oracle.h"))):
        return
    subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _build()
    L = C.CDLL(LIB_PATH)
    fp, u32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    L.orc_tea4.restype = C.c_uint32
    L.orc_tea4.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_lcg.restype = C.c_uint32
    L.orc_lcg.argtypes = [u32p]
    L.orc_rnd.restype = C.c_float
    L.orc_rnd.argtypes = [u32p]
    L.orc_camera_uvw.restype = None
    L.orc_camera_uvw.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, fp, fp, fp]
    L.orc_make_color.restype = None
    L.orc_make_color.argtypes = [fp, u8p]
    L.orc_vec_kat.restype = None
    L.orc_vec_kat.argtypes = [fp, fp, C.c_float, fp]
    L.orc_det_sincos.restype = None
    L.orc_det_sincos.argtypes = [C.c_float, fp, fp]
    L.orc_det_pow_inv24.restype = C.c_float
    L.orc_det_pow_inv24.argtypes = [C.c_float]
    L.orc_cosine_sample_hemisphere.restype = None
    L.orc_cosine_sample_hemisphere.argtypes = [C.c_float, C.c_float, fp]
    L.orc_onb_inverse_transform.restype = None
    L.orc_onb_inverse_transform.argtypes = [fp, fp]
    L.orc_work_pixel.restype = None
    L.orc_work_pixel.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)] * 2
    L.orc_work_num_samples.restype = C.c_int
    L.orc_work_num_samples.argtypes = [C.c_int] * 3
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_scene_destroy.restype = None
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_scene_set_prim_flags.restype = None
    L.orc_scene_set_prim_flags.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_scene_build_bvh.restype = C.c_int
    L.orc_scene_build_bvh.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_scene_set_bvh.restype = C.c_int
    L.orc_scene_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.orc_scene_set_instances.restype = C.c_int
    L.orc_scene_set_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
    L.orc_scene_num_nodes.restype = C.c_uint32
    L.orc_scene_num_nodes.argtypes = [C.c_void_p]
    L.orc_scene_bvh_depth.restype = C.c_int
    L.orc_scene_bvh_depth.argtypes = [C.c_void_p]
    L.orc_scene_validate_bvh.restype = C.c_int
    L.orc_scene_validate_bvh.argtypes = [C.c_void_p]
    L.orc_trace_closest.restype = None
    L.orc_trace_closest.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.POINTER(OrcStats)]
    L.orc_trace_any.restype = None
    L.orc_trace_any.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(OrcStats)]
    L.orc_render_rows.restype = C.c_int
    L.orc_render_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcParams), C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.POINTER(OrcStats)]
    L.orc_render_pixels.restype = C.c_int
    L.orc_render_pixels.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcParams), C.c_uint32, C.c_void_p, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.POINTER(OrcStats)]
    L.orc_imgui_camera_compute_uvw.restype = None
    L.orc_imgui_camera_compute_uvw.argtypes = [C.POINTER(OrcImguiCamera)]
    L.orc_imgui_camera_set_fd.restype = None
    L.orc_imgui_camera_set_fd.argtypes = [C.POINTER(OrcImguiCamera), C.c_float]
    L.orc_imgui_compute_ray.restype = None
    L.orc_imgui_compute_ray.argtypes = [C.POINTER(OrcImguiCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp, fp]
    L.orc_imgui_light_wi.restype = None
    L.orc_imgui_light_wi.argtypes = [C.POINTER(OrcImguiLight), fp, u32p, fp, fp]
    L.orc_imgui_render_rows.restype = C.c_int
    L.orc_imgui_render_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcImguiFrame), C.c_uint32, C.c_uint32,
                                        C.POINTER(OrcStats)]
    L.orc_whitted_render_rows.restype = C.c_int
    L.orc_whitted_render_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcWhittedFrame), C.c_uint32, C.c_uint32,
                                          C.POINTER(OrcStats)]
    L.orc_sample_texture.restype = None
    L.orc_sample_texture.argtypes = [C.POINTER(OrcTextureW), C.c_float, C.c_float, C.POINTER(C.c_float * 4)]
    L.orc_det_pow22.restype = C.c_float
    L.orc_det_pow22.argtypes = [C.c_float]
    L.orc_primary_rays.restype = None
    L.orc_primary_rays.argtypes = [C.POINTER(OrcParams), C.c_uint32, C.c_uint32, C.c_void_p]
    _lib = L
    return L


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_uvw(eye, lookat, up, fov_y, aspect, fod=1.0):
    U, V, W = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    lib().orc_camera_uvw(_f3(eye), _f3(lookat), _f3(up), fov_y, aspect, fod, U, V, W)
    return (np.array(U, dtype=np.float32), np.array(V, dtype=np.float32), np.array(W, dtype=np.float32))


class Scene:
    """Owns an OrcScene.  `scene` is a dict as produced by optix_raytracer_amd.scenes.* :
    vertices (n*3,4) f32, mat_index (n,) u32, emission (m,3), diffuse (m,3), bg (4,)"""

    def __init__(self, scene):
        self.verts = np.ascontiguousarray(scene["vertices"], dtype=np.float32).reshape(-1, 4)
        assert self.verts.shape[0] % 3 == 0
        self.n_tris = self.verts.shape[0] // 3
        mat = np.ascontiguousarray(scene["mat_index"], dtype=np.uint32)
        em = np.ascontiguousarray(scene["emission"], dtype=np.float32).reshape(-1, 3)
        df = np.ascontiguousarray(scene["diffuse"], dtype=np.float32).reshape(-1, 3)
        bg = np.ascontiguousarray(scene.get("bg", np.zeros(4)), dtype=np.float32)
        self._h = lib().orc_scene_create(self.verts.ctypes.data, self.n_tris, mat.ctypes.data, em.ctypes.data,
                                         df.ctypes.data, em.shape[0], bg.ctypes.data)
        self.has_bvh = False

    def __del__(self):
        try:
            if self._h:
                lib().orc_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_prim_flags(self, flags):
        """OptixGeometryFlags per primitive (None clears): 1 DISABLE_ANYHIT, 4 DISABLE_TRIANGLE_FACE_CULLING"""
        if flags is None:
            lib().orc_scene_set_prim_flags(self._h, None)
        else:
            f = np.ascontiguousarray(flags, np.uint32)
            assert f.shape[0] == self.n_tris
            lib().orc_scene_set_prim_flags(self._h, f.ctypes.data)
        return self

    def build_bvh(self, leaf_max=4, max_depth=48):
        rc = lib().orc_scene_build_bvh(self._h, leaf_max, max_depth)
        if rc:
            raise RuntimeError("orc_scene_build_bvh failed: %d" % rc)
        self.has_bvh = True
        return self

    def set_bvh(self, nodes, tris):
        nodes = np.ascontiguousarray(nodes)
        tris = np.ascontiguousarray(tris)
        assert nodes.nbytes % 64 == 0 and tris.nbytes % 48 == 0
        rc = lib().orc_scene_set_bvh(self._h, nodes.ctypes.data, nodes.nbytes // 64, tris.ctypes.data,
                                     tris.nbytes // 48)
        if rc:
            raise RuntimeError("orc_scene_set_bvh: BVH failed validation: %d" % rc)
        self.has_bvh = True
        return self

    def set_instances(self, tlas_nodes, order, inv, prim_base, gas):
        """gas: [(nodes, tris)] per instance, the arrays of Context.accel_export.  Render with mode=MODE_INST afterwards."""
        tlas = np.ascontiguousarray(tlas_nodes, np.float32)
        order, base = np.ascontiguousarray(order, np.uint32), np.ascontiguousarray(prim_base, np.uint32)
        inv = np.ascontiguousarray(inv, np.float32)
        k = order.shape[0]
        keep = [(np.ascontiguousarray(n), np.ascontiguousarray(t)) for n, t in gas]
        gn = (C.c_void_p * k)(*[n.ctypes.data for n, _ in keep])
        gt = (C.c_void_p * k)(*[t.ctypes.data for _, t in keep])
        nn = np.asarray([n.nbytes // 64 for n, _ in keep], np.uint32)
        nt = np.asarray([t.nbytes // 48 for _, t in keep], np.uint32)
        rc = lib().orc_scene_set_instances(self._h, tlas.ctypes.data, tlas.nbytes // 64, order.ctypes.data, k, inv.ctypes.data,
                                           base.ctypes.data, gn, nn.ctypes.data, gt, nt.ctypes.data)
        if rc:
            raise RuntimeError("orc_scene_set_instances: %d" % rc)
        self.has_bvh = True
        return self

    def validate(self):
        return lib().orc_scene_validate_bvh(self._h)

    @property
    def num_nodes(self):
        return lib().orc_scene_num_nodes(self._h)

    @property
    def depth(self):
        return lib().orc_scene_bvh_depth(self._h)

    def trace_closest(self, rays, mode=MODE_BVH, stats=None):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        lib().orc_trace_closest(self._h, mode, rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data, u.ctypes.data,
                                v.ctypes.data, C.byref(stats) if stats is not None else None)
        return t, prim, u, v

    def trace_any(self, rays, mode=MODE_BVH, stats=None):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        occ = np.empty(n, np.uint8)
        lib().orc_trace_any(self._h, mode, rays.ctypes.data, n, occ.ctypes.data,
                            C.byref(stats) if stats is not None else None)
        return occ

    def render_whitted(self, launch, lights, materials, material_index, normals4=None, accum=None, mode=MODE_BVH,
                       texcoords=(None, None), texcoord_present=(None, None), colors4=None, color_present=None, textures=()):
        """SDK/cuda/whitted.cu restatement.  launch: dict(width,height,subframe,eye,U,V,W,miss_color);
        lights: list of dict(type,color,intensity,position); materials: list of dict(base_color,metallic,roughness,
        emissive_factor[, <name>_tex: dict(texture,texcoord,offset,rotation,scale)]); textures: list of dict(texels
        (h,w,4) uint8/uint16 or None, address_s, address_t, filter).  Returns (accum HxWx4 f32, frame HxWx4 u8, stats)."""
        w, h = int(launch["width"]), int(launch["height"])
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        frame = np.zeros((h, w, 4), np.uint8)
        f = OrcWhittedFrame()
        f.width, f.height, f.subframe_index = w, h, int(launch.get("subframe", 0))
        f.accum_buffer, f.frame_buffer = accum.ctypes.data, frame.ctypes.data
        for k in ("eye", "U", "V", "W"):
            setattr(f, k, _f3(launch[k]))
        f.miss_color = _f3(launch.get("miss_color", (0.1, 0.1, 0.1)))
        la = (OrcLightW * max(len(lights), 1))()
        for k, r in enumerate(lights):
            la[k].type = int(r.get("type", 0))
            la[k].color = _f3(r["color"])
            la[k].intensity = float(r.get("intensity", 0.0))
            la[k].position = _f3(r.get("position", (0, 0, 0)))
            la[k].falloff = int(r.get("falloff", 2))
        f.lights, f.nlights = la, len(lights)
        ma = (OrcMaterialW * len(materials))()
        for k, m in enumerate(materials):
            ma[k].base_color = (C.c_float * 4)(*[float(x) for x in m["base_color"]])
            ma[k].metallic, ma[k].roughness = float(m["metallic"]), float(m["roughness"])
            ma[k].emissive_factor = _f3(m["emissive_factor"])
            ma[k].alpha_cutoff = float(m.get("alpha_cutoff", 0.0))
            ma[k].alpha_mode, ma[k].double_sided = int(m.get("alpha_mode", 0)), int(m.get("double_sided", 0))
            for name in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emissive_tex"):
                ti, src = getattr(ma[k], name), m.get(name)
                ti.texture, ti.texcoord = -1, 0
                ti.offset, ti.rotation, ti.scale = (C.c_float * 2)(0, 0), (C.c_float * 2)(0, 1), (C.c_float * 2)(1, 1)
                if src is not None:
                    ti.texture, ti.texcoord = int(src["texture"]), int(src.get("texcoord", 0))
                    ti.offset = (C.c_float * 2)(*[float(x) for x in src.get("offset", (0, 0))])
                    ti.rotation = (C.c_float * 2)(*[float(x) for x in src.get("rotation", (0, 1))])
                    ti.scale = (C.c_float * 2)(*[float(x) for x in src.get("scale", (1, 1))])
        f.materials, f.nmaterials = ma, len(materials)
        keep = []
        mi = np.ascontiguousarray(material_index, np.uint32)
        f.material_index = mi.ctypes.data
        if normals4 is not None:
            normals4 = np.ascontiguousarray(normals4, np.float32)
            f.normals4 = normals4.ctypes.data
        for j in range(2):
            if texcoords[j] is not None:
                a = np.ascontiguousarray(texcoords[j], np.float32); keep.append(a)
                f.texcoords[j] = a.ctypes.data
                if texcoord_present[j] is not None:
                    b = np.ascontiguousarray(texcoord_present[j], np.uint8); keep.append(b)
                    f.texcoord_present[j] = b.ctypes.data
        if colors4 is not None:
            a = np.ascontiguousarray(colors4, np.float32); keep.append(a)
            f.colors4 = a.ctypes.data
            if color_present is not None:
                b = np.ascontiguousarray(color_present, np.uint8); keep.append(b)
                f.color_present = b.ctypes.data
        ta = (OrcTextureW * max(len(textures), 1))()
        for k, t in enumerate(textures):
            ta[k].address_s, ta[k].address_t, ta[k].filter = int(t.get("address_s", 0)), int(t.get("address_t", 0)), int(t.get("filter", 1))
            if t.get("texels") is not None:
                tx = np.ascontiguousarray(t["texels"]); keep.append(tx)
                ta[k].height, ta[k].width = tx.shape[0], tx.shape[1]
                ta[k].bits = 16 if tx.dtype == np.uint16 else 8
                ta[k].texels = tx.ctypes.data
        f.textures, f.ntextures = ta, len(textures)
        st = OrcStats()
        rc = lib().orc_whitted_render_rows(self._h, mode, C.byref(f), 0, h, C.byref(st))
        if rc:
            raise RuntimeError("orc_whitted_render_rows: %d" % rc)
        return accum, frame, st.as_dict()

    def make_params(self, launch, accum=None, frame=None):
        """launch: dict with width,height,spp,subframe,eye,U,V,W,light{corner,v1,v2,normal,emission}"""
        p = OrcParams()
        p.subframe_index = int(launch.get("subframe", 0))
        p.width, p.height = int(launch["width"]), int(launch["height"])
        p.samples_per_launch = int(launch.get("spp", 1))
        for k in ("eye", "U", "V", "W"):
            setattr(p, k, _f3(launch[k]))
        for k in ("corner", "v1", "v2", "normal", "emission"):
            setattr(p.light, k, _f3(launch["light"][k]))
        if accum is not None:
            p.accum_buffer = accum.ctypes.data
        if frame is not None:
            p.frame_buffer = frame.ctypes.data
        return p

    def render(self, launch, mode=MODE_BVH, max_depth=0, accum=None, threads=1, rows=None, want_frame=True):
        """Renders rows [y0,y1) (default: all).  Returns (accum HxWx4 f32, frame HxWx4 u8, stats dict)."""
        w, h = int(launch["width"]), int(launch["height"])
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        frame = np.zeros((h, w, 4), np.uint8) if want_frame else None
        p = self.make_params(launch, accum, frame)
        y0, y1 = rows if rows is not None else (0, h)
        L = lib()
        if threads <= 1:
            st = OrcStats()
            rc = L.orc_render_rows(self._h, mode, C.byref(p), max_depth, y0, y1, C.byref(st))
            if rc:
                raise RuntimeError("orc_render_rows: %d" % rc)
            return accum, frame, st.as_dict()
        import threading
        # interleaved small bands so every thread sees a similar mix of the image
        band = 4
        bands = [(y, min(y + band, y1)) for y in range(y0, y1, band)]
        stats = [OrcStats() for _ in range(threads)]
        errs = []

        def work(tid):
            for b in bands[tid::threads]:
                rc = L.orc_render_rows(self._h, mode, C.byref(p), max_depth, b[0], b[1], C.byref(stats[tid]))
                if rc:
                    errs.append(rc)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if errs:
            raise RuntimeError("orc_render_rows: %s" % errs)
        tot = {k: 0 for k, _ in OrcStats._fields_}
        for s in stats:
            for k, v in s.as_dict().items():
                tot[k] = max(tot[k], v) if k == "max_depth_seen" else tot[k] + v
        return accum, frame, tot

    def render_pixels(self, launch, pixel_index, mode=MODE_BVH, max_depth=0, prev=None):
        pix = np.ascontiguousarray(pixel_index, dtype=np.uint32)
        out = np.zeros((pix.shape[0], 4), np.float32)
        p = self.make_params(launch)
        st = OrcStats()
        prevp = np.ascontiguousarray(prev, dtype=np.float32).ctypes.data if prev is not None else None
        rc = lib().orc_render_pixels(self._h, mode, C.byref(p), max_depth, pix.ctypes.data, pix.shape[0],
                                     out.ctypes.data, prevp, C.byref(st))
        if rc:
            raise RuntimeError("orc_render_pixels: %d" % rc)
        return out, st.as_dict()

    def primary_rays(self, launch, rows=None):
        w, h = int(launch["width"]), int(launch["height"])
        y0, y1 = rows if rows is not None else (0, h)
        rays = np.empty(((y1 - y0) * w, 8), np.float32)
        p = self.make_params(launch)
        lib().orc_primary_rays(C.byref(p), y0, y1, rays.ctypes.data)
        return rays
