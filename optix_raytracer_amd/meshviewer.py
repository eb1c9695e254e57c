"""Host side of the glTF viewer above the C ABI, mirroring SDK/optixMeshViewer/optixMeshViewer.cpp
(same function names and order of calls; failures raise, like OPTIX_CHECK / CUDA_CHECK):

    sutil::loadScene + scene.finalize()  (main :445-447)      -> loadScene
    initCameraState :326                                       -> initCameraState
    initLaunchParams :190                                      -> initLaunchParams (two point lights, miss 0.1)
    handleCameraUpdate :235 / updateState :274                 -> handleCameraUpdate / updateState
    launchSubframe :284                                        -> launchSubframe

Scene::finalize (Scene.cpp:803-815: createContext, buildMeshAccels, buildInstanceAccel, createPTXModule,
createProgramGroups, createPipeline, createSBT) collapses to the accel build(s) plus mipt_whitted_set_geometry; the programs
are fixed-function HIP kernels inside libmipt.so.  loadScene(two_level=True) builds what the reference builds -- one
triangle structure per mesh (object space) and an instance structure over them (mipt_accel_build_instances); the default is
one BVH over the flattened, world-space triangles (1.5 x faster, twenty times the memory).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import api
from .camera import Camera
from .pathtracer import OutputBuffer  # noqa: F401  (the viewer uses the same CUDAOutputBuffer<uchar4>)

LIGHT_POINT, LIGHT_AMBIENT = 0, 1          # Light::Type (cuda/Light.h:37-41)
FALLOFF_NONE, FALLOFF_LINEAR, FALLOFF_QUADRATIC = 0, 1, 2


class MeshViewerState:
    def __init__(self):
        self.context = None
        self.scene = None            # api.Scene
        self.handle = 0
        self.d_vertices = None       # (n*3, 4) f32 world-space triangle soup
        self.d_normals = None        # (n*3, 4) f32 or None
        self.d_material_index = None
        self.d_texcoords = [None, None]   # (n*3, 2) f32 per TEXCOORD set
        self.d_texcoord_present = [None, None]
        self.d_colors = None         # (n*3, 4) f32 COLOR_0
        self.d_color_present = None
        self.textures = None         # (ctypes MiptTextureData array, count): texels live in the api.Scene
        self.materials = None        # ctypes array, host
        self.d_lights = None         # uint8 tensor holding MiptLight records
        self.lights = None           # host copy (ctypes array)
        self.params = L.MiptWhittedLaunchParams()
        self.accum = None
        self.camera = None
        self.camera_changed = True
        self.device = None
        self.stream = None
        self.width, self.height = 1024, 768       # optixMeshViewer.cpp:84-85


def loadScene(state, filename, device_index=0, log=None, log_level=0, two_level=False, instanced_shading=None):
    """sutil::loadScene( infile, scene ); scene.finalize();
    two_level: one triangle structure per mesh + the instance structure (what the reference builds); instanced_shading (default: with
    two_level): the shading tables per MESH in object space + instance records (Scene::createSBT) instead of flattened world-space arrays"""
    state.device = torch.device("cuda", device_index)
    torch.cuda.set_device(state.device)
    state.scene = api.Scene(filename)
    state.context = api.Context(device_index, log=log, log_level=log_level)
    instanced_shading = two_level if instanced_shading is None else (instanced_shading and two_level)
    if state.scene.info()["num_triangles"] == 0:
        raise api.MiptError(1, "MIPT_ERROR_INVALID_VALUE", "scene has no triangles")
    state.instanced_shading = False
    if not instanced_shading:
        verts, mat = state.scene.flatten()
        normals = state.scene.flatten_normals()
        state.d_vertices = torch.from_numpy(np.ascontiguousarray(verts)).to(state.device)
        state.d_normals = torch.from_numpy(np.ascontiguousarray(normals)).to(state.device) if np.any(normals[:, 3] != 0) else None
        state.d_material_index = torch.from_numpy(mat.astype(np.int32)).to(state.device)
        for j in range(2):
            uv, has = state.scene.flatten_texcoords(j)
            if has.any():
                state.d_texcoords[j] = torch.from_numpy(np.ascontiguousarray(uv)).to(state.device)
                state.d_texcoord_present[j] = torch.from_numpy(np.ascontiguousarray(has)).to(state.device)
        col, has = state.scene.flatten_colors()
        if has.any():
            state.d_colors = torch.from_numpy(np.ascontiguousarray(col)).to(state.device)
            state.d_color_present = torch.from_numpy(np.ascontiguousarray(has)).to(state.device)
    state.materials = state.scene.material_table()
    if two_level:
        # Scene::buildMeshAccels (one GAS per mesh, object space) + Scene::buildInstanceAccel (Scene.cpp:817-1212)
        from . import scenes
        meshes, insts = scenes.gltf_two_level(state.scene)
        state.mesh_handles, state.d_mesh_vertices = [], []
        for mv, mm in meshes:
            dv = torch.from_numpy(np.ascontiguousarray(mv)).to(state.device)
            state.d_mesh_vertices.append(dv)
            state.mesh_handles.append(state.context.accel_build([{
                "vertices": dv.data_ptr(), "num_vertices": mv.shape[0], "stride": 16,
                "sbt_index_offsets": np.minimum(mm, len(state.materials) - 1).astype(np.uint32), "flags": geometry_flags(state.materials)}]))
        state.instances = [(state.mesh_handles[m], xf) for m, xf in insts]
        state.handle = state.context.accel_build_instances(state.instances)
    else:
        state.handle = state.context.accel_build([{
            "vertices": state.d_vertices.data_ptr(), "num_vertices": verts.shape[0], "stride": 16,
            "sbt_index_offsets": np.minimum(mat, len(state.materials) - 1).astype(np.uint32), "flags": geometry_flags(state.materials),
        }])
    state.textures = state.scene.texture_table()
    ptr = lambda t: t.data_ptr() if t is not None else None            # noqa: E731
    if two_level and instanced_shading:
        # Scene::createSBT's shading data (Scene.cpp:1368-1480): one set of OBJECT-space buffers per mesh, one record per instance --
        # instancing then saves shading memory as well as BVH memory (the flattened arrays above are dropped)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(state.device)   # noqa: E731
        state.d_mesh_shading = []
        mesh_recs = (L.MiptWhittedMesh * len(meshes))()
        for m, (mv, mm) in enumerate(meshes):
            sh = state.scene.mesh_shading(m)
            keep = {"v": state.d_mesh_vertices[m], "mat": dev(mm.astype(np.int32)), "n": dev(sh["normals"]) if np.any(sh["normals"][:, 3] != 0) else None,
                    "uv": [None, None], "uvp": [None, None], "c": None, "cp": None}
            for j in range(2):
                uv, has = sh["texcoords"][j]
                if has.any():
                    keep["uv"][j], keep["uvp"][j] = dev(uv), dev(has)
            col, has = sh["colors"]
            if has.any():
                keep["c"], keep["cp"] = dev(col), dev(has)
            state.d_mesh_shading.append(keep)
            g = mesh_recs[m].geometry
            g.vertices4, g.normals4, g.material_index = ptr(keep["v"]), ptr(keep["n"]), ptr(keep["mat"])
            g.texcoords = (C.c_void_p * 2)(ptr(keep["uv"][0]), ptr(keep["uv"][1]))
            g.texcoord_present = (C.c_void_p * 2)(ptr(keep["uvp"][0]), ptr(keep["uvp"][1]))
            g.colors4, g.color_present = ptr(keep["c"]), ptr(keep["cp"])
            mesh_recs[m].num_triangles = mv.shape[0] // 3
        inst_recs = (L.MiptWhittedInstance * len(insts))()
        for i, (m, xf) in enumerate(insts):
            inst_recs[i].mesh = m
            inst_recs[i].transform = (C.c_float * 12)(*[float(x) for x in np.asarray(xf, np.float32).reshape(-1)])
        state.context._check(state.context._lib.mipt_whitted_set_geometry_instanced(
            state.context._h, mesh_recs, len(meshes), inst_recs, len(insts), state.materials, len(state.materials), state.textures[0], state.textures[1]))
        state.instanced_shading = True
        return
    geo = L.MiptWhittedGeometry()
    geo.vertices4, geo.normals4, geo.material_index = ptr(state.d_vertices), ptr(state.d_normals), ptr(state.d_material_index)
    geo.texcoords = (C.c_void_p * 2)(ptr(state.d_texcoords[0]), ptr(state.d_texcoords[1]))
    geo.texcoord_present = (C.c_void_p * 2)(ptr(state.d_texcoord_present[0]), ptr(state.d_texcoord_present[1]))
    geo.colors4, geo.color_present = ptr(state.d_colors), ptr(state.d_color_present)
    state.context._check(state.context._lib.mipt_whitted_set_geometry(
        state.context._h, C.byref(geo), state.materials, len(state.materials), state.textures[0], state.textures[1]))


def geometry_flags(materials):
    """OptixGeometryFlags per material as Scene::buildMeshAccels picks them (Scene.cpp:904-966): opaque -> DISABLE_ANYHIT,
    mask -> NONE, blend -> REQUIRE_SINGLE_ANYHIT_CALL, each | DISABLE_TRIANGLE_FACE_CULLING when doubleSided; the last
    table entry (primitives without a material) is opaque and single sided"""
    out = []
    for m in materials:
        base = {0: L.GEOMETRY_FLAG_DISABLE_ANYHIT, 1: L.GEOMETRY_FLAG_NONE, 2: L.GEOMETRY_FLAG_REQUIRE_SINGLE_ANYHIT_CALL}[int(m.alpha_mode)]
        out.append(base | (L.GEOMETRY_FLAG_DISABLE_TRIANGLE_FACE_CULLING if m.double_sided else 0))
    return np.asarray(out, np.uint32)


def initCameraState(state):
    cam = state.scene.camera()
    state.camera = Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], cam["aspect"])
    state.camera_changed = True


def make_lights(records):
    """records: iterable of dicts(type, color, intensity, position, falloff) -> ctypes MiptLight array"""
    records = list(records)
    arr = (L.MiptLight * max(len(records), 1))()
    for k, r in enumerate(records):
        arr[k].type = int(r.get("type", LIGHT_POINT))
        arr[k].color = (C.c_float * 3)(*[float(x) for x in r["color"]])
        arr[k].intensity = float(r.get("intensity", 0.0))
        arr[k].position = (C.c_float * 3)(*[float(x) for x in r.get("position", (0, 0, 0))])
        arr[k].falloff = int(r.get("falloff", FALLOFF_QUADRATIC))
    return arr, len(records)


def default_lights(aabb_min, aabb_max):
    """the two point lights of initLaunchParams (:199-211), fp32 arithmetic as written there"""
    lo, hi = np.asarray(aabb_min, np.float32), np.asarray(aabb_max, np.float32)
    ext = hi - lo
    loffset = np.float32(ext[int(np.argmax(ext))])                    # Aabb::maxExtent
    center = (lo + hi) * np.float32(0.5)                              # Aabb::center
    return [
        {"type": LIGHT_POINT, "color": (1.0, 1.0, 0.8), "intensity": 5.0,
         "position": center + np.array([loffset] * 3, np.float32), "falloff": FALLOFF_QUADRATIC},
        {"type": LIGHT_POINT, "color": (0.8, 0.8, 1.0), "intensity": 3.0,
         "position": center + np.array([-loffset, np.float32(0.5) * loffset, np.float32(-0.5) * loffset], np.float32),
         "falloff": FALLOFF_QUADRATIC},
    ]


def initLaunchParams(state, width=None, height=None, lights=None):
    if width:
        state.width, state.height = int(width), int(height)
    p = state.params
    p.width, p.height = state.width, state.height
    state.accum = torch.zeros((state.height, state.width, 4), dtype=torch.float32, device=state.device)
    p.accum_buffer = state.accum.data_ptr()
    p.frame_buffer = None
    p.subframe_index = 0
    info = state.scene.info()
    recs = default_lights(info["aabb_min"], info["aabb_max"]) if lights is None else lights
    state.lights, n = make_lights(recs)
    raw = np.frombuffer(bytes(state.lights), np.uint8).copy()
    state.d_lights = torch.from_numpy(raw).to(state.device)
    p.lights.data = state.d_lights.data_ptr()
    p.lights.count = n
    p.lights.byte_stride = 0
    p.lights.elmt_byte_size = C.sizeof(L.MiptLight)
    p.miss_color = (C.c_float * 3)(0.1, 0.1, 0.1)
    p.handle = state.handle
    state.stream = torch.cuda.Stream(device=state.device)
    state.stream.wait_stream(torch.cuda.current_stream(state.device))      # the buffers above were cleared on torch's current stream


def handleCameraUpdate(state):
    if not state.camera_changed:
        return
    state.camera_changed = False
    p = state.params
    state.camera.setAspectRatio(np.float32(state.width) / np.float32(state.height))
    p.eye = (C.c_float * 3)(*[float(x) for x in state.camera.eye()])
    U, V, W = state.camera.UVWFrame()
    p.U = (C.c_float * 3)(*U.tolist())
    p.V = (C.c_float * 3)(*V.tolist())
    p.W = (C.c_float * 3)(*W.tolist())


def updateState(state):
    if state.camera_changed:
        state.params.subframe_index = 0
    handleCameraUpdate(state)


def launchSubframe(output_buffer, state):
    state.params.frame_buffer = output_buffer.map()
    output_buffer.setStream(state.stream)
    state.context._check(state.context._lib.mipt_whitted_launch(
        state.context._h, state.stream.cuda_stream, C.byref(state.params), state.width, state.height, 1))
    output_buffer.unmap()
    state.context.sync()


def cleanup(state):
    if state.context:
        state.context.destroy()
        state.context = None
    state.d_vertices = state.d_normals = state.d_material_index = state.d_lights = state.accum = None
    state.d_texcoords = [None, None]; state.d_texcoord_present = [None, None]; state.d_colors = state.d_color_present = None


def render(filename, width, height, subframes=1, lights=None, device_index=0, two_level=False, instanced_shading=None):
    """main() of the sample in file-output mode (:466-485): returns (state, uchar4 image as numpy)"""
    st = MeshViewerState()
    loadScene(st, filename, device_index, two_level=two_level, instanced_shading=instanced_shading)
    initCameraState(st)
    initLaunchParams(st, width, height, lights)
    out = OutputBuffer(width, height, st.device)
    for _ in range(subframes):
        updateState(st)
        launchSubframe(out, st)
        st.params.subframe_index += 1
    return st, out.getHostPointer()
