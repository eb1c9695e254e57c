"""ctypes binding of libmipt.so -- the C ABI declared in include/mipt.h.

The library must exist (built in-tree by `__graft_entry__.build()` or
`make -C optix_raytracer_amd/csrc`); there is no fallback path: a missing
library raises ImportError-like RuntimeError on first use.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIPT_LIBRARY: an experiment build of the library (tools/probe_bound.sh), always under optix_raytracer_amd/lib/
LIB_PATH = os.path.join(_HERE, "lib", os.path.basename(os.environ.get("MIPT_LIBRARY", "libmipt.so")))


class MiptParallelogramLight(C.Structure):
    _fields_ = [("corner", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3),
                ("normal", C.c_float * 3), ("emission", C.c_float * 3)]


class MiptParams(C.Structure):
    """`struct Params`, SDK/optixPathTracer/optixPathTracer.h:91-107 (152 bytes)."""
    _fields_ = [("subframe_index", C.c_uint), ("accum_buffer", C.c_void_p), ("frame_buffer", C.c_void_p),
                ("width", C.c_uint), ("height", C.c_uint), ("samples_per_launch", C.c_uint),
                ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
                ("light", MiptParallelogramLight), ("handle", C.c_ulonglong)]


class MiptMissData(C.Structure):
    _fields_ = [("bg_color", C.c_float * 4)]


class MiptHitGroupData(C.Structure):
    _fields_ = [("emission_color", C.c_float * 3), ("diffuse_color", C.c_float * 3), ("vertices", C.c_void_p)]


class MiptAccelInfo(C.Structure):
    _fields_ = [("num_nodes", C.c_uint32), ("num_triangles", C.c_uint32), ("depth", C.c_uint32),
                ("leaf_max", C.c_uint32), ("device_bytes", C.c_uint64), ("build_seconds", C.c_double),
                ("node_format", C.c_uint32), ("num_instances", C.c_uint32), ("stack_entries", C.c_uint32), ("mirror_nodes", C.c_uint32)]


class MiptInstance(C.Structure):
    """OptixInstance (include/optix_types.h), 80 bytes"""
    _fields_ = [("transform", C.c_float * 12), ("instanceId", C.c_uint), ("sbtOffset", C.c_uint), ("visibilityMask", C.c_uint),
                ("flags", C.c_uint), ("traversableHandle", C.c_ulonglong), ("pad", C.c_uint * 2)]


class MiptBuildInput(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("num_vertices", C.c_uint32), ("vertex_stride_bytes", C.c_uint32),
                ("indices", C.c_void_p), ("index_format", C.c_uint32), ("index_stride_bytes", C.c_uint32),
                ("num_triangles", C.c_uint32), ("sbt_index_offsets", C.c_void_p), ("transform3x4", C.c_void_p),
                ("flags", C.c_void_p), ("num_sbt_records", C.c_uint32)]


class MiptSceneInfo(C.Structure):
    _fields_ = [("num_buffers", C.c_uint32), ("num_images", C.c_uint32), ("num_textures", C.c_uint32),
                ("num_materials", C.c_uint32), ("num_meshes", C.c_uint32), ("num_instances", C.c_uint32),
                ("num_cameras", C.c_uint32), ("num_triangles", C.c_uint32),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3)]


class MiptTextureInfo(C.Structure):
    """MaterialData::Texture (cuda/MaterialData.h:60-72), 32 bytes"""
    _fields_ = [("texture", C.c_int32), ("texcoord", C.c_int32), ("offset", C.c_float * 2), ("rotation", C.c_float * 2),
                ("scale", C.c_float * 2)]


class MiptMaterialData(C.Structure):
    _fields_ = [("base_color", C.c_float * 4), ("metallic", C.c_float), ("roughness", C.c_float),
                ("emissive_factor", C.c_float * 3), ("alpha_cutoff", C.c_float), ("alpha_mode", C.c_int32),
                ("double_sided", C.c_int32), ("base_color_tex", MiptTextureInfo), ("metallic_roughness_tex", MiptTextureInfo),
                ("normal_tex", MiptTextureInfo), ("emissive_tex", MiptTextureInfo)]


class MiptTextureData(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bits", C.c_uint32), ("address_s", C.c_uint32),
                ("address_t", C.c_uint32), ("filter", C.c_uint32), ("image", C.c_int32), ("texels", C.c_void_p)]


class MiptBufferView(C.Structure):
    """BufferView<T> (SDK/cuda/BufferView.h:33-49), 16 bytes"""
    _fields_ = [("data", C.c_ulonglong), ("count", C.c_uint), ("byte_stride", C.c_ushort), ("elmt_byte_size", C.c_ushort)]


class MiptWhittedLaunchParams(C.Structure):
    """whitted::LaunchParams (SDK/cuda/whitted.h:59-77), 128 bytes"""
    _fields_ = [("width", C.c_uint), ("height", C.c_uint), ("subframe_index", C.c_uint), ("accum_buffer", C.c_void_p),
                ("frame_buffer", C.c_void_p), ("max_depth", C.c_int), ("scene_epsilon", C.c_float),
                ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
                ("lights", MiptBufferView), ("miss_color", C.c_float * 3), ("handle", C.c_ulonglong)]


class MiptLight(C.Structure):
    """struct Light (SDK/cuda/Light.h:33-71), 36 bytes: type 0 POINT, 1 AMBIENT"""
    _fields_ = [("type", C.c_int), ("color", C.c_float * 3), ("intensity", C.c_float), ("position", C.c_float * 3),
                ("falloff", C.c_int)]


class MiptWhittedGeometry(C.Structure):
    _fields_ = [("vertices4", C.c_void_p), ("normals4", C.c_void_p), ("material_index", C.c_void_p),
                ("texcoords", C.c_void_p * 2), ("texcoord_present", C.c_void_p * 2), ("colors4", C.c_void_p),
                ("color_present", C.c_void_p)]


class MiptWhittedMesh(C.Structure):
    """per-mesh OBJECT-space shading tables of mipt_whitted_set_geometry_instanced"""
    _fields_ = [("geometry", MiptWhittedGeometry), ("num_triangles", C.c_uint32), ("reserved", C.c_uint32)]


class MiptWhittedInstance(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("transform", C.c_float * 12)]


class MiptImguiCamera(C.Structure):
    """class Camera of SDK/imgui_test (camera.h:155-173), 92 bytes"""
    _fields_ = [("eye", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3), ("ortho", C.c_uint8),
                ("pad_", C.c_uint8 * 3), ("fov", C.c_float), ("fd", C.c_float), ("aperture", C.c_float),
                ("speed", C.c_float), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3)]


class MiptImguiLight(C.Structure):
    """class LightVariant (light.h:10-51), 44 bytes"""
    _fields_ = [("a", C.c_float * 3), ("b", C.c_float * 4), ("c", C.c_float * 3), ("tag", C.c_int32)]


class MiptImguiParams(C.Structure):
    """struct Params of SDK/imgui_test (optixTriangle.h:42-108), 128 bytes"""
    _fields_ = [("image_width", C.c_uint), ("image_height", C.c_uint), ("samples_per_frame", C.c_uint),
                ("camera", C.c_void_p), ("dt", C.c_uint), ("dirty", C.c_uint8), ("image", C.c_void_p),
                ("film", C.c_void_p), ("tfactor", C.c_float), ("handle", C.c_ulonglong), ("normals", C.c_void_p),
                ("vertices", C.c_void_p), ("mat_indices", C.c_void_p), ("nmat_indices", C.c_int),
                ("lights", C.c_void_p), ("nlights", C.c_int), ("materials", C.c_void_p), ("nmaterials", C.c_int)]


class MiptLaunchStats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64),
                ("bounces", C.c_uint32), ("reserved", C.c_uint32)]


class MiptLaunchTimings(C.Structure):
    _fields_ = [("ms_raygen", C.c_float), ("ms_trace_closest", C.c_float), ("ms_shade", C.c_float),
                ("ms_trace_any", C.c_float), ("ms_resolve", C.c_float),
                ("n_trace_closest", C.c_uint32), ("n_trace_any", C.c_uint32)]


LOG_CALLBACK = C.CFUNCTYPE(None, C.c_uint, C.c_char_p, C.c_char_p, C.c_void_p)

IMAGE_UNSIGNED_BYTE4, IMAGE_FLOAT4, IMAGE_FLOAT3 = 0, 1, 2
GEOMETRY_FLAG_NONE, GEOMETRY_FLAG_DISABLE_ANYHIT, GEOMETRY_FLAG_REQUIRE_SINGLE_ANYHIT_CALL = 0, 1, 2
GEOMETRY_FLAG_DISABLE_TRIANGLE_FACE_CULLING = 4
RAY_FLAG_NONE, RAY_FLAG_CULL_BACK_FACING_TRIANGLES = 0, 16
OPT_MAX_DEPTH, OPT_COUNT_VISITS, OPT_TRACE_GRID, OPT_PERSISTENT, OPT_PROFILE = 1, 2, 3, 4, 5
OPT_REFILL_MIN, OPT_STEAL_MIN, OPT_BURST, OPT_BATCH_SAMPLES, OPT_BUILD, OPT_TAIL_PATHS, OPT_NODE_FORMAT, OPT_TILE_ORDER = 7, 8, 9, 10, 12, 13, 14, 15

# every symbol include/mipt.h declares: (name, restype, argtypes)
_vp, _u32, _u64 = C.c_void_p, C.c_uint32, C.c_uint64
_fp = C.POINTER(C.c_float)
SYMBOLS = [
    ("mipt_abi_version", C.c_int, []),
    ("mipt_context_create", C.c_int, [C.c_int, LOG_CALLBACK, _vp, C.c_int, C.POINTER(_vp)]),
    ("mipt_context_destroy", C.c_int, [_vp]),
    ("mipt_error_name", C.c_char_p, [C.c_int]),
    ("mipt_last_error_string", C.c_char_p, [_vp]),
    ("mipt_set_option", C.c_int, [_vp, C.c_int, _u64]),
    ("mipt_get_option", C.c_int, [_vp, C.c_int, C.POINTER(_u64)]),
    ("mipt_accel_build", C.c_int, [_vp, C.POINTER(MiptBuildInput), _u32, C.POINTER(C.c_ulonglong)]),
    ("mipt_accel_update", C.c_int, [_vp, C.c_ulonglong, C.POINTER(MiptBuildInput), _u32]),
    ("mipt_accel_destroy", C.c_int, [_vp, C.c_ulonglong]),
    ("mipt_accel_info", C.c_int, [_vp, C.c_ulonglong, C.POINTER(MiptAccelInfo)]),
    ("mipt_accel_export", C.c_int, [_vp, C.c_ulonglong, _vp, _vp]),
    ("mipt_accel_build_instances", C.c_int, [_vp, _vp, _u32, C.POINTER(C.c_ulonglong)]),
    ("mipt_accel_update_instances", C.c_int, [_vp, C.c_ulonglong, _vp, _u32]),
    ("mipt_accel_export_instances", C.c_int, [_vp, C.c_ulonglong, _vp, _vp, _vp, _vp, _vp]),
    ("mipt_host_bvh_create", C.c_int, [_vp, _u32, C.POINTER(_vp)]),
    ("mipt_host_bvh_create_spatial", C.c_int, [_vp, _u32, C.c_float, C.POINTER(_vp)]),
    ("mipt_host_bvh_info", C.c_int, [_vp, C.POINTER(MiptAccelInfo)]),
    ("mipt_host_bvh_copy", C.c_int, [_vp, _vp, _vp]),
    ("mipt_host_bvh_destroy", C.c_int, [_vp]),
    ("mipt_sbt_set_hitgroups", C.c_int, [_vp, C.POINTER(MiptHitGroupData), _u32]),
    ("mipt_sbt_set_miss", C.c_int, [_vp, C.POINTER(MiptMissData)]),
    ("mipt_launch", C.c_int, [_vp, _vp, C.POINTER(MiptParams), C.c_uint, C.c_uint, C.c_uint]),
    ("mipt_sync", C.c_int, [_vp]),
    ("mipt_get_launch_stats", C.c_int, [_vp, C.POINTER(MiptLaunchStats)]),
    ("mipt_get_launch_timings", C.c_int, [_vp, C.POINTER(MiptLaunchTimings)]),
    ("mipt_trace_closest", C.c_int, [_vp, _vp, C.c_ulonglong, _vp, _u32, _vp]),
    ("mipt_trace_any", C.c_int, [_vp, _vp, C.c_ulonglong, _vp, _u32, _vp]),
    ("mipt_shard_num_samples", C.c_int, [C.c_uint, C.c_uint, C.c_uint, C.POINTER(_u32)]),
    ("mipt_shard_pixel", C.c_int, [C.c_uint, C.c_uint, C.c_uint, C.c_uint, _u32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("mipt_launch_shard", C.c_int, [_vp, _vp, C.POINTER(MiptParams), C.c_uint, C.c_uint, C.c_uint, C.c_uint, _vp]),
    ("mipt_unpermute", C.c_int, [_vp, _vp, _vp, C.c_uint, C.c_uint, C.c_uint, _vp, _vp]),
    ("mipt_group_create", C.c_int, [C.POINTER(C.c_int), _u32, LOG_CALLBACK, _vp, C.c_int, C.POINTER(_vp)]),
    ("mipt_group_destroy", C.c_int, [_vp]),
    ("mipt_group_size", C.c_int, [_vp, C.POINTER(_u32)]),
    ("mipt_group_member", C.c_int, [_vp, _u32, C.POINTER(_vp), C.POINTER(C.c_int), C.POINTER(_vp)]),
    ("mipt_group_launch", C.c_int, [_vp, C.POINTER(MiptParams), C.c_uint, C.c_uint]),
    ("mipt_group_allgather", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    ("mipt_group_sync", C.c_int, [_vp]),
    ("mipt_group_uses_rccl", C.c_int, [_vp]),
    ("mipt_group_last_error_string", C.c_char_p, [_vp]),
    ("mipt_imgui_launch", C.c_int, [_vp, _vp, C.POINTER(MiptImguiParams), C.c_uint, C.c_uint, C.c_uint]),
    ("mipt_imgui_camera_set_fd", None, [C.POINTER(MiptImguiCamera), C.c_float]),
    ("mipt_imgui_camera_compute_uvw", None, [C.POINTER(MiptImguiCamera)]),
    ("mipt_scene_load", C.c_int, [C.c_char_p, C.POINTER(_vp), C.c_char_p, C.c_size_t]),
    ("mipt_scene_destroy", C.c_int, [_vp]),
    ("mipt_scene_info", C.c_int, [_vp, C.POINTER(MiptSceneInfo)]),
    ("mipt_scene_camera", C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp]),
    ("mipt_scene_material", C.c_int, [_vp, _u32, C.POINTER(MiptMaterialData)]),
    ("mipt_scene_flatten", C.c_int, [_vp, _vp, _vp]),
    ("mipt_scene_flatten_normals", C.c_int, [_vp, _vp]),
    ("mipt_scene_mesh_num_triangles", C.c_int, [_vp, _u32, C.POINTER(_u32)]),
    ("mipt_scene_mesh_triangles", C.c_int, [_vp, _u32, _vp, _vp]),
    ("mipt_scene_instance", C.c_int, [_vp, _u32, C.POINTER(_u32), _fp]),
    ("mipt_scene_texture", C.c_int, [_vp, _u32, C.POINTER(MiptTextureData)]),
    ("mipt_scene_flatten_texcoords", C.c_int, [_vp, _u32, _vp, _vp]),
    ("mipt_scene_flatten_colors", C.c_int, [_vp, _vp, _vp]),
    ("mipt_save_image", C.c_int, [C.c_char_p, _vp, _u32, _u32, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    ("mipt_whitted_set_geometry", C.c_int, [_vp, C.POINTER(MiptWhittedGeometry), C.POINTER(MiptMaterialData), _u32,
                                            C.POINTER(MiptTextureData), _u32]),
    ("mipt_whitted_set_geometry_instanced", C.c_int, [_vp, C.POINTER(MiptWhittedMesh), _u32, C.POINTER(MiptWhittedInstance), _u32,
                                                      C.POINTER(MiptMaterialData), _u32, C.POINTER(MiptTextureData), _u32]),
    ("mipt_scene_mesh_normals", C.c_int, [_vp, _u32, _vp]),
    ("mipt_scene_mesh_texcoords", C.c_int, [_vp, _u32, _u32, _vp, _vp]),
    ("mipt_scene_mesh_colors", C.c_int, [_vp, _u32, _vp, _vp]),
    ("mipt_whitted_launch", C.c_int, [_vp, _vp, C.POINTER(MiptWhittedLaunchParams), C.c_uint, C.c_uint, C.c_uint]),
    ("mipt_camera_uvw", None, [_fp, _fp, _fp, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp]),
    ("mipt_scene_random_triangles", C.c_int, [_u32, _u32, C.c_float, _vp]),
]

_lib = None


def load():
    """Loads libmipt.so.  torch is imported first so that the process has ONE HIP runtime
    (the one bundled with PyTorch-ROCm, same SONAME libamdhip64.so.7) shared by torch tensors,
    RCCL and this library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmipt.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C optix_raytracer_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.mipt_abi_version() != 4:
        raise RuntimeError("libmipt.so ABI version mismatch")
    _lib = L
    return L
