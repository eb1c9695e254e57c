"""Regenerates tests/golden/oracle_cornell_32x32_spp4.npz and oracle_whitted_*.npz from the CPU oracle (regression fixtures of the
oracle itself; the reference cannot be run in this environment -- SURVEY.md 8c)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle as O  # noqa: E402
from optix_raytracer_amd import scenes  # noqa: E402


def cornell_fixture():
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    accum, frame, st = s.render(scenes.launch_desc(sc, 32, 32, spp=4))
    np.savez_compressed(os.path.join(HERE, "oracle_cornell_32x32_spp4.npz"), accum=accum, frame=frame,
                        closest_rays=st["closest_rays"], shadow_rays=st["shadow_rays"])
    print(st, accum[..., :3].mean())

# ---- the Whitted / glTF-PBR restatement on the two generated glTF fixtures (textures, alpha modes, culling):
# tests/golden/oracle_whitted_*.npz pin the oracle against accidental change (CPU only, brute-force traversal so that
# the fixture does not depend on the BVH builder)
import pathlib  # noqa: E402
import tempfile  # noqa: E402

import _gltf_fixture as FX  # noqa: E402
from optix_raytracer_amd import api, meshviewer as MV  # noqa: E402


def whitted_fixture(gltf, w, h):
    sc = api.Scene(gltf)
    verts, mat = sc.flatten()
    mats_c = sc.material_table()
    mats = []
    for m, sm in zip(mats_c, sc.materials() + [None]):
        d = {"base_color": list(m.base_color), "metallic": m.metallic, "roughness": m.roughness, "emissive_factor": list(m.emissive_factor),
             "alpha_mode": m.alpha_mode, "alpha_cutoff": m.alpha_cutoff, "double_sided": m.double_sided}
        if sm is not None:
            for k in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emissive_tex"):
                d[k] = sm[k]
        mats.append(d)
    osc = O.Scene({"vertices": verts, "mat_index": np.zeros(len(mat), np.uint32), "emission": np.zeros((1, 3), np.float32),
                   "diffuse": np.ones((1, 3), np.float32)})
    gf = MV.geometry_flags(mats_c)
    osc.set_prim_flags(gf[np.minimum(mat, len(gf) - 1)])
    cam = sc.camera()
    U, V, W = O.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], w / h)
    info = sc.info()
    lights = MV.default_lights(info["aabb_min"], info["aabb_max"]) + [{"type": 1, "color": (0.05, 0.05, 0.06)}]
    uv = [sc.flatten_texcoords(j) for j in range(2)]
    col = sc.flatten_colors()
    normals = sc.flatten_normals()
    acc = None
    for sf in range(2):
        la = {"width": w, "height": h, "subframe": sf, "eye": cam["eye"], "U": U, "V": V, "W": W, "miss_color": (0.1, 0.1, 0.1)}
        acc, frame, st = osc.render_whitted(la, lights, mats, mat, normals4=normals if normals[:, 3].any() else None, accum=acc, mode=O.MODE_BRUTE,
                                            texcoords=(uv[0][0] if uv[0][1].any() else None, uv[1][0] if uv[1][1].any() else None),
                                            texcoord_present=(uv[0][1], uv[1][1]), colors4=col[0] if col[1].any() else None, color_present=col[1],
                                            textures=sc.textures())
    return acc, frame, st


if __name__ == "__main__":
    cornell_fixture()
    tmp = pathlib.Path(tempfile.mkdtemp())
    for name, writer in (("textured", FX.write_textured), ("alpha", FX.write_alpha)):
        acc, frame, st = whitted_fixture(writer(tmp), 48, 32)
        np.savez_compressed(os.path.join(HERE, "oracle_whitted_%s_48x32.npz" % name), accum=acc, frame=frame,
                            closest_rays=st["closest_rays"], shadow_rays=st["shadow_rays"])
        print(name, st["closest_rays"], st["shadow_rays"], acc[..., :3].mean())
