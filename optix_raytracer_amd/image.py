"""sutil::saveImage (SDK/sutil/sutil.cpp:542-709) above the C ABI: same container choice by extension, same
vertical flip / sRGB rules, same error texts (raised as MiptError instead of sutil::Exception)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .api import MiptError


def saveImage(fname, image, disable_srgb_conversion=False):
    """image: (h, w, 4) uint8 -> UNSIGNED_BYTE4, (h, w, 4) float32 -> FLOAT4, (h, w, 3) float32 -> FLOAT3
    (sutil::ImageBuffer); row 0 is the bottom row of the picture, as in the reference's output buffers."""
    a = np.ascontiguousarray(image)
    if a.ndim != 3:
        raise MiptError(1, "MIPT_ERROR_INVALID_VALUE", "sutil::saveImage(): Unrecognized image buffer pixel format.")
    if a.dtype == np.uint8 and a.shape[2] == 4:
        fmt = L.IMAGE_UNSIGNED_BYTE4
    elif a.dtype == np.float32 and a.shape[2] == 4:
        fmt = L.IMAGE_FLOAT4
    elif a.dtype == np.float32 and a.shape[2] == 3:
        fmt = L.IMAGE_FLOAT3
    else:
        raise MiptError(1, "MIPT_ERROR_INVALID_VALUE", "sutil::saveImage(): Unrecognized image buffer pixel format.")
    lib = L.load()
    err = C.create_string_buffer(256)
    rc = lib.mipt_save_image(str(fname).encode(), a.ctypes.data, a.shape[1], a.shape[0], fmt,
                             1 if disable_srgb_conversion else 0, err, 256)
    if rc:
        raise MiptError(rc, lib.mipt_error_name(rc).decode(), err.value.decode())
