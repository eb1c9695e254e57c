"""sutil::saveImage rules (SDK/sutil/sutil.cpp:542-709) for the PPM / PNG / EXR writers of libmipt (host only)."""
import struct
import zlib

import numpy as np
import pytest

from optix_raytracer_amd import image as IMG
from optix_raytracer_amd.api import MiptError

f32 = np.float32


def _read_ppm(path):
    b = path.read_bytes()
    assert b[:3] == b"P6\n"
    head, rest = b[3:].split(b"\n255\n", 1)
    w, h = [int(x) for x in head.split()]
    return np.frombuffer(rest, np.uint8).reshape(h, w, 3)


def _read_png(path):
    b = path.read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat += data
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    assert (depth, ctype) == (8, 6)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


def _read_exr(path):
    b = path.read_bytes()
    assert struct.unpack("<II", b[:8]) == (20000630, 2)
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        n = struct.unpack("<I", b[pos:pos + 4])[0]
        attrs[name] = (typ, b[pos + 4:pos + 4 + n]); pos += 4 + n
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, cb, p = [], attrs["channels"][1], 0
    while cb[p] != 0:
        e = cb.index(b"\0", p); names.append(cb[p:e].decode()); assert struct.unpack("<I", cb[e + 1:e + 5])[0] == 1   # HALF
        p = e + 17
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0" and names == sorted(names)
    offs = struct.unpack("<%dQ" % h, b[pos:pos + 8 * h])
    out = {}
    for y in range(h):
        yy, size = struct.unpack("<ii", b[offs[y]:offs[y] + 8])
        assert yy == y and size == 2 * w * len(names)
        line = np.frombuffer(b[offs[y] + 8:offs[y] + 8 + size], np.float16).reshape(len(names), w)
        for k, nme in enumerate(names):
            out.setdefault(nme, np.zeros((h, w), np.float16))[y] = line[k]
    return out


def _expect_quant(f, srgb=True):
    f = np.asarray(f, np.float32)
    if srgb:
        with np.errstate(invalid="ignore"):
            powed = np.power(f.astype(np.float32), np.float32(1.0 / 2.4), dtype=np.float32)
        f = np.where(f < np.float32(0.0031308), np.float32(12.92) * f, np.float32(1.055) * powed - np.float32(0.055)).astype(np.float32)
    v = np.nan_to_num((np.float32(256.0) * f).astype(np.float64), nan=0.0)
    return np.clip(np.trunc(v), 0, 255).astype(np.uint8)


def test_ppm_from_uchar4_is_flipped_rgb(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    IMG.saveImage(tmp_path / "a.ppm", img)
    assert np.array_equal(_read_ppm(tmp_path / "a.ppm"), img[::-1, :, :3])


@pytest.mark.parametrize("comps", [3, 4])
def test_ppm_from_float_applies_srgb_and_clamps(tmp_path, comps):
    rng = np.random.default_rng(2)
    img = rng.uniform(-0.2, 1.4, (6, 9, comps)).astype(f32)
    img[0, 0, :3] = (0.18, 0.09, 0.045)                    # SURVEY.md a11 known answers
    img[0, 1, :3] = (0.5, 0.25, 0.125)
    img[0, 2, :3] = (2.0, 1.0, 0.5)
    IMG.saveImage(tmp_path / "b.PPM", img)
    got = _read_ppm(tmp_path / "b.PPM")
    want = _expect_quant(img[::-1, :, :3])
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1         # libm powf vs numpy: last-ulp ties only
    assert (got != want).mean() < 0.02
    assert tuple(got[-1, 0]) == (118, 84, 60) and tuple(got[-1, 1]) == (188, 137, 99) and tuple(got[-1, 2]) == (255, 255, 188)
    IMG.saveImage(tmp_path / "c.ppm", img, disable_srgb_conversion=True)
    assert np.array_equal(_read_ppm(tmp_path / "c.ppm"), _expect_quant(img[::-1, :, :3], srgb=False))


def test_png_is_rgba8_flipped(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (300, 120, 4), dtype=np.uint8)             # > 65535 bytes: several stored blocks
    IMG.saveImage(tmp_path / "a.png", img)
    assert np.array_equal(_read_png(tmp_path / "a.png"), img[::-1])
    for bad in (np.zeros((2, 2, 3), f32), np.zeros((2, 2, 4), f32)):
        with pytest.raises(MiptError, match="to PNG not implemented yet"):
            IMG.saveImage(tmp_path / "x.png", bad)


@pytest.mark.parametrize("comps", [3, 4])
def test_exr_is_half_unflipped(tmp_path, comps):
    rng = np.random.default_rng(4)
    img = rng.uniform(-3, 70000, (4, 5, comps)).astype(f32)
    img[0, 0, 0], img[0, 1, 0], img[1, 0, 1], img[1, 1, 2] = 1e-7, 65520.0, -0.0, np.inf
    IMG.saveImage(tmp_path / "a.exr", img)
    ch = _read_exr(tmp_path / "a.exr")
    assert sorted(ch) == (["A", "B", "G", "R"] if comps == 4 else ["B", "G", "R"])
    with np.errstate(over="ignore"):
        want = img.astype(np.float16)                                     # numpy rounds to nearest even as well
    for k, nme in enumerate("RGBA"[:comps]):
        assert np.array_equal(ch[nme].view(np.uint16), want[:, :, k].view(np.uint16)), nme
    with pytest.raises(MiptError, match="uchar4 images to EXR"):
        IMG.saveImage(tmp_path / "x.exr", np.zeros((2, 2, 4), np.uint8))


def test_extension_errors(tmp_path):
    img = np.zeros((2, 2, 4), np.uint8)
    with pytest.raises(MiptError, match="unsupported filetype 'jpg'"):
        IMG.saveImage(tmp_path / "a.jpg", img)
    with pytest.raises(MiptError, match="filename extension"):
        IMG.saveImage("a.pp", img)
    with pytest.raises(MiptError, match="could not write"):
        IMG.saveImage(tmp_path / "nodir" / "a.ppm", img)
