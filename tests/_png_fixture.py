"""PNG encoder for tests: any colour type / bit depth / filter, zlib with stored, fixed or dynamic Huffman blocks.
Returns the bytes and the RGBA expansion (uint8 or uint16) a decoder must produce."""
import struct
import zlib

import numpy as np


def _chunk(typ, data):
    return struct.pack(">I", len(data)) + typ + data + struct.pack(">I", zlib.crc32(typ + data))


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, mode):
    """rows: list of bytes; mode 0..4 or 'mix' (cycles through all five)"""
    out = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = (y % 5) if mode == "mix" else mode
        out.append(ft)
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out.append((v - pred) & 0xff)
        prev = row
    return bytes(out)


def encode(samples, ctype, depth, filter_mode="mix", zmode="dynamic", palette=None, trns=None, idat_split=0):
    """samples: (h, w, channels) integer array of raw sample values (palette indices for ctype 3)"""
    s = np.asarray(samples)
    h, w, ch = s.shape
    assert ch == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    rows = []
    for y in range(h):
        if depth == 16:
            rows.append(s[y].astype(">u2").tobytes())
        elif depth == 8:
            rows.append(s[y].astype(np.uint8).tobytes())
        else:
            bits = "".join(format(int(v), "0%db" % depth) for v in s[y].reshape(-1))
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    bpp = max(1, ch * depth // 8)
    raw = _filter_rows(rows, bpp, filter_mode)
    if zmode == "stored":
        z = zlib.compress(raw, 0)
    elif zmode == "fixed":
        co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
        z = co.compress(raw) + co.flush()
    else:
        z = zlib.compress(raw, 9)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        png += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        png += _chunk(b"tRNS", bytes(trns))
    if idat_split:
        for i in range(0, len(z), idat_split):
            png += _chunk(b"IDAT", z[i:i + idat_split])
    else:
        png += _chunk(b"IDAT", z)
    png += _chunk(b"IEND", b"")
    # ---- expected RGBA
    maxv = 65535 if depth == 16 else 255
    dt = np.uint16 if depth == 16 else np.uint8
    rgba = np.zeros((h, w, 4), np.uint32)
    if ctype == 3:
        pal = np.asarray(palette, np.uint32).reshape(-1, 3)
        alpha = np.full(len(pal), 255, np.uint32)
        if trns is not None:
            alpha[:len(trns)] = list(trns)
        rgba[..., :3] = pal[s[..., 0]]
        rgba[..., 3] = alpha[s[..., 0]]
    elif ctype == 0:
        g = s[..., 0].astype(np.uint32)
        gs = g * 255 // ((1 << depth) - 1) if depth < 8 else g
        rgba[..., 0] = rgba[..., 1] = rgba[..., 2] = gs
        rgba[..., 3] = maxv
        if trns is not None:
            key = (trns[0] << 8 | trns[1]) & ((1 << depth) - 1)
            rgba[..., 3] = np.where(g == key, 0, maxv)
    elif ctype == 4:
        rgba[..., 0] = rgba[..., 1] = rgba[..., 2] = s[..., 0]
        rgba[..., 3] = s[..., 1]
    elif ctype == 2:
        rgba[..., :3] = s
        rgba[..., 3] = maxv
        if trns is not None:
            m = (1 << depth) - 1
            key = [(trns[2 * k] << 8 | trns[2 * k + 1]) & m for k in range(3)]
            rgba[..., 3] = np.where((s[..., 0] == key[0]) & (s[..., 1] == key[1]) & (s[..., 2] == key[2]), 0, maxv)
    else:
        rgba[...] = s
    return png, rgba.astype(dt)
