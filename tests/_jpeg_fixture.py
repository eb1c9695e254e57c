"""Baseline JPEG encoder for tests (own canonical Huffman tables, any 1..2 sampling factors, optional restart
intervals) + a float64 restatement of the decode (dequantise, IDCT, level shift, replicate chroma, JFIF YCbCr -> RGB)
giving the pixels a decoder should produce to within rounding."""
import struct

import numpy as np

ZIGZAG = np.asarray([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
_C = np.asarray([[(np.sqrt(0.125) if u == 0 else 0.5) * np.cos((2 * x + 1) * u * np.pi / 16) for u in range(8)] for x in range(8)])


def _fdct(block):
    return _C.T @ block @ _C


def _idct(coef):
    return _C @ coef @ _C.T


class _Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, code, length):
        for i in range(length - 1, -1, -1):
            self.acc = (self.acc << 1) | ((code >> i) & 1); self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xff:
                    self.out.append(0)
                self.acc = 0; self.n = 0

    def flush(self):
        while self.n:
            self.put(1, 1)


def _category(v):
    return 0 if v == 0 else int(abs(v)).bit_length()


def _vbits(v, s):
    return v if v >= 0 else v + (1 << s) - 1


def encode(rgb_or_gray, sampling=(1, 1), quality_scale=1.0, restart=0, progressive_marker=False):
    """rgb_or_gray: (h, w, 3) or (h, w) uint8.  Returns (jpeg bytes, expected RGBA uint8)."""
    img = np.asarray(rgb_or_gray, np.float64)
    gray = img.ndim == 2
    h, w = img.shape[:2]
    if gray:
        planes = [img]; samp = [(1, 1)]
    else:
        R, G, B = img[..., 0], img[..., 1], img[..., 2]
        Y = 0.299 * R + 0.587 * G + 0.114 * B
        Cb = -0.168736 * R - 0.331264 * G + 0.5 * B + 128
        Cr = 0.5 * R - 0.418688 * G - 0.081312 * B + 128
        planes = [Y, Cb, Cr]; samp = [sampling, (1, 1), (1, 1)]
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    mcux, mcuy = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    base_q = np.asarray([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                         18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
    q = [np.clip(np.round(base_q * quality_scale), 1, 255).astype(int).reshape(8, 8), np.clip(np.round(base_q * quality_scale * 1.5), 1, 255).astype(int).reshape(8, 8)]
    # subsample + pad planes to whole MCUs (edge replication)
    comp = []
    for k, (pl, (sh, sv)) in enumerate(zip(planes, samp)):
        fx, fy = hmax // sh, vmax // sv
        ph, pw = mcuy * vmax * 8, mcux * hmax * 8
        padded = np.pad(pl, ((0, ph - h), (0, pw - w)), mode="edge")
        sub = padded.reshape(ph // fy, fy, pw // fx, fx).mean(axis=(1, 3))
        comp.append(sub)
    # quantised coefficients per block
    coefs = []
    for k, sub in enumerate(comp):
        qt = q[0 if k == 0 else 1]
        bh, bw = sub.shape[0] // 8, sub.shape[1] // 8
        cb = np.zeros((bh, bw, 8, 8), int)
        for by in range(bh):
            for bx in range(bw):
                cb[by, bx] = np.round(_fdct(sub[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] - 128) / qt).astype(int)
        coefs.append(cb)
    # huffman tables: DC 12 symbols of length 4, AC 162 symbols of length 8 (canonical order = HUFFVAL order)
    dc_syms = list(range(12))
    ac_syms = [0x00, 0xf0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    dc_code = {s: (i, 4) for i, s in enumerate(dc_syms)}
    ac_code = {s: (i, 8) for i, s in enumerate(ac_syms)}
    out = bytearray(b"\xff\xd8")
    out += b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"
    for t in range(1 if gray else 2):
        out += b"\xff\xdb" + struct.pack(">HB", 67, t) + bytes(int(q[t].reshape(-1)[z]) for z in ZIGZAG)
    out += (b"\xff\xc2" if progressive_marker else b"\xff\xc0") + struct.pack(">HBHHB", 8 + 3 * len(comp), 8, h, w, len(comp))
    for k, (sh, sv) in enumerate(samp):
        out += bytes([k + 1, (sh << 4) | sv, 0 if k == 0 else 1])
    for tc, syms, length in ((0, dc_syms, 4), (1, ac_syms, 8)):
        bits = [0] * 16; bits[length - 1] = len(syms)
        out += b"\xff\xc4" + struct.pack(">HB", 19 + len(syms), (tc << 4) | 0) + bytes(bits) + bytes(syms)
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comp), len(comp))
    for k in range(len(comp)):
        out += bytes([k + 1, 0x00])
    out += b"\0\x3f\0"
    bw_ = _Bits()
    pred = [0] * len(comp)
    n_mcu = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if restart and n_mcu and n_mcu % restart == 0:
                bw_.flush()
                out += bytes(bw_.out); bw_ = _Bits()
                out += bytes([0xff, 0xd0 + ((n_mcu // restart - 1) % 8)])
                pred = [0] * len(comp)
            for k, (sh, sv) in enumerate(samp):
                for by in range(sv):
                    for bx in range(sh):
                        zz = coefs[k][my * sv + by, mx * sh + bx].reshape(-1)[ZIGZAG]
                        d = int(zz[0]) - pred[k]; pred[k] = int(zz[0])
                        s = _category(d)
                        bw_.put(*dc_code[s]); bw_.put(_vbits(d, s), s)
                        run = 0
                        last = max([i for i in range(1, 64) if zz[i] != 0], default=0)
                        for i in range(1, last + 1):
                            v = int(zz[i])
                            if v == 0:
                                run += 1
                                continue
                            while run > 15:
                                bw_.put(*ac_code[0xf0]); run -= 16
                            s = _category(v)
                            bw_.put(*ac_code[(run << 4) | s]); bw_.put(_vbits(v, s), s)
                            run = 0
                        if last < 63:
                            bw_.put(*ac_code[0x00])
            n_mcu += 1
    bw_.flush()
    out += bytes(bw_.out) + b"\xff\xd9"
    # ---- what a decoder should give
    rec = []
    for k, cb in enumerate(coefs):
        qt = q[0 if k == 0 else 1]
        bh, bw = cb.shape[:2]
        pl = np.zeros((bh * 8, bw * 8))
        for by in range(bh):
            for bx in range(bw):
                pl[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = np.clip(np.floor(_idct(cb[by, bx] * qt) + 128.5), 0, 255)
        sh, sv = samp[k]
        rec.append(np.repeat(np.repeat(pl, vmax // sv, axis=0), hmax // sh, axis=1)[:h, :w])
    rgba = np.full((h, w, 4), 255, np.uint8)
    if gray:
        rgba[..., 0] = rgba[..., 1] = rgba[..., 2] = rec[0].astype(np.uint8)
    else:
        Y, cb_, cr_ = rec[0], rec[1] - 128, rec[2] - 128
        for k, plane in enumerate((Y + 1.402 * cr_, Y - 0.344136 * cb_ - 0.714136 * cr_, Y + 1.772 * cb_)):
            rgba[..., k] = np.clip(np.floor(plane + 0.5), 0, 255).astype(np.uint8)
    return bytes(out), rgba
