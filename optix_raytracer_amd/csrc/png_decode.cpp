// png_decode.cpp -- the image side of sutil::loadScene (Scene.cpp:303-320): the reference lets tinygltf / stb_image
// decode every glTF image to 4 components of 8 or 16 bits; this is an own decoder for the PNG container
// (RFC 2083: colour types 0, 2, 3, 4, 6, bit depths 1..16, tRNS, all five filters; no Adam7) on top of an own
// DEFLATE decoder (RFC 1951).
#include "png_decode.h"

#include <cstring>

namespace mipt {

namespace {

struct BitReader {
    const uint8_t* p; size_t n, pos = 0; uint32_t buf = 0; int cnt = 0; bool bad = false;
    uint32_t bits(int k)
    {
        while (cnt < k) {
            if (pos >= n) { bad = true; return 0; }
            buf |= (uint32_t)p[pos++] << cnt; cnt += 8;
        }
        const uint32_t v = k ? buf & ((1u << k) - 1u) : 0u;
        buf >>= k; cnt -= k;
        return v;
    }
};

struct Huffman {                       // canonical code, decoded bit by bit (counts per length + sorted symbols)
    uint16_t count[16], symbol[288];
    bool build(const uint8_t* lengths, int n)
    {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; ++len) { left <<= 1; left -= count[len]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; ++i) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            code |= (int)br.bits(1);
            if (br.bad) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t  kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t  kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

bool inflate_block(BitReader& br, const Huffman& lit, const Huffman& dist, std::vector<uint8_t>& out)
{
    for (;;) {
        const int sym = lit.decode(br);
        if (sym < 0) return false;
        if (sym < 256) { out.push_back((uint8_t)sym); continue; }
        if (sym == 256) return true;
        if (sym > 285) return false;
        const uint32_t len = kLenBase[sym - 257] + br.bits(kLenExtra[sym - 257]);
        const int ds = dist.decode(br);
        if (ds < 0 || ds > 29) return false;
        const uint32_t d = kDistBase[ds] + br.bits(kDistExtra[ds]);
        if (br.bad || d > out.size()) return false;
        const size_t from = out.size() - d;
        for (uint32_t i = 0; i < len; ++i) out.push_back(out[from + i]);
    }
}

}  // namespace

bool inflate_raw(const uint8_t* src, size_t n, std::vector<uint8_t>& out)
{
    BitReader br{ src, n };
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.buf = 0; br.cnt = 0;                                   // skip to the byte boundary
            if (br.pos + 4 > n) return false;
            const uint32_t len = src[br.pos] | ((uint32_t)src[br.pos + 1] << 8), nlen = src[br.pos + 2] | ((uint32_t)src[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > n) return false;
            out.insert(out.end(), src + br.pos, src + br.pos + len);
            br.pos += len;
        } else if (type == 1) {
            uint8_t l[288];
            for (int i = 0; i < 144; ++i) l[i] = 8;
            for (int i = 144; i < 256; ++i) l[i] = 9;
            for (int i = 256; i < 280; ++i) l[i] = 7;
            for (int i = 280; i < 288; ++i) l[i] = 8;
            uint8_t d[30]; for (int i = 0; i < 30; ++i) d[i] = 5;
            Huffman lit, dist;
            if (!lit.build(l, 288) || !dist.build(d, 30) || !inflate_block(br, lit, dist, out)) return false;
        } else if (type == 2) {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            if (br.bad || nlen > 286 || ndist > 30) return false;
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            uint8_t lengths[320]; std::memset(lengths, 0, sizeof lengths);
            for (int i = 0; i < ncode; ++i) lengths[order[i]] = (uint8_t)br.bits(3);
            Huffman cl;
            if (!cl.build(lengths, 19)) return false;
            uint8_t ll[320]; int idx = 0;
            while (idx < nlen + ndist) {
                const int sym = cl.decode(br);
                if (sym < 0) return false;
                if (sym < 16) { ll[idx++] = (uint8_t)sym; continue; }
                int rep; uint8_t val = 0;
                if (sym == 16) { if (idx == 0) return false; val = ll[idx - 1]; rep = 3 + (int)br.bits(2); }
                else if (sym == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (br.bad || idx + rep > nlen + ndist) return false;
                while (rep--) ll[idx++] = val;
            }
            if (ll[256] == 0) return false;
            Huffman lit, dist;
            if (!lit.build(ll, nlen)) return false;
            if (!dist.build(ll + nlen, ndist)) return false;
            if (!inflate_block(br, lit, dist, out)) return false;
        } else return false;
        if (last) return true;
    }
}

namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

bool decode_png(const uint8_t* data, size_t size, DecodedImage& out, std::string& why)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (size < 8 || std::memcmp(data, sig, 8) != 0) { why = "not a PNG"; return false; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= size) {
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        if (pos + 12 + (size_t)len > size) { why = "PNG chunk reaches past the end of the data"; return false; }
        const uint8_t* body = data + pos + 8;
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 32768 || h > 32768) { why = "PNG without a sane IHDR"; return false; }
    if (interlace != 0) { why = "Adam7-interlaced PNG is not supported"; return false; }
    int channels;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break;
                     default: why = "PNG colour type not supported"; return false; }
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok) { why = "PNG bit depth not valid for its colour type"; return false; }
    if (ctype == 3 && plte.size() < 3) { why = "palette PNG without PLTE"; return false; }
    if (idat.size() < 6) { why = "PNG without image data"; return false; }
    std::vector<uint8_t> raw;
    const size_t stride = ((size_t)w * (size_t)channels * (size_t)depth + 7) / 8;
    raw.reserve((stride + 1) * h);
    if (!inflate_raw(idat.data() + 2, idat.size() - 2, raw) || raw.size() < (stride + 1) * h) { why = "PNG data does not inflate to the image size"; return false; }
    // ---- unfilter in place
    const size_t bpp = ((size_t)channels * (size_t)depth + 7) / 8;
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t* row = &raw[(stride + 1) * y + 1];
        const uint8_t* up = y ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
        const int ft = raw[(stride + 1) * y];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? row[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int v = row[i];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break;
                          default: why = "PNG filter type not valid"; return false; }
            row[i] = (uint8_t)v;
        }
    }
    // ---- expand to RGBA
    out.width = w; out.height = h; out.bits = depth == 16 ? 16 : 8;
    const size_t comp_bytes = out.bits / 8;
    out.rgba.assign((size_t)w * h * 4 * comp_bytes, 0);
    auto put = [&](size_t px, int c, uint32_t v) {
        if (out.bits == 8) out.rgba[px * 4 + (size_t)c] = (uint8_t)v;
        else { out.rgba[(px * 4 + (size_t)c) * 2] = (uint8_t)(v & 0xff); out.rgba[(px * 4 + (size_t)c) * 2 + 1] = (uint8_t)(v >> 8); }
    };
    const uint32_t maxv = out.bits == 16 ? 65535u : 255u;
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &raw[(stride + 1) * y + 1];
        for (uint32_t x = 0; x < w; ++x) {
            const size_t px = (size_t)y * w + x;
            uint32_t s[4] = { 0, 0, 0, 0 };
            for (int c = 0; c < channels; ++c) {
                if (depth == 16) { const size_t o = ((size_t)x * (size_t)channels + (size_t)c) * 2; s[c] = ((uint32_t)row[o] << 8) | row[o + 1]; }
                else if (depth == 8) s[c] = row[(size_t)x * (size_t)channels + (size_t)c];
                else { const size_t bit = (size_t)x * (size_t)depth; s[c] = (row[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1u << depth) - 1u); }
            }
            if (ctype == 3) {
                const uint32_t i = s[0];
                if ((size_t)i * 3 + 2 >= plte.size()) { why = "PNG palette index out of range"; return false; }
                put(px, 0, plte[i * 3]); put(px, 1, plte[i * 3 + 1]); put(px, 2, plte[i * 3 + 2]);
                put(px, 3, i < trns.size() ? trns[i] : 255u);
            } else if (ctype == 0) {
                const uint32_t g = depth < 8 ? s[0] * 255u / ((1u << depth) - 1u) : s[0];
                bool transparent = false;
                if (trns.size() >= 2) transparent = ((((uint32_t)trns[0] << 8) | trns[1]) & ((1u << depth) - 1u)) == s[0];
                put(px, 0, g); put(px, 1, g); put(px, 2, g); put(px, 3, transparent ? 0u : maxv);
            } else if (ctype == 4) {
                put(px, 0, s[0]); put(px, 1, s[0]); put(px, 2, s[0]); put(px, 3, s[1]);
            } else if (ctype == 2) {
                bool transparent = false;
                if (trns.size() >= 6) {
                    const uint32_t m = (1u << depth) - 1u;
                    transparent = (((((uint32_t)trns[0] << 8) | trns[1]) & m) == s[0]) && (((((uint32_t)trns[2] << 8) | trns[3]) & m) == s[1]) &&
                                  (((((uint32_t)trns[4] << 8) | trns[5]) & m) == s[2]);
                }
                put(px, 0, s[0]); put(px, 1, s[1]); put(px, 2, s[2]); put(px, 3, transparent ? 0u : maxv);
            } else {
                put(px, 0, s[0]); put(px, 1, s[1]); put(px, 2, s[2]); put(px, 3, s[3]);
            }
        }
    }
    return true;
}

}  // namespace mipt
