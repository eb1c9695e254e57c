// jpeg_decode.cpp -- baseline / extended sequential and progressive JPEG (ITU T.81: SOF0 / SOF1 / SOF2, 8 bit, Huffman,
// 1 or 3 components, sampling factors 1..2, restart intervals, any number of scans) -> RGBA8, for glTF images (the
// reference gets this from stb_image through tinygltf).  Arithmetic-coded, lossless, hierarchical and 12-bit files are
// refused.  Every scan fills a buffer of quantised coefficients; the picture is reconstructed once at the end.  IDCT: separable float DCT-III on the
// dequantised block (the definition in T.81 A.3.3), level shift + round to nearest; chroma is upsampled by pixel
// replication; YCbCr -> RGB per JFIF.  Decoders differ by a level or two in these steps; none is "the" JPEG output.
#include "png_decode.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace mipt {

namespace {

struct HuffTable {
    bool present = false;
    uint8_t bits[17] = { 0 }, vals[256] = { 0 };
    int mincode[17], maxcode[18], valptr[17];
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct JpegBits {
    const uint8_t* p; size_t n, pos; uint32_t buf = 0; int cnt = 0; bool hit_marker = false;
    int bit()
    {
        if (cnt == 0) {
            uint8_t b = 0;
            if (!hit_marker && pos < n) {
                b = p[pos++];
                if (b == 0xff) {
                    const uint8_t b2 = pos < n ? p[pos] : 0;
                    if (b2 == 0) ++pos;                       // stuffed zero
                    else { hit_marker = true; --pos; b = 0; } // a marker: feed zeros, leave it for the caller
                }
            }
            buf = b; cnt = 8;
        }
        --cnt;
        return (int)((buf >> cnt) & 1u);
    }
    int receive(int s) { int v = 0; for (int i = 0; i < s; ++i) v = (v << 1) | bit(); return v; }
    void reset() { cnt = 0; buf = 0; hit_marker = false; }
};

int decode_symbol(JpegBits& br, const HuffTable& h)
{
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

inline int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

void idct8x8(const float in[64], uint8_t out[64])
{
    static float C[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; ++x) for (int u = 0; u < 8; ++u)
            C[x][u] = (u == 0 ? 0.353553390593273762f : 0.5f) * (float)std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        init = true;
    }
    float tmp[64];
    for (int v = 0; v < 8; ++v) for (int x = 0; x < 8; ++x) { float s = 0; for (int u = 0; u < 8; ++u) s += C[x][u] * in[v * 8 + u]; tmp[v * 8 + x] = s; }
    for (int x = 0; x < 8; ++x) for (int y = 0; y < 8; ++y) {
        float s = 0; for (int v = 0; v < 8; ++v) s += C[y][v] * tmp[v * 8 + x];
        const float r = std::floor(s + 128.5f);
        out[y * 8 + x] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    int bw = 0, bh = 0;            // blocks per row / column of the MCU-padded block grid (interleaved scans)
    int sw = 0, sh = 0;            // blocks that cover the component itself (scans of this component alone, T.81 A.2.2)
    std::vector<int16_t> coef;     // bw * bh blocks of 64 quantised coefficients, natural (row-major) order
};

inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

struct Scan { int ss = 0, se = 63, ah = 0, al = 0; bool progressive = false; int eobrun = 0; };

// one block of one scan (T.81 F.2 sequential; G.2 progressive: DC first / refinement, AC first / refinement with EOB runs)
bool decode_block(JpegBits& br, const HuffTable* hdc, const HuffTable* hac, Component& C, int16_t* blk, Scan& sc, std::string& why)
{
    if (!sc.progressive) {
        const int t = decode_symbol(br, hdc[C.td]);
        if (t < 0 || t > 11) { why = "JPEG Huffman code not valid"; return false; }
        C.pred = (int)((unsigned)C.pred + (unsigned)extend(br.receive(t), t));      // (wraps on corrupt data instead of overflowing)
        blk[0] = (int16_t)C.pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(br, hac[C.ta]);
            if (rs < 0) { why = "JPEG Huffman code not valid"; return false; }
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) { why = "JPEG coefficient index out of range"; return false; }
            blk[kZigzag[k]] = (int16_t)extend(br.receive(sz), sz);
            ++k;
        }
        return true;
    }
    const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
    if (sc.ss == 0) {
        if (sc.ah == 0) {
            const int t = decode_symbol(br, hdc[C.td]);
            if (t < 0 || t > 11) { why = "JPEG Huffman code not valid"; return false; }
            C.pred = (int)((unsigned)C.pred + (unsigned)extend(br.receive(t), t));
            blk[0] = (int16_t)((unsigned)C.pred << sc.al);
        } else if (br.bit()) {
            blk[0] = (int16_t)(blk[0] | p1);
        }
        return true;
    }
    if (sc.ah == 0) {
        if (sc.eobrun > 0) { --sc.eobrun; return true; }
        for (int k = sc.ss; k <= sc.se;) {
            const int rs = decode_symbol(br, hac[C.ta]);
            if (rs < 0) { why = "JPEG Huffman code not valid"; return false; }
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r == 15) { k += 16; continue; }
                sc.eobrun = (1 << r) - 1;
                if (r) sc.eobrun += br.receive(r);
                break;
            }
            k += r;
            if (k > sc.se) { why = "JPEG coefficient index out of range"; return false; }
            blk[kZigzag[k]] = (int16_t)(extend(br.receive(sz), sz) * p1);
            ++k;
        }
        return true;
    }
    // AC refinement: a correction bit for every coefficient that is already non-zero, new +-1 coefficients in between
    int k = sc.ss;
    if (sc.eobrun == 0) {
        while (k <= sc.se) {
            const int rs = decode_symbol(br, hac[C.ta]);
            if (rs < 0) { why = "JPEG Huffman code not valid"; return false; }
            int r = rs >> 4;
            const int sz = rs & 15;
            int val = 0;
            if (sz != 0) {
                if (sz != 1) { why = "JPEG refinement scan with a coefficient size other than 1"; return false; }
                val = br.bit() ? p1 : m1;
            } else if (r != 15) {
                sc.eobrun = 1 << r;
                if (r) sc.eobrun += br.receive(r);
                break;
            }
            while (k <= sc.se) {
                int16_t& c = blk[kZigzag[k]];
                if (c != 0) {
                    if (br.bit() && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
                } else if (--r < 0) {
                    break;                                     // the zero coefficient this code is about
                }
                ++k;
            }
            if (val != 0) {
                if (k > sc.se) { why = "JPEG coefficient index out of range"; return false; }
                blk[kZigzag[k]] = (int16_t)val;
            }
            ++k;
        }
    }
    if (sc.eobrun > 0) {
        for (; k <= sc.se; ++k) {
            int16_t& c = blk[kZigzag[k]];
            if (c != 0 && br.bit() && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
        }
        --sc.eobrun;
    }
    return true;
}

}  // namespace

bool decode_jpeg(const uint8_t* data, size_t size, DecodedImage& out, std::string& why)
{
    if (size < 4 || data[0] != 0xff || data[1] != 0xd8) { why = "not a JPEG"; return false; }
    uint16_t qt[4][64]; bool have_qt[4] = { false, false, false, false };
    HuffTable hdc[4], hac[4];
    Component comp[3]; int ncomp = 0;
    int width = 0, height = 0, restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool have_frame = false, progressive = false, have_scan = false;
    size_t pos = 2;
    while (pos + 4 <= size) {
        if (data[pos] != 0xff) { ++pos; continue; }
        const uint8_t m = data[pos + 1];
        if (m == 0xff) { ++pos; continue; }
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) { pos += 2; continue; }
        if (m == 0xd9) break;
        const size_t len = be16(data + pos + 2);
        if (len < 2 || pos + 2 + len > size) { why = "JPEG segment reaches past the end of the data"; return false; }
        const uint8_t* s = data + pos + 4; const size_t n = len - 2;
        if (m == 0xdb) {
            size_t i = 0;
            while (i < n) {
                const int pq = s[i] >> 4, tq = s[i] & 15; ++i;
                if (tq > 3 || i + (pq ? 128u : 64u) > n) { why = "bad DQT"; return false; }
                for (int k = 0; k < 64; ++k) { qt[tq][kZigzag[k]] = pq ? (uint16_t)be16(s + i + 2 * k) : s[i + k]; }
                i += pq ? 128 : 64; have_qt[tq] = true;
            }
        } else if (m == 0xc4) {
            size_t i = 0;
            while (i + 17 <= n) {
                const int tc = s[i] >> 4, th = s[i] & 15; ++i;
                if (tc > 1 || th > 3) { why = "bad DHT"; return false; }
                HuffTable& h = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = s[i + l - 1]; total += h.bits[l]; }
                i += 16;
                if (total > 256 || i + (size_t)total > n) { why = "bad DHT"; return false; }
                std::memcpy(h.vals, s + i, (size_t)total); i += (size_t)total;
                h.build(); h.present = true;
            }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
            if (have_frame) { why = "JPEG with more than one frame header"; return false; }
            if (n < 6 || s[0] != 8) { why = "only 8-bit JPEG is supported"; return false; }
            progressive = m == 0xc2;
            height = (int)be16(s + 1); width = (int)be16(s + 3); ncomp = s[5];
            if ((ncomp != 1 && ncomp != 3) || n < 6 + 3 * (size_t)ncomp || width <= 0 || height <= 0 || width > 32768 || height > 32768 ||
                (long long)width * height > (1ll << 28)) { why = "JPEG frame header not supported"; return false; }
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = s[6 + 3 * c]; comp[c].h = s[7 + 3 * c] >> 4; comp[c].v = s[7 + 3 * c] & 15; comp[c].tq = s[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) { why = "JPEG sampling factors not supported"; return false; }
            }
            if (ncomp == 1) comp[0].h = comp[0].v = 1;            // a single component is never interleaved (A.2.2)
            for (int c = 0; c < ncomp; ++c) { hmax = comp[c].h > hmax ? comp[c].h : hmax; vmax = comp[c].v > vmax ? comp[c].v : vmax; }
            mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (int c = 0; c < ncomp; ++c) {
                comp[c].bw = mcux * comp[c].h; comp[c].bh = mcuy * comp[c].v;
                const int cw = (width * comp[c].h + hmax - 1) / hmax, ch = (height * comp[c].v + vmax - 1) / vmax;
                comp[c].sw = (cw + 7) / 8; comp[c].sh = (ch + 7) / 8;
                comp[c].coef.assign((size_t)comp[c].bw * comp[c].bh * 64, 0);
            }
            have_frame = true;
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            why = "lossless / hierarchical / arithmetic-coded JPEG is not supported"; return false;
        } else if (m == 0xdd) {
            if (n >= 2) restart = (int)be16(s);
        } else if (m == 0xda) {
            if (!have_frame) { why = "JPEG scan before the frame header"; return false; }
            const int ns = s[0];
            if (ns < 1 || ns > ncomp || n < 1 + 2 * (size_t)ns + 3) { why = "JPEG scan header not valid"; return false; }
            int order[3] = { 0, 0, 0 };
            for (int k = 0; k < ns; ++k) {
                int c = -1;
                for (int j = 0; j < ncomp; ++j) if (comp[j].id == s[1 + 2 * k]) c = j;
                if (c < 0) { why = "JPEG scan names an unknown component"; return false; }
                comp[c].td = s[2 + 2 * k] >> 4; comp[c].ta = s[2 + 2 * k] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) { why = "JPEG scan uses a missing table"; return false; }
                order[k] = c;
            }
            Scan sc;
            sc.progressive = progressive;
            sc.ss = s[1 + 2 * ns]; sc.se = s[2 + 2 * ns]; sc.ah = s[3 + 2 * ns] >> 4; sc.al = s[3 + 2 * ns] & 15;
            if (!progressive) { sc.ss = 0; sc.se = 63; sc.ah = 0; sc.al = 0; }
            else if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || (sc.ss == 0 && sc.se != 0) || (sc.ss != 0 && ns != 1)) { why = "JPEG progressive scan parameters not valid"; return false; }
            for (int k = 0; k < ns; ++k) {
                const Component& C = comp[order[k]];
                const bool need_dc = sc.ss == 0 && sc.ah == 0, need_ac = !progressive || sc.ss != 0;
                if ((need_dc && !hdc[C.td].present) || (need_ac && !hac[C.ta].present)) { why = "JPEG scan uses a missing table"; return false; }
            }
            // ---- entropy-coded segment: MCUs of h x v blocks per component, or the blocks of one component in raster order
            JpegBits br{ data, size, pos + 2 + len };
            for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
            const int units_x = ns == 1 ? comp[order[0]].sw : mcux, units_y = ns == 1 ? comp[order[0]].sh : mcuy;
            int until_restart = restart;
            for (int uy = 0; uy < units_y; ++uy)
                for (int ux = 0; ux < units_x; ++ux) {
                    if (restart && until_restart == 0) {
                        br.reset();                                 // expect RSTn
                        while (br.pos + 1 < size && !(data[br.pos] == 0xff && data[br.pos + 1] >= 0xd0 && data[br.pos + 1] <= 0xd7)) ++br.pos;
                        if (br.pos + 1 >= size) { why = "JPEG restart marker missing"; return false; }
                        br.pos += 2;
                        for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                        sc.eobrun = 0;
                        until_restart = restart;
                    }
                    if (ns == 1) {
                        Component& C = comp[order[0]];
                        if (!decode_block(br, hdc, hac, C, &C.coef[((size_t)uy * C.bw + ux) * 64], sc, why)) return false;
                    } else {
                        for (int k = 0; k < ns; ++k) {
                            Component& C = comp[order[k]];
                            for (int by = 0; by < C.v; ++by)
                                for (int bx = 0; bx < C.h; ++bx)
                                    if (!decode_block(br, hdc, hac, C, &C.coef[((size_t)(uy * C.v + by) * C.bw + (ux * C.h + bx)) * 64], sc, why)) return false;
                        }
                    }
                    if (restart) --until_restart;
                }
            have_scan = true;
            // the next marker follows the entropy-coded data
            size_t q = br.pos;
            while (q + 1 < size && !(data[q] == 0xff && data[q + 1] != 0x00 && data[q + 1] != 0xff && !(data[q + 1] >= 0xd0 && data[q + 1] <= 0xd7))) ++q;
            pos = q;
            continue;
        }
        pos += 2 + len;
    }
    if (!have_scan) { why = "JPEG without a scan"; return false; }
    // ---- dequantise, inverse DCT
    std::vector<uint8_t> plane[3];
    for (int c = 0; c < ncomp; ++c) {
        if (!have_qt[comp[c].tq]) { why = "JPEG component uses a missing quantisation table"; return false; }
        const int pw = comp[c].bw * 8;
        plane[c].assign((size_t)pw * comp[c].bh * 8, 0);
        for (int by = 0; by < comp[c].bh; ++by)
            for (int bx = 0; bx < comp[c].bw; ++bx) {
                const int16_t* blk = &comp[c].coef[((size_t)by * comp[c].bw + bx) * 64];
                float coef[64];
                for (int k = 0; k < 64; ++k) coef[k] = (float)((int)blk[k] * (int)qt[comp[c].tq][k]);
                uint8_t px[64];
                idct8x8(coef, px);
                for (int y = 0; y < 8; ++y) std::memcpy(&plane[c][(size_t)(by * 8 + y) * pw + bx * 8], px + 8 * y, 8);
            }
    }
    // ---- colour conversion
    out.width = (uint32_t)width; out.height = (uint32_t)height; out.bits = 8;
    out.rgba.assign((size_t)width * height * 4, 255);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            uint8_t* o = &out.rgba[((size_t)y * width + x) * 4];
            auto at = [&](int c) { return (float)plane[c][(size_t)(y * comp[c].v / vmax) * (comp[c].bw * 8) + (size_t)(x * comp[c].h / hmax)]; };
            if (ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)at(0); continue; }
            const float Y = at(0), cb = at(1) - 128.f, cr = at(2) - 128.f;
            const float rgb[3] = { Y + 1.402f * cr, Y - 0.344136f * cb - 0.714136f * cr, Y + 1.772f * cb };
            for (int k = 0; k < 3; ++k) { const float r = std::floor(rgb[k] + 0.5f); o[k] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r)); }
        }
    return true;
}

}  // namespace mipt
