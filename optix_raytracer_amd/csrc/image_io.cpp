// image_io.cpp -- image output with the rules of sutil::saveImage (SDK/sutil/sutil.cpp:542-709); host only.
//   .ppm  uchar4 / float3 / float4 -> binary P6, rows written top to bottom (buffer row 0 is the BOTTOM row),
//         float channels through toSRGB (unless disabled) and int(256 * x) clamped to 0..255   (:549-619, savePPM :97-117)
//   .png  uchar4 only, RGBA8, flipped vertically (stbi_flip_vertically_on_write(true), :626-638)
//   .exr  float3 / float4 only, half-float channels, rows in buffer order (SaveEXR does not flip, :667-693)
// The reference leans on stb_image_write and tinyexr; this file writes the containers itself: PNG with stored
// (uncompressed) deflate blocks, EXR with NO_COMPRESSION scan lines.  Same pixels, larger files.
#include "mipt.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

int fail(char* errbuf, size_t errlen, int code, const char* fmt, ...)
{
    if (errbuf && errlen) {
        va_list ap; va_start(ap, fmt);
        std::vsnprintf(errbuf, errlen, fmt, ap);
        va_end(ap);
    }
    return code;
}

float to_srgb(float c)                                   // sutil.cpp:535-540 (host libm powf, as the reference)
{
    const float inv_gamma = 1.0f / 2.4f;
    const float powed = std::pow(c, inv_gamma);
    return c < 0.0031308f ? 12.92f * c : 1.055f * powed - 0.055f;
}

uint8_t quantize(float f, bool disable_srgb)
{
    const float x = 256.0f * (disable_srgb ? f : to_srgb(f));
    if (!(x == x)) return 0;                             // NaN: static_cast<int32_t> is undefined there; choose 0
    const int32_t v = x >= 2147483648.0f ? 0x7fffffff : (x <= -2147483648.0f ? -0x7fffffff - 1 : (int32_t)x);
    return (uint8_t)(v < 0 ? 0 : v > 0xff ? 0xff : v);
}

bool write_file(const char* fname, const std::vector<uint8_t>& bytes)
{
    FILE* f = std::fopen(fname, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    return std::fclose(f) == 0 && ok;
}

// ---- PNG ---------------------------------------------------------------------------------------
uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

void put_be32(std::vector<uint8_t>& o, uint32_t v) { for (int s = 24; s >= 0; s -= 8) o.push_back((uint8_t)(v >> s)); }

void png_chunk(std::vector<uint8_t>& o, const char* type, const std::vector<uint8_t>& data)
{
    put_be32(o, (uint32_t)data.size());
    const size_t at = o.size();
    o.insert(o.end(), type, type + 4);
    o.insert(o.end(), data.begin(), data.end());
    put_be32(o, crc32_of(o.data() + at, o.size() - at));
}

std::vector<uint8_t> png_rgba8(const uint8_t* rgba, uint32_t w, uint32_t h, bool flip)
{
    std::vector<uint8_t> raw;                            // filter byte 0 + row
    raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = rgba + (size_t)(flip ? h - 1 - y : y) * w * 4;
        raw.push_back(0);
        raw.insert(raw.end(), row, row + (size_t)w * 4);
    }
    std::vector<uint8_t> z;                              // zlib stream of stored blocks
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;                               // adler32
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        for (size_t i = pos; i < pos + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    } while (pos < raw.size());
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> o = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8 bit RGBA
    png_chunk(o, "IHDR", ihdr);
    png_chunk(o, "IDAT", z);
    png_chunk(o, "IEND", {});
    return o;
}

// ---- EXR ---------------------------------------------------------------------------------------
uint16_t float_to_half(float f)                          // round to nearest even, overflow -> inf
{
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mant = x & 0x007fffffu;
    const int32_t  exp = (int32_t)((x >> 23) & 0xffu);
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0u));
    const int32_t e = exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        const uint32_t m = mant | 0x00800000u;
        const uint32_t shift = (uint32_t)(14 - e);
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) ++hm;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = ((uint32_t)e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;    // may carry into the exponent: still correct
    return (uint16_t)(sign | h);
}

void put_le32(std::vector<uint8_t>& o, uint32_t v) { for (int s = 0; s < 32; s += 8) o.push_back((uint8_t)(v >> s)); }
void put_le64(std::vector<uint8_t>& o, uint64_t v) { for (int s = 0; s < 64; s += 8) o.push_back((uint8_t)(v >> s)); }
void put_str(std::vector<uint8_t>& o, const char* s) { o.insert(o.end(), s, s + std::strlen(s) + 1); }
void put_attr(std::vector<uint8_t>& o, const char* name, const char* type, const std::vector<uint8_t>& v)
{
    put_str(o, name); put_str(o, type); put_le32(o, (uint32_t)v.size()); o.insert(o.end(), v.begin(), v.end());
}

std::vector<uint8_t> exr_half(const float* px, uint32_t w, uint32_t h, int comps)
{
    std::vector<uint8_t> o;
    put_le32(o, 20000630u); put_le32(o, 2u);                               // magic, version 2, single-part scan line
    const char* names4[4] = { "A", "B", "G", "R" };                        // channels are stored in alphabetical order
    const int   src4[4] = { 3, 2, 1, 0 };
    const int   first = comps == 4 ? 0 : 1;
    std::vector<uint8_t> ch;
    for (int c = first; c < 4; ++c) { put_str(ch, names4[c]); put_le32(ch, 1u /* HALF */); put_le32(ch, 0u); put_le32(ch, 1u); put_le32(ch, 1u); }
    ch.push_back(0);
    put_attr(o, "channels", "chlist", ch);
    put_attr(o, "compression", "compression", { 0 });
    std::vector<uint8_t> box; put_le32(box, 0); put_le32(box, 0); put_le32(box, w - 1); put_le32(box, h - 1);
    put_attr(o, "dataWindow", "box2i", box);
    put_attr(o, "displayWindow", "box2i", box);
    put_attr(o, "lineOrder", "lineOrder", { 0 });
    std::vector<uint8_t> one; { const float f = 1.0f; uint32_t u; std::memcpy(&u, &f, 4); put_le32(one, u); }
    put_attr(o, "pixelAspectRatio", "float", one);
    std::vector<uint8_t> v2; put_le32(v2, 0); put_le32(v2, 0);
    put_attr(o, "screenWindowCenter", "v2f", v2);
    put_attr(o, "screenWindowWidth", "float", one);
    o.push_back(0);
    const int nch = 4 - first;
    const uint64_t line_bytes = 8ull + (uint64_t)w * 2ull * (uint64_t)nch;
    const uint64_t table_at = o.size();
    for (uint32_t y = 0; y < h; ++y) put_le64(o, table_at + 8ull * h + line_bytes * y);
    for (uint32_t y = 0; y < h; ++y) {
        put_le32(o, y); put_le32(o, (uint32_t)(line_bytes - 8));
        for (int c = first; c < 4; ++c)
            for (uint32_t x = 0; x < w; ++x) {
                const uint16_t hv = float_to_half(px[((size_t)y * w + x) * (size_t)comps + (size_t)src4[c]]);
                o.push_back((uint8_t)(hv & 0xff)); o.push_back((uint8_t)(hv >> 8));
            }
    }
    return o;
}

}  // namespace

extern "C" int mipt_save_image(const char* fname, const void* data, uint32_t width, uint32_t height, int pixel_format,
                               int disable_srgb_conversion, char* errbuf, size_t errlen)
{
    if (!fname || !data || width == 0 || height == 0)
        return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "mipt_save_image: null file name / data or empty image");
    const std::string filename(fname);
    if (filename.length() < 5)
        return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage(): Failed to determine filename extension");
    const std::string ext = filename.substr(filename.length() - 3);
    const bool srgb_off = disable_srgb_conversion != 0;
    std::vector<uint8_t> bytes;
    if (ext == "PPM" || ext == "ppm") {
        std::vector<uint8_t> pix((size_t)width * height * 3);
        const int comps = pixel_format == MIPT_IMAGE_FLOAT3 ? 3 : 4;
        if (pixel_format != MIPT_IMAGE_UNSIGNED_BYTE4 && pixel_format != MIPT_IMAGE_FLOAT4 && pixel_format != MIPT_IMAGE_FLOAT3)
            return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage(): Unrecognized image buffer pixel format.");
        for (uint32_t j = 0; j < height; ++j)
            for (uint32_t i = 0; i < width; ++i) {
                uint8_t* dst = &pix[3 * ((size_t)width * (height - j - 1) + i)];
                const size_t src = ((size_t)width * j + i) * (size_t)comps;
                for (int e = 0; e < 3; ++e)
                    dst[e] = pixel_format == MIPT_IMAGE_UNSIGNED_BYTE4 ? ((const uint8_t*)data)[src + (size_t)e]
                                                                       : quantize(((const float*)data)[src + (size_t)e], srgb_off);
            }
        char head[64];
        const int n = std::snprintf(head, sizeof head, "P6\n%u %u\n255\n", width, height);
        bytes.assign(head, head + n);
        bytes.insert(bytes.end(), pix.begin(), pix.end());
    } else if (ext == "PNG" || ext == "png") {
        if (pixel_format == MIPT_IMAGE_FLOAT3)
            return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage(): saving of float3 images to PNG not implemented yet");
        if (pixel_format == MIPT_IMAGE_FLOAT4)
            return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage(): saving of float4 images to PNG not implemented yet");
        if (pixel_format != MIPT_IMAGE_UNSIGNED_BYTE4)
            return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage: Unrecognized image buffer pixel format.");
        bytes = png_rgba8((const uint8_t*)data, width, height, true);
    } else if (ext == "EXR" || ext == "exr") {
        if (pixel_format == MIPT_IMAGE_UNSIGNED_BYTE4)
            return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage(): saving of uchar4 images to EXR not implemented yet");
        if (pixel_format != MIPT_IMAGE_FLOAT4 && pixel_format != MIPT_IMAGE_FLOAT3)
            return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage: Unrecognized image buffer pixel format.");
        bytes = exr_half((const float*)data, width, height, pixel_format == MIPT_IMAGE_FLOAT4 ? 4 : 3);
    } else {
        return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "sutil::saveImage(): Failed unsupported filetype '%s'", ext.c_str());
    }
    if (!write_file(fname, bytes)) return fail(errbuf, errlen, MIPT_ERROR_INVALID_VALUE, "mipt_save_image: could not write '%s'", fname);
    return MIPT_SUCCESS;
}
