// png_decode.h -- PNG -> RGBA8 / RGBA16 and JPEG -> RGBA8 for glTF images (host only).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mipt {

struct DecodedImage {
    uint32_t width = 0, height = 0;
    uint32_t bits = 0;                 // 8 or 16 per component; always 4 components (what tinygltf hands sutil::Scene)
    std::vector<uint8_t> rgba;         // bits == 16: little-endian uint16 per component
};

// false + `why` when the bytes are not a PNG this decoder handles (Adam7 interlacing is not supported)
bool decode_png(const uint8_t* data, size_t size, DecodedImage& out, std::string& why);

// baseline / extended sequential / progressive JPEG (8 bit, Huffman, 1 or 3 components) -> RGBA8 (jpeg_decode.cpp)
bool decode_jpeg(const uint8_t* data, size_t size, DecodedImage& out, std::string& why);

// raw DEFLATE (RFC 1951); false on a malformed stream
bool inflate_raw(const uint8_t* src, size_t n, std::vector<uint8_t>& out);

}  // namespace mipt
