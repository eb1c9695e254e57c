#include "png_decode.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <string>
#include <fstream>
#include <iterator>
int main(int argc, char** argv) {
    std::mt19937 rng(7);
    int ok = 0, bad = 0;
    for (int f = 1; f < argc; ++f) {
        std::ifstream in(argv[f], std::ios::binary);
        std::vector<uint8_t> src((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        for (int it = 0; it < 4000; ++it) {
            std::vector<uint8_t> b = src;
            int mode = it % 4;
            if (mode == 0) b.resize(20 + rng() % (b.size() - 20));
            else if (mode == 1) { int n = 1 + rng() % 6; for (int k = 0; k < n; ++k) b[2 + rng() % (b.size() - 2)] = (uint8_t)rng(); }
            else if (mode == 2) { size_t i = 2 + rng() % (b.size() - 12); b.erase(b.begin() + i, b.begin() + i + 1 + rng() % 8); }
            else { size_t i = 2 + rng() % (b.size() - 2); b.insert(b.begin() + i, (uint8_t)rng()); }
            mipt::DecodedImage out; std::string why;
            if ((argv[f][strlen(argv[f])-2]==0x6e ? mipt::decode_png(b.data(), b.size(), out, why) : mipt::decode_jpeg(b.data(), b.size(), out, why))) ++ok; else ++bad;
        }
    }
    printf("ok %d refused %d\n", ok, bad);
}
