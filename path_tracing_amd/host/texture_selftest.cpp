// texture_selftest — runs TextureData::append_texture of the C++ host mirror on a file of raw image bytes and
// writes the flattened f32 texels, so that a test can compare them with the Python mirror bit for bit.
//   texture_selftest <format> <width> <height> <in.bin> <out.f32>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

#include "raytrace_plugin.hpp"

int main(int argc, char** argv) {
    using namespace path_tracing;
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s <format> <width> <height> <in.bin> <out.f32>\n", argv[0]);
        return 2;
    }
    TextureFormat f = TextureFormat::Other;
    const char* names[] = {"Rgba8UnormSrgb", "Rgba16Float", "Rgb9e5Ufloat", "R8Unorm", "Rg8Unorm"};
    for (int k = 0; k < 5; ++k)
        if (!std::strcmp(argv[1], names[k])) f = (TextureFormat)k;
    std::ifstream in(argv[4], std::ios::binary);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    TextureData td;
    td.data.assign(3, 0.5f);  // something already in the pool: the offset must point past it
    try {
        const PtTexture t = td.append_texture(bytes.data(), bytes.size(), (uint32_t)std::atoi(argv[2]), (uint32_t)std::atoi(argv[3]), f);
        std::printf("%u %u %u %u\n", t.width, t.height, t.offset, t.format);
    } catch (const std::invalid_argument& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    std::ofstream out(argv[5], std::ios::binary);
    out.write(reinterpret_cast<const char*>(td.data.data() + 3), (std::streamsize)((td.data.size() - 3) * sizeof(float)));
    return 0;
}
