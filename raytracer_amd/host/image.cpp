// PNG and OpenEXR writers (src/image.cpp:12-53 of the reference, which goes
// through libpng and tinyexr; their headers are not available to this build, so
// both containers are written directly: PNG with zlib's deflate, EXR as
// uncompressed scan lines).  Rows are flipped on the way out: the renderer's row
// 0 is the bottom of the picture (image.cpp:18-21, 38-42).
#include "image.hpp"

#include <zlib.h>

#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {
namespace {

void put32be(std::vector<unsigned char>& out, std::uint32_t v) {
    out.push_back(static_cast<unsigned char>(v >> 24));
    out.push_back(static_cast<unsigned char>(v >> 16));
    out.push_back(static_cast<unsigned char>(v >> 8));
    out.push_back(static_cast<unsigned char>(v));
}

void chunk(std::vector<unsigned char>& file, const char* type, const std::vector<unsigned char>& body) {
    put32be(file, static_cast<std::uint32_t>(body.size()));
    const std::size_t start = file.size();
    file.insert(file.end(), type, type + 4);
    file.insert(file.end(), body.begin(), body.end());
    const uLong crc = crc32(0L, file.data() + start, static_cast<uInt>(file.size() - start));
    put32be(file, static_cast<std::uint32_t>(crc));
}

void writeFile(const std::filesystem::path& path, const void* data, std::size_t size, const char* what) {
    std::ofstream out(path, std::ios::binary);
    if (!out || !out.write(static_cast<const char*>(data), static_cast<std::streamsize>(size)))
        throw std::runtime_error(std::string("Unable to write ") + what + ": " + path.string());
}

// binary32 -> binary16, round to nearest even, overflow to infinity.
std::uint16_t toHalf(float value) {
    std::uint32_t bits;
    std::memcpy(&bits, &value, 4);
    const std::uint32_t sign = (bits >> 16) & 0x8000U;
    const std::uint32_t exponent = (bits >> 23) & 0xffU;
    std::uint32_t mantissa = bits & 0x7fffffU;
    if (exponent == 0xffU)
        return static_cast<std::uint16_t>(sign | 0x7c00U | (mantissa ? 0x200U : 0U));
    const int e = static_cast<int>(exponent) - 127 + 15;
    if (e >= 31)
        return static_cast<std::uint16_t>(sign | 0x7c00U);
    if (e <= 0) {
        if (e < -10)
            return static_cast<std::uint16_t>(sign);
        mantissa |= 0x800000U;
        const int shift = 14 - e;
        std::uint32_t half = mantissa >> shift;
        const std::uint32_t rest = mantissa & ((1U << shift) - 1U), halfway = 1U << (shift - 1);
        if (rest > halfway || (rest == halfway && (half & 1U)))
            ++half;
        return static_cast<std::uint16_t>(sign | half);
    }
    std::uint32_t half = static_cast<std::uint32_t>(e) << 10 | mantissa >> 13;
    const std::uint32_t rest = mantissa & 0x1fffU;
    if (rest > 0x1000U || (rest == 0x1000U && (half & 1U)))
        ++half;  // a carry into the exponent is the right answer, up to infinity
    return static_cast<std::uint16_t>(sign | half);
}

}  // namespace

void writePng(const std::filesystem::path& path, std::uint32_t width, std::uint32_t height,
              std::span<const std::uint8_t> pixels) {
    const auto stride = static_cast<std::size_t>(width) * 4;
    if (pixels.size() != stride * height)
        throw std::runtime_error("Invalid RGBA image size");
    std::vector<unsigned char> raw((stride + 1) * height);
    for (std::uint32_t row = 0; row < height; ++row) {
        raw[(stride + 1) * row] = 0;  // filter "none"
        std::memcpy(raw.data() + (stride + 1) * row + 1, pixels.data() + (height - row - 1) * stride, stride);
    }
    uLongf size = compressBound(static_cast<uLong>(raw.size()));
    std::vector<unsigned char> compressed(size);
    if (compress2(compressed.data(), &size, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK)
        throw std::runtime_error("Unable to write PNG: deflate failed");
    compressed.resize(size);

    std::vector<unsigned char> file = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> header;
    put32be(header, width);
    put32be(header, height);
    header.insert(header.end(), {8, 6, 0, 0, 0});  // 8 bits, RGBA, deflate, adaptive filters, no interlace
    chunk(file, "IHDR", header);
    chunk(file, "IDAT", compressed);
    chunk(file, "IEND", {});
    writeFile(path, file.data(), file.size(), "PNG");
}

void writeExr(const std::filesystem::path& path, std::uint32_t width, std::uint32_t height,
              std::span<const float> pixels) {
    const auto stride = static_cast<std::size_t>(width) * 4;
    if (pixels.size() != stride * height)
        throw std::runtime_error("Invalid RGBA image size");
    std::vector<unsigned char> file;
    auto put32 = [&](std::uint32_t v) {
        for (int k = 0; k < 4; ++k)
            file.push_back(static_cast<unsigned char>(v >> (8 * k)));
    };
    auto put64 = [&](std::uint64_t v) {
        for (int k = 0; k < 8; ++k)
            file.push_back(static_cast<unsigned char>(v >> (8 * k)));
    };
    auto text = [&](const char* s) { file.insert(file.end(), s, s + std::strlen(s) + 1); };
    auto attribute = [&](const char* name, const char* type, std::uint32_t size) {
        text(name);
        text(type);
        put32(size);
    };
    auto putFloat = [&](float v) {
        std::uint32_t bits;
        std::memcpy(&bits, &v, 4);
        put32(bits);
    };
    put32(20000630U);  // magic
    put32(2U);         // version 2, single-part scan-line image
    attribute("channels", "chlist", 4 * 18 + 1);
    for (const char* name : {"A", "B", "G", "R"}) {  // channels are stored in alphabetical order
        text(name);
        put32(1U);  // HALF
        put32(0U);  // pLinear + reserved
        put32(1U);
        put32(1U);  // sampling
    }
    file.push_back(0);
    attribute("compression", "compression", 1);
    file.push_back(0);  // none
    attribute("dataWindow", "box2i", 16);
    put32(0U), put32(0U), put32(width - 1U), put32(height - 1U);
    attribute("displayWindow", "box2i", 16);
    put32(0U), put32(0U), put32(width - 1U), put32(height - 1U);
    attribute("lineOrder", "lineOrder", 1);
    file.push_back(0);  // increasing y
    attribute("pixelAspectRatio", "float", 4);
    putFloat(1.0f);
    attribute("screenWindowCenter", "v2f", 8);
    putFloat(0.0f), putFloat(0.0f);
    attribute("screenWindowWidth", "float", 4);
    putFloat(1.0f);
    file.push_back(0);  // end of header

    const std::size_t lineBytes = static_cast<std::size_t>(width) * 4 * 2;
    const std::size_t tableStart = file.size();
    const std::size_t dataStart = tableStart + static_cast<std::size_t>(height) * 8;
    for (std::uint32_t y = 0; y < height; ++y)
        put64(dataStart + static_cast<std::size_t>(y) * (8 + lineBytes));
    static const int channelOf[4] = {3, 2, 1, 0};  // A, B, G, R out of R, G, B, A
    for (std::uint32_t y = 0; y < height; ++y) {
        const float* row = pixels.data() + (height - y - 1) * stride;  // file row 0 is the top of the picture
        put32(y);
        put32(static_cast<std::uint32_t>(lineBytes));
        for (int c = 0; c < 4; ++c)
            for (std::uint32_t x = 0; x < width; ++x) {
                const std::uint16_t h = toHalf(row[static_cast<std::size_t>(x) * 4 + channelOf[c]]);
                file.push_back(static_cast<unsigned char>(h & 0xffU));
                file.push_back(static_cast<unsigned char>(h >> 8));
            }
    }
    writeFile(path, file.data(), file.size(), "EXR");
}

}  // namespace rt
