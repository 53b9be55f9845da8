// Image writers of the command-line renderer: same two functions as the
// reference's src/image.hpp:9-14.
#pragma once

#include <cstdint>
#include <filesystem>
#include <span>

namespace rt {

// RGBA8, row 0 = bottom row of the picture (the renderer's order); the file is written top row first.
void writePng(const std::filesystem::path& path, std::uint32_t width, std::uint32_t height,
              std::span<const std::uint8_t> pixels);
// Linear RGBA float, row 0 = bottom row; written as a scan-line OpenEXR file with half-float A, B, G, R channels.
void writeExr(const std::filesystem::path& path, std::uint32_t width, std::uint32_t height,
              std::span<const float> pixels);

}  // namespace rt
