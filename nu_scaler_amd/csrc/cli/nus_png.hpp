// nus_png.hpp -- minimal PNG codec for the command-line tool: 8-bit, non-interlaced, colour types
// 0 / 2 / 3 / 4 / 6 in, RGBA8 out; RGBA8 in, colour type 6 (filter 0) out.  zlib does the (de)compression
// and the CRCs; the reference uses the `image` crate here (`image::open(..)?.to_rgba8()`, `save`:
// Nu_scale/src/upscale/mod.rs:316-332).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace nus_cli {

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba; // width * height * 4
};

// Both return an empty string on success, else the error text.
std::string read_png(const std::string &path, Image &img);
std::string write_png(const std::string &path, const Image &img);

} // namespace nus_cli
