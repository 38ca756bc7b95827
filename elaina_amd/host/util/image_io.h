// image_io.h -- the file formats of the reference's exports (reference core/texture.cu:82-116,
// util/image.cpp:28-103): 8-bit RGBA PNG with value = clamp((int)(v * 255), 0, 255), and OpenEXR
// RGBA in half precision; both flipped vertically on write like stbi_flip_vertically_on_write /
// tinyexr::save_exr(.., flip = true).  Plus raw-float PFM (what parity is measured on) and a PNG
// reader for mask images.  Self-contained: no zlib / stb / tinyexr in this image.
#pragma once
#include <filesystem>
#include <vector>

namespace elaina {

namespace fs = std::filesystem;

// rgb: width*height*3 floats, rows top to bottom as stored in the channel buffers
void write_png(const fs::path &path, int width, int height, const std::vector<float> &rgb);
void write_exr(const fs::path &path, int width, int height, const std::vector<float> &rgb);
void write_pfm(const fs::path &path, int width, int height, const std::vector<float> &rgb);

uint16_t float_to_half(float f);   // round to nearest even, IEEE binary16

// 8-bit PNG reader (grey, grey+alpha, RGB, RGBA, palette; non-interlaced) for mask images
// (reference core/problem.cu:216-242 loads them through stb_image).  rgba: height*width*4 bytes,
// rows top to bottom as stored in the file.
void read_png(const fs::path &path, int *width, int *height, std::vector<uint8_t> *rgba);

// The float formats of Image::loadImage (reference core/texture.cu:26-80).  rgb: height*width*3 floats, rows in file order
// (PFM: bottom to top by the format's convention; EXR: top to bottom).  EXR: single part, scan lines, compression NONE / ZIPS /
// ZIP, HALF or FLOAT channels R G B (or Y).  JPEG and Radiance .hdr have no decoder in this build.
void read_pfm(const fs::path &path, int *width, int *height, std::vector<float> *rgb);
void read_exr(const fs::path &path, int *width, int *height, std::vector<float> *rgb);
// Radiance RGBE (.hdr), flat or run-length scan lines, rows top to bottom
void read_hdr(const fs::path &path, int *width, int *height, std::vector<float> *rgb);
float half_to_float(uint16_t h);

// mask_path of the scene configuration (reference core/problem.cu:216-242): any format above, by content; a pixel is on when
// any of R, G, B is non-zero; mask: height*width bytes, rows flipped vertically like the reference's load.
void read_mask_image(const fs::path &path, int *width, int *height, std::vector<uint8_t> *mask);

}  // namespace elaina
