#include "image_io.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>

namespace elaina {

// ---- PNG -------------------------------------------------------------------------------------
static uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}

static void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

static void png_chunk(std::ofstream &f, const char type[4], const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    f.write(reinterpret_cast<const char *>(head.data()), 4);
    f.write(type, 4);
    if (!data.empty()) f.write(reinterpret_cast<const char *>(data.data()), (std::streamsize)data.size());
    uint32_t crc = crc32_update(0xffffffffu, reinterpret_cast<const uint8_t *>(type), 4);
    crc = crc32_update(crc, data.data(), data.size()) ^ 0xffffffffu;
    std::vector<uint8_t> tail;
    put_be32(tail, crc);
    f.write(reinterpret_cast<const char *>(tail.data()), 4);
}

void write_png(const fs::path &path, int width, int height, const std::vector<float> &rgb)
{
    if ((size_t)width * height * 3 != rgb.size()) throw std::runtime_error("write_png: size mismatch");
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot write " + path.string());
    // raw scanlines: filter byte 0 + RGBA8, bottom row first (vertical flip)
    const size_t stride = 1 + (size_t)width * 4;
    std::vector<uint8_t> raw(stride * height);
    for (int y = 0; y < height; ++y) {
        uint8_t *row = raw.data() + stride * (size_t)y;
        const float *src = rgb.data() + (size_t)(height - 1 - y) * width * 3;
        row[0] = 0;
        for (int x = 0; x < width; ++x) {
            for (int c = 0; c < 3; ++c) {
                const float v = src[3 * x + c];
                const int q = std::isfinite(v) ? (int)std::min(std::max(v * 255.0f, -1.0f), 256.0f) : 0;
                row[1 + 4 * x + c] = (uint8_t)std::min(std::max(q, 0), 255);
            }
            row[1 + 4 * x + 3] = 255;
        }
    }
    // zlib stream of stored deflate blocks
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;   // adler32
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        for (size_t i = pos; i < pos + n; ++i) {
            a = (a + raw[i]) % 65521u;
            b = (b + a) % 65521u;
        }
        pos += n;
        if (last) break;
    }
    put_be32(z, (b << 16) | a);
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    f.write(reinterpret_cast<const char *>(sig), 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGBA
    png_chunk(f, "IHDR", ihdr);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
}

// ---- OpenEXR, single part, scan lines, no compression, half RGBA -------------------------------
uint16_t float_to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x007fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));   // inf / nan
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);                                           // overflow
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;                                                   // underflow
        m |= 0x00800000u;
        const int shift = 14 - e;
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;   // may carry into the exponent: correct
    return (uint16_t)(sign | h);
}

namespace {
struct Bytes {
    std::vector<uint8_t> v;
    void raw(const void *p, size_t n) { const uint8_t *b = static_cast<const uint8_t *>(p); v.insert(v.end(), b, b + n); }
    void u8(uint8_t x) { v.push_back(x); }
    void i32(int32_t x) { raw(&x, 4); }
    void f32(float x) { raw(&x, 4); }
    void u64(uint64_t x) { raw(&x, 8); }
    void str(const char *s) { raw(s, std::strlen(s) + 1); }
    void attr(const char *name, const char *type, int32_t size) { str(name); str(type); i32(size); }
};
}  // namespace

void write_exr(const fs::path &path, int width, int height, const std::vector<float> &rgb)
{
    if ((size_t)width * height * 3 != rgb.size()) throw std::runtime_error("write_exr: size mismatch");
    Bytes h;
    h.i32(20000630);   // magic
    h.i32(2);          // version 2, single-part scan lines
    // channels, alphabetical: A B G R, all HALF (pixel type 1), linear, sampling 1 1
    h.attr("channels", "chlist", 4 * (2 + 4 + 4 + 4 + 4) + 1);
    for (const char *c : {"A", "B", "G", "R"}) {
        h.str(c);
        h.i32(1);
        h.u8(0); h.u8(0); h.u8(0); h.u8(0);
        h.i32(1); h.i32(1);
    }
    h.u8(0);
    h.attr("compression", "compression", 1); h.u8(0);
    h.attr("dataWindow", "box2i", 16); h.i32(0); h.i32(0); h.i32(width - 1); h.i32(height - 1);
    h.attr("displayWindow", "box2i", 16); h.i32(0); h.i32(0); h.i32(width - 1); h.i32(height - 1);
    h.attr("lineOrder", "lineOrder", 1); h.u8(0);
    h.attr("pixelAspectRatio", "float", 4); h.f32(1.0f);
    h.attr("screenWindowCenter", "v2f", 8); h.f32(0.0f); h.f32(0.0f);
    h.attr("screenWindowWidth", "float", 4); h.f32(1.0f);
    h.u8(0);           // end of header
    const size_t line_bytes = (size_t)width * 4 * 2;
    const uint64_t table_at = h.v.size();
    const uint64_t first_line = table_at + 8ull * (uint64_t)height;
    for (int y = 0; y < height; ++y) h.u64(first_line + (uint64_t)y * (8 + line_bytes));
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot write " + path.string());
    f.write(reinterpret_cast<const char *>(h.v.data()), (std::streamsize)h.v.size());
    std::vector<uint16_t> line((size_t)width * 4);
    const uint16_t one = float_to_half(1.0f);
    for (int y = 0; y < height; ++y) {
        const float *src = rgb.data() + (size_t)(height - 1 - y) * width * 3;   // vertical flip
        for (int x = 0; x < width; ++x) {
            line[x] = one;                                            // A
            line[(size_t)width + x] = float_to_half(src[3 * x + 2]);      // B
            line[2 * (size_t)width + x] = float_to_half(src[3 * x + 1]);  // G
            line[3 * (size_t)width + x] = float_to_half(src[3 * x]);      // R
        }
        const int32_t yy = y, nb = (int32_t)line_bytes;
        f.write(reinterpret_cast<const char *>(&yy), 4);
        f.write(reinterpret_cast<const char *>(&nb), 4);
        f.write(reinterpret_cast<const char *>(line.data()), (std::streamsize)line_bytes);
    }
}

// ---- PFM ---------------------------------------------------------------------------------------
void write_pfm(const fs::path &path, int width, int height, const std::vector<float> &rgb)
{
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot write " + path.string());
    f << "PF\n" << width << " " << height << "\n-1.0\n";  // little endian, rows top to bottom as stored
    f.write(reinterpret_cast<const char *>(rgb.data()), (std::streamsize)(rgb.size() * sizeof(float)));
}

}  // namespace elaina
