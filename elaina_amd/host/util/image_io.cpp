#include "image_io.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
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

// ---- PNG reader: zlib inflate (RFC 1950/1951) + unfiltering (PNG 1.2 section 6) ------------------
namespace {
struct BitReader {
    const uint8_t *p;
    size_t n, pos = 0;
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    uint32_t bits(int k)
    {
        while (bitcnt < k) {
            if (pos >= n) throw std::runtime_error("png: truncated deflate stream");
            bitbuf |= (uint32_t)p[pos++] << bitcnt;
            bitcnt += 8;
        }
        const uint32_t v = bitbuf & ((k == 32) ? 0xffffffffu : ((1u << k) - 1u));
        bitbuf >>= k;
        bitcnt -= k;
        return v;
    }
    void align() { bitbuf = 0; bitcnt = 0; }
};

struct Huffman {
    uint16_t count[16] = {0}, symbol[288] = {0};
    void build(const uint8_t *lengths, int n)
    {
        for (int i = 0; i < 16; ++i) count[i] = 0;
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = (uint16_t)(offs[i] + count[i]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)br.bits(1);
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw std::runtime_error("png: bad Huffman code");
    }
};

std::vector<uint8_t> inflate_zlib(const std::vector<uint8_t> &z)
{
    if (z.size() < 6 || (z[0] & 0x0f) != 8) throw std::runtime_error("png: not a zlib stream");
    BitReader br{z.data() + 2, z.size() - 2};
    std::vector<uint8_t> out;
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131,
                                       163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                                       2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) throw std::runtime_error("png: truncated stored block");
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) throw std::runtime_error("png: truncated stored block");
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lengths[i] = 8;
                for (int i = 144; i < 256; ++i) lengths[i] = 9;
                for (int i = 256; i < 280; ++i) lengths[i] = 7;
                for (int i = 280; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman code;
                code.build(cl, 19);
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = code.decode(br);
                    if (sym < 16) lengths[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) throw std::runtime_error("png: bad code lengths");
                            val = lengths[i - 1];
                            rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (i + rep > nlen + ndist) throw std::runtime_error("png: bad code lengths");
                        while (rep--) lengths[i++] = (uint8_t)val;
                    }
                }
                lit.build(lengths, nlen);
                dist.build(lengths + nlen, ndist);
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    if (sym > 285) throw std::runtime_error("png: bad length symbol");
                    const int len = lbase[sym - 257] + (int)br.bits(lext[sym - 257]);
                    const int ds = dist.decode(br);
                    if (ds > 29) throw std::runtime_error("png: bad distance symbol");
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (d > out.size()) throw std::runtime_error("png: distance too far back");
                    for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
                }
            }
        } else {
            throw std::runtime_error("png: bad block type");
        }
        if (last) break;
    }
    return out;
}
}  // namespace

void read_png(const fs::path &path, int *width, int *height, std::vector<uint8_t> *rgba)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path.string());
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (b.size() < 8 || std::memcmp(b.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file: " + path.string());
    auto be32 = [&](size_t o) { return ((uint32_t)b[o] << 24) | ((uint32_t)b[o + 1] << 16) | ((uint32_t)b[o + 2] << 8) | b[o + 3]; };
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    while (pos + 12 <= b.size()) {
        const uint32_t n = be32(pos);
        const std::string type(reinterpret_cast<const char *>(&b[pos + 4]), 4);
        if (pos + 12 + n > b.size()) throw std::runtime_error("png: truncated chunk");
        const uint8_t *d = &b[pos + 8];
        if (type == "IHDR") {
            // the header must be the first chunk, exactly 13 bytes long, and describe a sane image
            // before any size is computed from it (mask images are user input)
            if (pos != 8 || n != 13) throw std::runtime_error("png: malformed IHDR chunk");
            w = be32(pos + 8); h = be32(pos + 12);
            if (w == 0 || h == 0 || w > (1u << 16) || h > (1u << 16)) throw std::runtime_error("png: image dimensions out of range");
            depth = d[8]; ctype = d[9]; interlace = d[12];
        } else if (pos == 8) {
            throw std::runtime_error("png: the first chunk is not IHDR");
        } else if (type == "PLTE") plte.assign(d, d + n);
        else if (type == "tRNS") trns.assign(d, d + n);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + n);
        else if (type == "IEND") break;
        pos += 12 + n;
    }
    if (w == 0 || h == 0 || depth != 8 || interlace != 0) throw std::runtime_error("png: only 8-bit non-interlaced images are read");
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) throw std::runtime_error("png: unknown colour type");
    std::vector<uint8_t> raw = inflate_zlib(idat);
    const size_t stride = (size_t)w * ch;
    if (raw.size() < (stride + 1) * h) throw std::runtime_error("png: image data too short");
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = &raw[(stride + 1) * y];
        uint8_t *dst = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        const int ft = src[0];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)ch ? dst[x - ch] : 0, bb = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = bb;
            else if (ft == 3) pred = (a + bb) >> 1;
            else if (ft == 4) {
                const int p = a + bb - c, pa = std::abs(p - a), pb = std::abs(p - bb), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
            } else if (ft != 0) throw std::runtime_error("png: bad filter type");
            dst[x] = (uint8_t)(src[1 + x] + pred);
        }
    }
    *width = (int)w;
    *height = (int)h;
    rgba->assign((size_t)w * h * 4, 255);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        uint8_t *o = &(*rgba)[4 * i];
        const uint8_t *s = &img[i * ch];
        if (ctype == 0) { o[0] = o[1] = o[2] = s[0]; }
        else if (ctype == 2) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
        else if (ctype == 3) {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) throw std::runtime_error("png: palette index out of range");
            o[0] = plte[3 * s[0]]; o[1] = plte[3 * s[0] + 1]; o[2] = plte[3 * s[0] + 2];
            if (s[0] < trns.size()) o[3] = trns[s[0]];
        } else if (ctype == 4) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
        else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; }
    }
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


// ---- readers of the float formats (mask images, reference Image::loadImage core/texture.cu:26-80) ----------------------
float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else {      // subnormal: renormalise
            int k = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) { mm <<= 1; ++k; }
            bits = sign | ((uint32_t)(113 - k) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// PFM ("PF" colour, "Pf" grey; scale < 0 = little endian).  Rows are returned in file order.
void read_pfm(const fs::path &path, int *width, int *height, std::vector<float> *rgb)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path.string());
    std::string magic;
    int w = 0, h = 0;
    double scale = 0;
    f >> magic >> w >> h >> scale;
    if ((magic != "PF" && magic != "Pf") || w <= 0 || h <= 0 || w > (1 << 16) || h > (1 << 16) || scale == 0 || !f)
        throw std::runtime_error("not a PFM file: " + path.string());
    f.get();        // the single whitespace byte after the scale
    const int ch = magic == "PF" ? 3 : 1;
    std::vector<uint8_t> raw((size_t)w * h * ch * 4);
    f.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)raw.size());
    if ((size_t)f.gcount() != raw.size()) throw std::runtime_error("pfm: file too short: " + path.string());
    rgb->assign((size_t)w * h * 3, 0.0f);
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int c = 0; c < 3; ++c) {
            const uint8_t *b = &raw[(i * ch + (ch == 3 ? c : 0)) * 4];
            const uint32_t u = scale < 0 ? ((uint32_t)b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24))
                                         : ((uint32_t)b[3] | (b[2] << 8) | (b[1] << 16) | ((uint32_t)b[0] << 24));
            std::memcpy(&(*rgb)[3 * i + c], &u, 4);
        }
    *width = w;
    *height = h;
}

// OpenEXR: single part, scan lines, compression NONE / ZIPS / ZIP, HALF or FLOAT channels; R, G, B (or Y) are
// returned as floats, rows top to bottom (increasing y), missing channels 0.
void read_exr(const fs::path &path, int *width, int *height, std::vector<float> *rgb)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path.string());
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    // (written so that neither an offset near 2^64 taken from the file nor a huge count can wrap the comparison)
    auto need = [&](size_t n) { if (pos > b.size() || n > b.size() - pos) throw std::runtime_error("exr: truncated file: " + path.string()); };
    auto i32 = [&]() { need(4); int32_t v; std::memcpy(&v, &b[pos], 4); pos += 4; return v; };
    auto str = [&]() { std::string s; for (;;) { need(1); const char c = (char)b[pos++]; if (!c) break; s.push_back(c); if (s.size() > 255) throw std::runtime_error("exr: bad string"); } return s; };
    if (i32() != 20000630) throw std::runtime_error("not an OpenEXR file: " + path.string());
    const int32_t version = i32();
    if ((version & 0xff) != 2 || (version & 0x1a00)) throw std::runtime_error("exr: only single-part scan-line files are read");
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1, line_order = 0;
    for (;;) {
        const std::string name = str();
        if (name.empty()) break;
        const std::string type = str();
        const int32_t size = i32();
        if (size < 0) throw std::runtime_error("exr: bad attribute size");
        need((size_t)size);
        const size_t end = pos + (size_t)size;
        if (name == "channels") {
            while (pos < end) {
                const std::string cn = str();
                if (cn.empty()) break;
                const int t = i32();
                pos += 4;                                   // pLinear + reserved
                const int xs = i32(), ys = i32();
                if (xs != 1 || ys != 1) throw std::runtime_error("exr: subsampled channels are not read");
                chans.push_back({cn, t});
            }
        } else if (name == "compression") compression = b[pos];
        else if (name == "dataWindow") { x0 = i32(); y0 = i32(); x1 = i32(); y1 = i32(); }
        else if (name == "lineOrder") line_order = b[pos];
        pos = end;
    }
    const int64_t w = (int64_t)x1 - x0 + 1, h = (int64_t)y1 - y0 + 1;
    if (w <= 0 || h <= 0 || w > (1 << 16) || h > (1 << 16) || chans.empty()) throw std::runtime_error("exr: bad header: " + path.string());
    if (compression != 0 && compression != 2 && compression != 3) throw std::runtime_error("exr: only NONE / ZIPS / ZIP compression is read");
    (void)line_order;       // the offset table is indexed by y either way
    const int lines_per_block = compression == 3 ? 16 : 1;
    const int64_t n_blocks = (h + lines_per_block - 1) / lines_per_block;
    size_t line_bytes = 0;
    for (const Chan &c : chans) {
        if (c.type != 1 && c.type != 2) throw std::runtime_error("exr: only HALF and FLOAT channels are read");
        line_bytes += (size_t)w * (c.type == 1 ? 2 : 4);
    }
    need((size_t)n_blocks * 8);
    std::vector<uint64_t> offs((size_t)n_blocks);
    std::memcpy(offs.data(), &b[pos], (size_t)n_blocks * 8);
    rgb->assign((size_t)w * h * 3, 0.0f);
    const size_t table_end = pos + (size_t)n_blocks * 8;      // the blocks lie behind the header and the offset table
    for (int64_t k = 0; k < n_blocks; ++k) {
        if (offs[(size_t)k] < (uint64_t)table_end || offs[(size_t)k] >= (uint64_t)b.size()) throw std::runtime_error("exr: block offset outside the file: " + path.string());
        pos = (size_t)offs[(size_t)k];
        const int32_t y = i32(), nbytes = i32();
        if (nbytes < 0 || y < y0 || y > y1) throw std::runtime_error("exr: bad block");
        need((size_t)nbytes);
        const int rows = (int)std::min<int64_t>(lines_per_block, (int64_t)y1 - y + 1);
        const size_t want = line_bytes * (size_t)rows;
        std::vector<uint8_t> px;
        if (compression == 0 || (size_t)nbytes == want) px.assign(&b[pos], &b[pos] + nbytes);
        else {
            std::vector<uint8_t> z(&b[pos], &b[pos] + nbytes), t = inflate_zlib(z);
            if (t.size() != want) throw std::runtime_error("exr: block inflates to the wrong size");
            for (size_t i = 1; i < t.size(); ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);      // the predictor
            px.resize(want);                                                                      // then de-interleave the halves
            const size_t half = (want + 1) / 2;
            for (size_t i = 0; i < want; ++i) px[i] = (i & 1) ? t[half + i / 2] : t[i / 2];
        }
        if (px.size() < want) throw std::runtime_error("exr: block too short");
        for (int r = 0; r < rows; ++r) {
            const uint8_t *lp = &px[line_bytes * (size_t)r];
            float *dst = &(*rgb)[(size_t)(y - y0 + r) * (size_t)w * 3];
            for (const Chan &c : chans) {            // channels are stored one after the other within a line, alphabetically
                const int slot = (c.name == "R") ? 0 : (c.name == "G") ? 1 : (c.name == "B") ? 2 : (c.name == "Y") ? 3 : -1;
                for (int64_t x = 0; x < w; ++x) {
                    float v;
                    if (c.type == 1) { uint16_t hv; std::memcpy(&hv, lp + 2 * x, 2); v = half_to_float(hv); }
                    else std::memcpy(&v, lp + 4 * x, 4);
                    if (slot >= 0 && slot < 3) dst[3 * x + slot] = v;
                    else if (slot == 3) dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = v;
                }
                lp += (size_t)w * (c.type == 1 ? 2 : 4);
            }
        }
    }
    *width = (int)w;
    *height = (int)h;
}

// Radiance RGBE picture (.hdr; stb_image's stbi__hdr_load is what the reference's Image::loadImage reaches for it,
// core/texture.cu:26-71): text header up to an empty line, "-Y h +X w", then flat RGBE pixels or the run-length scan lines
// of the "new" format (2 2 hi lo, then the four channels one after the other).  value = mantissa * 2^(e - 136), all four
// bytes zero-exponent = black: exact arithmetic, so the decoded floats are the reference's.  Rows top to bottom.
void read_hdr(const fs::path &path, int *width, int *height, std::vector<float> *rgb)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path.string());
    const std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto line = [&]() {
        std::string l;
        while (pos < b.size() && b[pos] != '\n') {
            l.push_back((char)b[pos++]);
            if (l.size() > 1024) throw std::runtime_error("hdr: header line too long: " + path.string());
        }
        if (pos >= b.size()) throw std::runtime_error("hdr: truncated header: " + path.string());
        ++pos;
        return l;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") throw std::runtime_error("not a Radiance picture: " + path.string());
    bool rgbe = false;
    for (;;) {
        const std::string l = line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") rgbe = true;
    }
    if (!rgbe) throw std::runtime_error("hdr: only FORMAT=32-bit_rle_rgbe is read: " + path.string());
    const std::string res = line();
    long h = 0, w = 0;
    if (std::sscanf(res.c_str(), "-Y %ld +X %ld", &h, &w) != 2 || w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24))
        throw std::runtime_error("hdr: only the -Y h +X w orientation is read: " + path.string());
    auto need = [&](size_t n) { if (pos > b.size() || n > b.size() - pos) throw std::runtime_error("hdr: truncated file: " + path.string()); };
    bool flat = w < 8 || w >= 32768;
    if (!flat) {
        need(4);
        flat = b[pos] != 2 || b[pos + 1] != 2 || (b[pos + 2] & 0x80);      // no scan-line marker: flat pixels throughout
    }
    // before anything is allocated from the header's numbers (each up to 2^24): the file must be able to hold the picture -- flat
    // pixels whole, a run-length coded one at least its scan-line markers and two bytes per channel run (a run covers <= 127 pixels)
    need(flat ? (size_t)w * (size_t)h * 4 : (size_t)h * (4 + 4 * 2 * (((size_t)w + 126) / 127)));
    std::vector<uint8_t> px((size_t)w * h * 4);
    if (flat) {
        need(px.size());
        std::memcpy(px.data(), &b[pos], px.size());
    } else {
        std::vector<uint8_t> scan((size_t)w * 4);
        for (long y = 0; y < h; ++y) {
            need(4);
            if (b[pos] != 2 || b[pos + 1] != 2 || (((long)b[pos + 2] << 8) | b[pos + 3]) != w) throw std::runtime_error("hdr: bad scan line: " + path.string());
            pos += 4;
            for (int ch = 0; ch < 4; ++ch) {
                long i = 0;
                while (i < w) {
                    need(1);
                    int count = b[pos++];
                    if (count > 128) {
                        count -= 128;
                        need(1);
                        const uint8_t v = b[pos++];
                        if (count == 0 || i + count > w) throw std::runtime_error("hdr: bad run: " + path.string());
                        for (int k = 0; k < count; ++k) scan[(size_t)(i++) * 4 + ch] = v;
                    } else {
                        if (count == 0 || i + count > w) throw std::runtime_error("hdr: bad run: " + path.string());
                        need((size_t)count);
                        for (int k = 0; k < count; ++k) scan[(size_t)(i++) * 4 + ch] = b[pos++];
                    }
                }
            }
            std::memcpy(&px[(size_t)y * w * 4], scan.data(), scan.size());
        }
    }
    rgb->assign((size_t)w * h * 3, 0.0f);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t *q = &px[4 * i];
        if (q[3] != 0) {
            const float f1 = std::ldexp(1.0f, (int)q[3] - (128 + 8));
            (*rgb)[3 * i] = (float)q[0] * f1; (*rgb)[3 * i + 1] = (float)q[1] * f1; (*rgb)[3 * i + 2] = (float)q[2] * f1;
        }
    }
    *width = (int)w;
    *height = (int)h;
}

// A mask image of any format the build can read: a pixel is on when any of R, G, B is non-zero; rows bottom to top
// (the reference loads it flipped vertically, core/problem.cu:216-242).
void read_mask_image(const fs::path &path, int *width, int *height, std::vector<uint8_t> *mask)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path.string());
    uint8_t head[8] = {0};
    f.read(reinterpret_cast<char *>(head), 8);
    f.close();
    int w = 0, h = 0;
    if (head[0] == 0x89 && head[1] == 'P' && head[2] == 'N' && head[3] == 'G') {
        std::vector<uint8_t> rgba;
        read_png(path, &w, &h, &rgba);
        mask->assign((size_t)w * h, 0);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const uint8_t *px = &rgba[4 * ((size_t)(h - 1 - y) * w + x)];
                (*mask)[(size_t)y * w + x] = (px[0] | px[1] | px[2]) ? 1 : 0;
            }
    } else {
        std::vector<float> rgb;
        bool file_is_bottom_up = false;
        if (head[0] == 0x76 && head[1] == 0x2f && head[2] == 0x31 && head[3] == 0x01) read_exr(path, &w, &h, &rgb);
        else if (head[0] == 'P' && (head[1] == 'F' || head[1] == 'f')) { read_pfm(path, &w, &h, &rgb); file_is_bottom_up = true; }
        else if (head[0] == '#' && head[1] == '?') read_hdr(path, &w, &h, &rgb);
        // OUT OF SCOPE, stated: the lossy formats stb_image also decodes (JPEG above all).  A mask is "any non-zero byte", which a
        // lossy codec's ringing next to every edge decides by the last bit of ITS inverse transform: no second decoder reproduces
        // stb_image's there, and stb_image is absent from the reference checkout (ext/ is empty) -- convert such masks to PNG
        else throw std::runtime_error("mask image: PNG, OpenEXR, PFM and Radiance .hdr files are read here; JPEG and the other lossy or "
                                      "palette formats of stb_image are out of scope (convert the mask to PNG): " + path.string());
        mask->assign((size_t)w * h, 0);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                // PFM stores its rows bottom to top already; the others are flipped like the PNG
                const float *px = &rgb[3 * ((size_t)(file_is_bottom_up ? y : h - 1 - y) * w + x)];
                (*mask)[(size_t)y * w + x] = (px[0] != 0.0f || px[1] != 0.0f || px[2] != 0.0f) ? 1 : 0;
            }
    }
    *width = w;
    *height = h;
}

}  // namespace elaina
