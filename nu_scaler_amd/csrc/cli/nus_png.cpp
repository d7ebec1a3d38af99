// nus_png.cpp -- see nus_png.hpp.
#include "nus_png.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace nus_cli {

namespace {

const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

bool read_file(const std::string &path, std::vector<uint8_t> &out)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) {
        std::fclose(f);
        return false;
    }
    out.resize((size_t)n);
    const bool ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// PNG specification section 9: undo the per-row filters in place.  bpp = bytes per complete pixel.
bool unfilter(std::vector<uint8_t> &raw, uint32_t w, uint32_t h, uint32_t bpp, std::vector<uint8_t> &px)
{
    const size_t stride = (size_t)w * bpp;
    if (raw.size() < (size_t)h * (stride + 1)) return false;
    px.assign((size_t)h * stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(size_t)y * (stride + 1)];
        const uint8_t *line = &raw[(size_t)y * (stride + 1) + 1];
        uint8_t *cur = &px[(size_t)y * stride];
        const uint8_t *up = y ? &px[(size_t)(y - 1) * stride] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: pred = paeth(a, b, c); break;
            default: return false;
            }
            cur[i] = (uint8_t)(line[i] + pred);
        }
    }
    return true;
}

} // namespace

std::string read_png(const std::string &path, Image &img)
{
    std::vector<uint8_t> data;
    if (!read_file(path, data)) return path + ": cannot read file";
    if (data.size() < 8 || std::memcmp(data.data(), kSig, 8) != 0) return path + ": not a PNG file";
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = -1, ctype = -1, interlace = -1;
    std::vector<uint8_t> idat, palette, trns;
    while (pos + 12 <= data.size()) {
        const uint32_t n = be32(&data[pos]);
        if (pos + 12 + (size_t)n > data.size()) return path + ": truncated PNG chunk";
        const uint8_t *typ = &data[pos + 4], *body = &data[pos + 8];
        if (!std::memcmp(typ, "IHDR", 4) && n >= 13) {
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (!std::memcmp(typ, "PLTE", 4)) {
            palette.assign(body, body + n);
        } else if (!std::memcmp(typ, "tRNS", 4)) {
            trns.assign(body, body + n);
        } else if (!std::memcmp(typ, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + n);
        } else if (!std::memcmp(typ, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)n;
    }
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return path + ": bad PNG dimensions";
    uint32_t ch;
    switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: ch = 0;
    }
    if (depth != 8 || interlace != 0 || ch == 0)
        return path + ": unsupported PNG flavour (8-bit non-interlaced grey / RGB / palette / +alpha only)";
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * ch + 1));
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return path + ": corrupt PNG image data";
    std::vector<uint8_t> px;
    if (!unfilter(raw, w, h, ch, px)) return path + ": corrupt PNG filter bytes";
    img.width = w;
    img.height = h;
    img.rgba.assign((size_t)w * h * 4, 255);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        uint8_t *o = &img.rgba[i * 4];
        const uint8_t *p = &px[i * ch];
        switch (ctype) {
        case 6: std::memcpy(o, p, 4); break;
        case 2: std::memcpy(o, p, 3); break;
        case 0: o[0] = o[1] = o[2] = p[0]; break;
        case 4: o[0] = o[1] = o[2] = p[0]; o[3] = p[1]; break;
        default: { // palette
            if ((size_t)p[0] * 3 + 2 >= palette.size()) return path + ": palette index out of range";
            std::memcpy(o, &palette[(size_t)p[0] * 3], 3);
            if (p[0] < trns.size()) o[3] = trns[p[0]];
        }
        }
    }
    return "";
}

std::string write_png(const std::string &path, const Image &img)
{
    if (img.rgba.size() != (size_t)img.width * img.height * 4 || img.width == 0 || img.height == 0)
        return "write_png: buffer size does not match width * height * 4";
    const size_t stride = (size_t)img.width * 4;
    std::vector<uint8_t> raw((size_t)img.height * (stride + 1));
    for (uint32_t y = 0; y < img.height; ++y) {
        raw[(size_t)y * (stride + 1)] = 0; // filter type None
        std::memcpy(&raw[(size_t)y * (stride + 1) + 1], &img.rgba[(size_t)y * stride], stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return "write_png: compression failed";
    comp.resize(clen);
    std::vector<uint8_t> out(kSig, kSig + 8);
    auto chunk = [&](const char *typ, const std::vector<uint8_t> &body) {
        put_be32(out, (uint32_t)body.size());
        const size_t start = out.size();
        out.insert(out.end(), typ, typ + 4);
        out.insert(out.end(), body.begin(), body.end());
        put_be32(out, (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
    };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, img.width);
    put_be32(ihdr, img.height);
    const uint8_t tail[5] = {8, 6, 0, 0, 0};
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk("IHDR", ihdr);
    chunk("IDAT", comp);
    chunk("IEND", {});
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return path + ": cannot open for writing";
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? "" : path + ": short write";
}

} // namespace nus_cli
