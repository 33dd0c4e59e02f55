// png_decode.cpp -- cv::imread's PNG leg for the file driver (backup/map2dfusion.cpp:129-132 reads whatever the dataset holds; the library's own
// save() writes PNG, image_io.cpp): chunks with their CRCs, the zlib stream, the five scanline filters, and the conversion cv::imread's default
// flag (IMREAD_COLOR) makes -- three 8-bit channels, BGR: grey replicated, palette looked up, alpha dropped, 16-bit samples cut to their high
// byte, 1 / 2 / 4-bit samples scaled to 8.  Interlaced (Adam7) files are refused.  PNG is lossless: the pixels are the file's, pinned against
// Pillow in tests/test_png.py.  Host code only.
#include "jpeg_decode.hpp"
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <zlib.h>

namespace pf {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
bool bad(const char* m) { set_error(std::string("png: ") + m); return false; }

inline uint8_t paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (uint8_t)(pa <= pb && pa <= pc ? a : pb <= pc ? b : c);
}

struct Header { int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0; };

bool parse(const uint8_t* d, size_t n, Header& H, std::vector<uint8_t>* idat, uint8_t pal[256][3], int* npal)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (n < 8 + 25 || std::memcmp(d, sig, 8)) return bad("not a PNG stream");
    size_t pos = 8; bool have_hdr = false, end = false;
    while (!end) {
        if (pos + 12 > n) return bad("truncated chunk");
        const uint32_t len = be32(d + pos);
        const uint8_t* tag = d + pos + 4;
        if (len > n - pos - 12) return bad("truncated chunk");
        const uint8_t* data = d + pos + 8;
        if ((uint32_t)crc32(crc32(0L, tag, 4), data, len) != be32(data + len)) return bad("chunk CRC mismatch");
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len != 13) return bad("bad IHDR");
            H.w = (int)be32(data); H.h = (int)be32(data + 4); H.depth = data[8]; H.ctype = data[9]; H.interlace = data[12];
            if (H.w <= 0 || H.h <= 0 || (unsigned long long)H.w * (unsigned long long)H.h > (1ull << 30)) return bad("bad image size");
            if (data[10] || data[11]) return bad("unknown compression or filter method");
            const int dpt = H.depth, ct = H.ctype;
            const bool ok = (ct == 0 && (dpt == 1 || dpt == 2 || dpt == 4 || dpt == 8 || dpt == 16)) || (ct == 3 && (dpt == 1 || dpt == 2 || dpt == 4 || dpt == 8)) ||
                            ((ct == 2 || ct == 4 || ct == 6) && (dpt == 8 || dpt == 16));
            if (!ok) return bad("colour type and bit depth do not go together");
            have_hdr = true;
            if (!idat) return true;
        } else if (!have_hdr) return bad("IHDR is not the first chunk");
        else if (!std::memcmp(tag, "PLTE", 4)) {
            if (len % 3 || len > 768) return bad("bad PLTE");
            *npal = (int)(len / 3);
            for (int i = 0; i < *npal; i++) { pal[i][0] = data[3 * i]; pal[i][1] = data[3 * i + 1]; pal[i][2] = data[3 * i + 2]; }
        } else if (!std::memcmp(tag, "IDAT", 4)) idat->insert(idat->end(), data, data + len);
        else if (!std::memcmp(tag, "IEND", 4)) end = true;
        else if (!(tag[0] & 0x20)) return bad("unknown critical chunk");
        pos += 12 + (size_t)len;
    }
    return have_hdr ? true : bad("no IHDR");
}

}  // namespace

bool png_info(const uint8_t* data, size_t len, int* rows, int* cols)
{
    Header H; int np = 0;
    if (!data || !parse(data, len, H, nullptr, nullptr, &np)) return false;
    if (rows) *rows = H.h;
    if (cols) *cols = H.w;
    return true;
}

bool png_decode_bgr(const uint8_t* data, size_t len, uint8_t* bgr, int rows, int cols, size_t stride)
{
    if (!data || !bgr) return bad("null buffer");
    try {
        Header H; std::vector<uint8_t> z; uint8_t pal[256][3]; int npal = 0;
        std::memset(pal, 0, sizeof(pal));
        if (!parse(data, len, H, &z, pal, &npal)) return false;
        if (H.h != rows || H.w != cols || stride < (size_t)cols * 3) return bad("the output buffer does not have the image's size");
        if (H.interlace) return bad("interlaced files are not supported");
        if (H.ctype == 3 && !npal) return bad("palette image without PLTE");
        const int ch = H.ctype == 0 ? 1 : H.ctype == 2 ? 3 : H.ctype == 3 ? 1 : H.ctype == 4 ? 2 : 4;
        const size_t bpp_bits = (size_t)ch * H.depth, line = ((size_t)H.w * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
        std::vector<uint8_t> raw((line + 1) * (size_t)H.h);
        uLongf got = (uLongf)raw.size();
        const int zr = uncompress(raw.data(), &got, z.data(), (uLong)z.size());
        if (zr != Z_OK || got != raw.size()) return bad("the zlib stream does not hold the image");
        std::vector<uint8_t> prev(line, 0);
        for (int y = 0; y < H.h; y++) {
            uint8_t* r = raw.data() + (line + 1) * (size_t)y;
            const int ft = r[0];
            uint8_t* c = r + 1;
            switch (ft) {
            case 0: break;
            case 1: for (size_t i = bpp; i < line; i++) c[i] = (uint8_t)(c[i] + c[i - bpp]); break;
            case 2: for (size_t i = 0; i < line; i++) c[i] = (uint8_t)(c[i] + prev[i]); break;
            case 3: for (size_t i = 0; i < line; i++) c[i] = (uint8_t)(c[i] + (((i >= bpp ? c[i - bpp] : 0) + prev[i]) >> 1)); break;
            case 4: for (size_t i = 0; i < line; i++) c[i] = (uint8_t)(c[i] + paeth(i >= bpp ? c[i - bpp] : 0, prev[i], i >= bpp ? prev[i - bpp] : 0)); break;
            default: return bad("unknown filter type");
            }
            std::memcpy(prev.data(), c, line);
            uint8_t* o = bgr + (size_t)y * stride;
            const int step = H.depth == 16 ? 2 : 1;                   // 16-bit samples: the high byte (cv::imread's conversion to 8 bits)
            for (int x = 0; x < H.w; x++) {
                int v[4] = { 0, 0, 0, 0 };
                if (H.depth >= 8) for (int k = 0; k < ch; k++) v[k] = c[((size_t)x * ch + k) * step];
                else {
                    const size_t bit = (size_t)x * H.depth;
                    v[0] = (c[bit >> 3] >> (8 - H.depth - (bit & 7))) & ((1 << H.depth) - 1);
                    if (H.ctype == 0) v[0] = v[0] * 255 / ((1 << H.depth) - 1);
                }
                if (H.ctype == 3) { if (v[0] >= npal) return bad("palette index out of range"); o[3 * x] = pal[v[0]][2]; o[3 * x + 1] = pal[v[0]][1]; o[3 * x + 2] = pal[v[0]][0]; }
                else if (ch <= 2) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = (uint8_t)v[0];
                else { o[3 * x] = (uint8_t)v[2]; o[3 * x + 1] = (uint8_t)v[1]; o[3 * x + 2] = (uint8_t)v[0]; }
            }
        }
        return true;
    } catch (const std::bad_alloc&) { return bad("out of memory"); }
}

}  // namespace pf
