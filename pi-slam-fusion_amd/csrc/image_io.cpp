// image_io.cpp -- output side of save(): the reference hands the mosaic to
// cv::imwrite (MultiBandMap2DCPU.cpp:841).  PNG (8-bit RGB, zlib stream split
// over IDAT chunks) when the name ends in .png, binary PPM otherwise.
#include "../../include/pifusion.h"
#include "jpeg_decode.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
#include <zlib.h>

namespace pf {

bool write_image_file(const char* filename, const uint8_t* bgr, int rows, int cols);      // also declared in fusion_map.hpp (save())

static void put_be32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

static bool png_chunk(FILE* f, const char* tag, const uint8_t* data, uint32_t len)
{
    uint8_t hdr[8];
    put_be32(hdr, len); std::memcpy(hdr + 4, tag, 4);
    uint32_t crc = crc32(0L, (const Bytef*)tag, 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t tail[4]; put_be32(tail, crc);
    return std::fwrite(hdr, 1, 8, f) == 8 && (!len || std::fwrite(data, 1, len, f) == len) && std::fwrite(tail, 1, 4, f) == 4;
}

static bool write_png(FILE* f, const uint8_t* bgr, int rows, int cols)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (std::fwrite(sig, 1, 8, f) != 8) return false;
    uint8_t ihdr[13];
    put_be32(ihdr, (uint32_t)cols); put_be32(ihdr + 4, (uint32_t)rows);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    if (!png_chunk(f, "IHDR", ihdr, 13)) return false;
    z_stream zs{};
    if (deflateInit(&zs, 1) != Z_OK) return false;
    std::vector<uint8_t> line((size_t)cols * 3 + 1), out(1 << 20);
    bool ok = true;
    for (int y = 0; y < rows && ok; y++) {
        line[0] = 0;
        const uint8_t* s = bgr + (size_t)y * cols * 3;
        for (int x = 0; x < cols; x++) { line[1 + 3 * x] = s[3 * x + 2]; line[2 + 3 * x] = s[3 * x + 1]; line[3 + 3 * x] = s[3 * x]; }
        zs.next_in = line.data(); zs.avail_in = (uInt)line.size();
        const int flush = (y == rows - 1) ? Z_FINISH : Z_NO_FLUSH;
        do {
            zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
            const int r = deflate(&zs, flush);
            if (r == Z_STREAM_ERROR) { ok = false; break; }
            const uint32_t have = (uint32_t)(out.size() - zs.avail_out);
            if (have && !png_chunk(f, "IDAT", out.data(), have)) { ok = false; break; }
        } while (zs.avail_out == 0);
    }
    deflateEnd(&zs);
    return ok && png_chunk(f, "IEND", nullptr, 0);
}

bool write_image_file(const char* filename, const uint8_t* bgr, int rows, int cols)
{
    FILE* f = std::fopen(filename, "wb");
    if (!f) { set_error(std::string("save: cannot open ") + filename); return false; }
    const size_t n = std::strlen(filename);
    bool ok;
    if (n >= 4 && (!std::strcmp(filename + n - 4, ".png") || !std::strcmp(filename + n - 4, ".PNG"))) ok = write_png(f, bgr, rows, cols);
    else {
        std::fprintf(f, "P6\n%d %d\n255\n", cols, rows);
        std::vector<uint8_t> line((size_t)cols * 3);
        ok = true;
        for (int y = 0; y < rows && ok; y++) {
            const uint8_t* s = bgr + (size_t)y * cols * 3;
            for (int x = 0; x < cols; x++) { line[3 * x] = s[3 * x + 2]; line[3 * x + 1] = s[3 * x + 1]; line[3 * x + 2] = s[3 * x]; }
            ok = std::fwrite(line.data(), 1, line.size(), f) == line.size();
        }
    }
    std::fclose(f);
    if (!ok) set_error("save: write failed");
    return ok;
}

}  // namespace pf

// the C ABI of the image files (include/pifusion.h): cv::imwrite / cv::imread of the file driver
extern "C" {
int pf_write_image(const char* filename, const uint8_t* bgr, int rows, int cols)
{ return filename && bgr && rows > 0 && cols > 0 && pf::write_image_file(filename, bgr, rows, cols); }
int pf_image_info(const char* filename, int* rows, int* cols)
{
    if (!filename || !rows || !cols) return 0;
    std::vector<uint8_t> b;
    if (!pf::read_file_bytes(filename, b)) return 0;
    if (b.size() >= 2 && b[0] == 0xFF && b[1] == 0xD8) return pf::jpeg_info(b.data(), b.size(), rows, cols, nullptr);
    if (b.size() >= 8 && b[0] == 0x89 && b[1] == 'P' && b[2] == 'N' && b[3] == 'G') return pf::png_info(b.data(), b.size(), rows, cols);
    std::vector<uint8_t> px;                                                  // PPM: the header is all there is to parse
    return pf::read_image_file(filename, px, rows, cols);
}
int pf_read_image(const char* filename, uint8_t* bgr, int rows, int cols)
{
    if (!filename || !bgr) return 0;
    std::vector<uint8_t> px; int r = 0, c = 0;
    if (!pf::read_image_file(filename, px, &r, &c)) return 0;
    if (r != rows || c != cols) { pf::set_error("pf_read_image: the buffer does not have the image's size"); return 0; }
    std::memcpy(bgr, px.data(), px.size());
    return 1;
}
}  // extern "C"
