// image_io.cpp -- output side of save(): the reference hands the mosaic to
// cv::imwrite (MultiBandMap2DCPU.cpp:841).  PNG (8-bit RGB, zlib stream split
// over IDAT chunks) when the name ends in .png, binary PPM otherwise.
#include "../../include/pifusion.h"
#include "jpeg_decode.hpp"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
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

// The zlib stream of the IDAT chunks is made band by band (kBandRows rows each) on a few threads: every band is a raw deflate stream of its
// own (no history before the band, ended on a byte boundary by Z_SYNC_FLUSH, the last one by Z_FINISH), and raw deflate streams laid end
// to end behind one zlib header are one valid zlib stream when the Adler-32 of all the bytes follows (adler32_combine) -- what pigz does with
// independent blocks.  The bands are fixed, so the file's bytes do not depend on the number of threads.  The reference hands the same
// pixels to cv::imwrite (MultiBandMap2DCPU.cpp:841), one thread.
constexpr int kBandRows = 256;

struct Band { std::unique_ptr<uint8_t[]> data; size_t size = 0; uLong adler = 1; int state = 0; };      // state: 0 pending, 1 done, -1 failed

static bool deflate_band(const uint8_t* bgr, int cols, int y0, int y1, bool last, Band& out)
{
    z_stream zs{};
    if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    const size_t line_bytes = (size_t)cols * 3 + 1, raw = line_bytes * (size_t)(y1 - y0);
    const size_t cap = deflateBound(&zs, (uLong)raw) + 64;
    if (cap > 0xffffffffu) { deflateEnd(&zs); return false; }
    out.data.reset(new (std::nothrow) uint8_t[cap]);                 // (not value-initialised: 30 MB a band)
    std::vector<uint8_t> lines;
    try { lines.resize(raw); } catch (const std::bad_alloc&) { out.data.reset(); }
    if (!out.data) { deflateEnd(&zs); return false; }
    // the band's filtered rows (filter type 0, BGR -> RGB), then ONE deflate call
    for (int y = y0; y < y1; y++) {
        uint8_t* line = lines.data() + (size_t)(y - y0) * line_bytes;
        line[0] = 0;
        const uint8_t* s = bgr + (size_t)y * cols * 3;
        for (int x = 0; x < cols; x++) { line[1 + 3 * x] = s[3 * x + 2]; line[2 + 3 * x] = s[3 * x + 1]; line[3 + 3 * x] = s[3 * x]; }
    }
    out.adler = adler32(adler32(0L, Z_NULL, 0), lines.data(), (uInt)raw);
    zs.next_in = lines.data(); zs.avail_in = (uInt)raw;
    zs.next_out = out.data.get(); zs.avail_out = (uInt)cap;
    const int r = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    const bool ok = r != Z_STREAM_ERROR && zs.avail_in == 0 && zs.avail_out > 0 && (!last || r == Z_STREAM_END);
    out.size = ok ? (size_t)zs.total_out : 0;
    deflateEnd(&zs);
    return ok;
}

static bool write_png(FILE* f, const uint8_t* bgr, int rows, int cols)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (std::fwrite(sig, 1, 8, f) != 8) return false;
    uint8_t ihdr[13];
    put_be32(ihdr, (uint32_t)cols); put_be32(ihdr + 4, (uint32_t)rows);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    if (!png_chunk(f, "IHDR", ihdr, 13)) return false;
    const int nbands = (rows + kBandRows - 1) / kBandRows;
    std::vector<Band> band(nbands);
    std::mutex mu; std::condition_variable cv;
    std::atomic<int> next{ 0 };
    auto work = [&] {
        for (int b = next.fetch_add(1); b < nbands; b = next.fetch_add(1)) {
            const int y0 = b * kBandRows, y1 = std::min(rows, y0 + kBandRows);
            const bool ok = deflate_band(bgr, cols, y0, y1, b == nbands - 1, band[b]);
            { std::lock_guard<std::mutex> l(mu); band[b].state = ok ? 1 : -1; }
            cv.notify_all();
        }
    };
    // the bands are compressed by up to eight threads while this one writes them out, in order, as they become ready
    const unsigned hc = std::thread::hardware_concurrency();
    const int nthreads = std::max(1, std::min({ nbands, 8, (int)hc }));
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work);
    // the stream: zlib header (deflate, 32 K window, fastest), the bands, the Adler-32 of everything -- cut into IDAT chunks of at most 1 MB
    std::vector<uint8_t> buf; buf.reserve((1 << 20) + 16);
    auto flush_chunk = [&]() { const bool ok = buf.empty() || png_chunk(f, "IDAT", buf.data(), (uint32_t)buf.size()); buf.clear(); return ok; };
    auto put = [&](const uint8_t* p, size_t n) {
        while (n) {
            const size_t room = ((size_t)1 << 20) - buf.size(), k = std::min(room, n);
            buf.insert(buf.end(), p, p + k); p += k; n -= k;
            if (buf.size() == ((size_t)1 << 20) && !flush_chunk()) return false;
        }
        return true;
    };
    const uint8_t zhdr[2] = { 0x78, 0x01 };
    bool ok = put(zhdr, 2);
    uLong ad = 1;
    for (int b = 0; b < nbands; b++) {
        { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return band[b].state != 0; }); }
        ok = ok && band[b].state == 1 && put(band[b].data.get(), band[b].size);
        const int y0 = b * kBandRows, y1 = std::min(rows, y0 + kBandRows);
        ad = b == 0 ? band[0].adler : adler32_combine(ad, band[b].adler, (z_off_t)((size_t)(y1 - y0) * ((size_t)cols * 3 + 1)));
        band[b].data.reset();
    }
    for (auto& t : th) t.join();
    uint8_t tail[4]; put_be32(tail, (uint32_t)ad);
    ok = ok && put(tail, 4) && flush_chunk();
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
