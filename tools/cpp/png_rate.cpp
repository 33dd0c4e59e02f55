// dev tool: rate of the PNG writer (csrc/image_io.cpp: 256-row bands deflated on up to eight threads while the caller writes them out) on a
// compressible 12 800 x 15 104 picture -- the bench mosaic's size; the bench's own keyframes are noise, which deflate stores at memory speed.
//   g++ -O2 -std=c++17 -Ipi-slam-fusion_amd/csrc tools/cpp/png_rate.cpp pi-slam-fusion_amd/csrc/{image_io,jpeg_decode,png_decode}.cpp -o /tmp/png_rate -lz -lpthread
//   /tmp/png_rate [rows cols] ; taskset -c 0 /tmp/png_rate          (one core: what a single deflate thread does)
#include "../../include/pifusion.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
namespace pf { static thread_local std::string g; void set_error(const std::string& m) { g = m; } const char* last_error() { return g.c_str(); } }
int main(int argc, char** argv)
{
    const int h = argc > 2 ? std::atoi(argv[1]) : 15104, w = argc > 2 ? std::atoi(argv[2]) : 12800;
    std::vector<uint8_t> px((size_t)h * w * 3);
    uint32_t s = 12345;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t* p = &px[((size_t)y * w + x) * 3];
            s = s * 1664525u + 1013904223u;                                   // a smooth picture with two bits of noise: what an orthomosaic deflates like
            p[0] = (uint8_t)(x / 3 + y / 5 + (s >> 30)); p[1] = (uint8_t)(x / 7 * 3 + y / 2 + ((s >> 28) & 3)); p[2] = (uint8_t)((((long)x * y) >> 12) + ((s >> 26) & 3));
        }
    const auto t0 = std::chrono::steady_clock::now();
    const int ok = pf_write_image("/tmp/png_rate.png", px.data(), h, w);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    FILE* f = std::fopen("/tmp/png_rate.png", "rb"); std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fclose(f);
    std::printf("ok %d: %d x %d, %.0f MB of pixels -> %.0f MB file in %.2f s = %.0f MB/s\n", ok, w, h, px.size() / 1e6, sz / 1e6, dt, px.size() / 1e6 / dt);
    std::remove("/tmp/png_rate.png");
    return ok ? 0 : 1;
}
