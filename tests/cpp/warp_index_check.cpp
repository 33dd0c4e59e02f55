// Exhaustive host-side check of the warp's source addressing (pi-slam-fusion_amd/csrc/warp_index.hpp, the code the
// kernel runs): for every 1/32-px-truncated source coordinate in a wide ring around frames of several shapes
// (packed, row-padded, BGR, BGRA, degenerate 1-pixel-wide / 1-row frames) and for every path the kernel can take
// for it (fast / single reflection / general reflection):
//   * the two 8-byte loads lie inside [0, frame_bytes) -- the bytes the caller handed over, nothing else;
//   * the four taps cut out of them are the pixels cv::borderInterpolate(BORDER_REFLECT) names
//     (OpenCV 2.4.9 remap), for the saturate_cast<short>'ed coordinate.
// Exit code 0 = all good; prints the first violation otherwise.
#include "../../pi-slam-fusion_amd/csrc/warp_index.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

static int ref_reflect(int p, int len)        // borderInterpolate, BORDER_REFLECT, written as OpenCV does (loop)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { if (p < 0) p = -p - 1; else p = len - 1 - (p - len); } while ((unsigned)p >= (unsigned)len);
    return p;
}

static long checked = 0;

static int check_frame(int srows, int scols, int cn, int step)
{
    const long total = pf::frame_bytes(srows, scols, step, cn);
    if (total < 8) return 0;                                   // refused at the boundary (feed: frame smaller than one load)
    std::vector<uint8_t> buf((size_t)total);
    for (long i = 0; i < total; i++) buf[(size_t)i] = 0xEE;   // row padding
    auto px = [&](int y, int x, int k) { return (uint8_t)(1 + ((y * 131 + x * 17 + k * 5) % 199)); };
    for (int y = 0; y < srows; y++) for (int x = 0; x < scols; x++) for (int k = 0; k < cn; k++) buf[(size_t)y * step + x * cn + k] = px(y, x, k);
    auto load8 = [&](uint32_t off, uint32_t& lo, uint32_t& hi) {
        if ((long)off + 8 > total) return false;
        lo = hi = 0;
        for (int i = 0; i < 4; i++) { lo |= (uint32_t)buf[off + i] << (8 * i); hi |= (uint32_t)buf[off + 4 + i] << (8 * i); }
        return true;
    };
    const int rx = 3 * scols + 40, ry = 3 * srows + 40;
    const int extra[] = { -40000, -32769, -32768, 32766, 32767, 32768, 40000 };       // around saturate_cast<short>
    std::vector<int> xs, ys;
    for (int v = -rx; v <= scols + rx; v++) xs.push_back(v);
    for (int v = -ry; v <= srows + ry; v++) ys.push_back(v);
    for (int e : extra) { xs.push_back(e); ys.push_back(e); }
    for (int uy : ys) for (int ux : xs) {
        const int sx = pf::sat_short(ux), sy = pf::sat_short(uy);
        const int ex[2] = { ref_reflect(sx, scols), ref_reflect(sx + 1, scols) }, ey[2] = { ref_reflect(sy, srows), ref_reflect(sy + 1, srows) };
        for (int path = 0; path < 3; path++) {
            pf::TapAddr ta;
            if (path == 0) { if (!pf::tap_is_fast(ux, uy, srows, scols)) continue; ta = pf::tap_addr_fast(ux, uy, step, cn); }
            else if (path == 1) { if (!pf::tap_is_near(ux, uy, srows, scols)) continue; ta = pf::tap_addr_border(ux, uy, true, srows, scols, step, cn, (uint32_t)total); }
            else ta = pf::tap_addr_border(ux, uy, false, srows, scols, step, cn, (uint32_t)total);
            uint32_t lo[2], hi[2];
            const uint32_t off[2] = { ta.off0, ta.off1 };
            for (int j = 0; j < 2; j++) {
                uint32_t lw, hw;
                if (!load8(off[j], lw, hw)) {
                    std::printf("OUT OF BOUNDS: frame %dx%d cn %d step %d (total %ld), ux %d uy %d path %d row %d: offset %u\n", srows, scols, cn, step, total, ux, uy, path, j, off[j]);
                    return 1;
                }
                pf::row_taps(lw, hw, (ta.flags >> (j ? pf::kBack1 : pf::kBack0)) & 7, cn, lo[j], hi[j]);
            }
            for (int j = 0; j < 2; j++) for (int t = 0; t < 2; t++) {
                const bool use_hi = ta.flags & (t ? pf::kT1Hi : pf::kT0Hi);
                const uint32_t v = use_hi ? hi[j] : lo[j];
                for (int k = 0; k < 3; k++)
                    if (((v >> (8 * k)) & 0xff) != px(ey[j], ex[t], k)) {
                        std::printf("WRONG TAP: frame %dx%d cn %d step %d, ux %d uy %d path %d: tap (row %d, col %d) channel %d = %u, expected pixel (%d,%d) = %u\n",
                                    srows, scols, cn, step, ux, uy, path, j, t, k, (v >> (8 * k)) & 0xff, ey[j], ex[t], px(ey[j], ex[t], k));
                        return 1;
                    }
            }
            checked++;
        }
    }
    return 0;
}

int main()
{
    const int shapes[][2] = { { 5, 7 }, { 3, 3 }, { 2, 2 }, { 1, 3 }, { 3, 1 }, { 1, 8 }, { 8, 1 }, { 4, 2 }, { 16, 11 }, { 48, 64 } };
    for (auto& sh : shapes)
        for (int cn = 3; cn <= 4; cn++)
            for (int pad : { 0, 1, 2, 3, 4, 5, 7, 8, 60 })
                if (check_frame(sh[0], sh[1], cn, sh[1] * cn + pad)) return 1;
    std::printf("warp addressing: %ld (coordinate, path) cases in bounds with the right taps\n", checked);
    return 0;
}
