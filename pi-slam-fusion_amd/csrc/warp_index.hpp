// warp_index.hpp -- source addressing of the multi-band warp (cv::remap INTER_LINEAR + BORDER_REFLECT):
// which 8 bytes of which two source rows a warped pixel loads, and how its four bilinear taps are cut out of them.
// Plain integer code shared by the kernel (kernels.hip, warp_fetch / warp_finish) and by the host-side exhaustive
// check tests/cpp/warp_index_check.cpp, which proves for every source coordinate around a frame that
//   * no byte outside [0, total) is ever loaded (total = the bytes the caller handed over), and
//   * the four taps are the pixels OpenCV's borderInterpolate(BORDER_REFLECT) names.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PF_HD __host__ __device__ __forceinline__
#else
#define PF_HD inline
#endif

namespace pf {

PF_HD int wi_min(int a, int b) { return a < b ? a : b; }
#if defined(__HIP_DEVICE_COMPILE__)
PF_HD int wi_mul24(int a, int b) { return __mul24(a, b); }      // rows, steps and pixel offsets fit 24 bits (frames < 2 GiB, host-checked)
#else
PF_HD int wi_mul24(int a, int b) { return a * b; }
#endif

PF_HD int border_reflect(int p, int len)          // BORDER_REFLECT  fedcba|abcdefgh|hgfedcb
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    const int m = 2 * len;
    int q = p % m;
    if (q < 0) q += m;
    return q < len ? q : m - 1 - q;
}
PF_HD int border_reflect101(int p, int len)       // BORDER_REFLECT_101 gfedcb|abcdefgh|gfedcba
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    const int m = 2 * len - 2;
    int q = p % m;
    if (q < 0) q += m;
    return q < len ? q : m - q;
}
// the same two maps for -len <= p < 2*len (one reflection, no division): everything a canvas up to three
// frames wide asks for.  p ^ (p >> 31) == (p < 0 ? -p-1 : p).
PF_HD bool reflect_is_near(int p, int len) { return (unsigned)(p + len) < 3u * (unsigned)len; }
PF_HD int border_reflect_near(int p, int len)
{
    const int q = p ^ (p >> 31);
    return wi_min(q, 2 * len - 1 - q);
}
PF_HD int border_reflect101_near(int p, int len)
{
    const int q = p < 0 ? -p : p;
    return len == 1 ? 0 : wi_min(q, 2 * len - 2 - q);
}
PF_HD int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// flags of a fetched pixel: tap = the "hi" pixel of the 8 loaded bytes, byte shifts of the two rows, weight in bounds
constexpr uint32_t kT0Hi = 1, kT1Hi = 2, kBack0 = 2, kBack1 = 5, kInb = 1u << 8;

struct TapAddr { uint32_t off0, off1, flags; };

// bytes of a frame that may be read: rows-1 full steps plus one row of pixels (the last row's padding is not the caller's)
PF_HD long frame_bytes(int srows, int scols, long sstep, int cn) { return (long)(srows - 1) * sstep + (long)scols * cn; }

// (ux, uy) strictly inside the frame and not on its last two rows: both taps of a row are the (lo, hi) pixels of
// one 8-byte load at the pixel, the second row is one step further, and both loads end inside the frame
// (a frame of one row has no such pixel: srows - 2 must not wrap)
PF_HD bool tap_is_fast(int ux, int uy, int srows, int scols) { return (unsigned)ux < (unsigned)(scols - 1) && (unsigned)uy < (unsigned)(srows < 2 ? 0 : srows - 2); }
PF_HD TapAddr tap_addr_fast(int ux, int uy, int sstep, int cn)
{
    const uint32_t off0 = (uint32_t)(wi_mul24(uy, sstep) + wi_mul24(cn, ux));
    return { off0, off0 + (uint32_t)sstep, kT1Hi };
}

// every other coordinate: the four taps after BORDER_REFLECT.  `near`: all four raw coordinates pass reflect_is_near
// (single reflection), else the general map.  After reflection the two taps of a row are the same or adjacent pixels:
// one 8-byte load at the lower one serves both; at the end of the frame the load is moved back and the bytes shifted.
PF_HD TapAddr tap_addr_border(int ux, int uy, bool near, int srows, int scols, int sstep, int cn, uint32_t total)
{
    const int sx = sat_short(ux), sy = sat_short(uy);
    int sx0, sx1, sy0, sy1;
    if (near) {
        sx0 = border_reflect_near(sx, scols); sx1 = border_reflect_near(sx + 1, scols);
        sy0 = border_reflect_near(sy, srows); sy1 = border_reflect_near(sy + 1, srows);
    } else {
        sx0 = border_reflect(sx, scols); sx1 = border_reflect(sx + 1, scols);
        sy0 = border_reflect(sy, srows); sy1 = border_reflect(sy + 1, srows);
    }
    const int xbase = sx0 < sx1 ? sx0 : sx1;
    uint32_t flags = 0;
    if (sx0 != xbase) flags |= kT0Hi;
    if (sx1 != xbase) flags |= kT1Hi;
    uint32_t off0 = (uint32_t)(wi_mul24(sy0, sstep) + wi_mul24(cn, xbase));
    uint32_t off1 = (uint32_t)(wi_mul24(sy1, sstep) + wi_mul24(cn, xbase));
    // last bytes of the frame: never read past it -- read earlier and shift
    const uint32_t back0 = off0 + 8 > total ? off0 + 8 - total : 0u, back1 = off1 + 8 > total ? off1 + 8 - total : 0u;
    off0 -= back0; off1 -= back1;
    flags |= back0 << kBack0 | back1 << kBack1;
    return { off0, off1, flags };
}
PF_HD bool tap_is_near(int ux, int uy, int srows, int scols)
{
    const int sx = sat_short(ux), sy = sat_short(uy);
    return reflect_is_near(sx, scols) && reflect_is_near(sx + 1, scols) && reflect_is_near(sy, srows) && reflect_is_near(sy + 1, srows);
}

// the two pixels (3 bytes each, little end first) of one row's 8 loaded bytes: "lo" at byte 0, "hi" at byte cn
PF_HD void row_taps(uint32_t lo_word, uint32_t hi_word, uint32_t back, int cn, uint32_t& lo, uint32_t& hi)
{
    uint64_t bits = (uint64_t)hi_word << 32 | lo_word;
    bits >>= 8 * back;
    lo = (uint32_t)bits & 0xffffffu;
    hi = (uint32_t)(bits >> (8 * cn)) & 0xffffffu;
}

}  // namespace pf
