// single_band.hip -- Map2DCPU semantics on the GPU (reference
// Map2DFusion/Map2DCPU.cpp:150-334): one 8-bit BGRA tile per mosaic cell, alpha = weight byte
// (dis*254, floor 2), cv::warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) on 8UC4 through
// OpenCV's 15-bit fixed-point bilinear taps, select `if (ele.a < dst.a) ele = dst`.
#include "kernels.hpp"
#include "env.hpp"
#include <cmath>
#include <cstdlib>

namespace pf {

#define PF_GLOBAL __attribute__((address_space(1)))

// alpha plane of the reference's 8UC4 weightImage (Map2DCPU.cpp:245-263)
__global__ __launch_bounds__(256) void k_weight8(uint8_t* __restrict__ w, int rows, int cols, float xc, float yc,
                                                  float dis_max, int weight_type)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= cols) return;
    float dis = ((float)i - yc) * ((float)i - yc) + ((float)j - xc) * ((float)j - xc);
    dis = 1.f - sqrtf(dis) / dis_max;
    uint8_t a;
    if (weight_type == 0) a = (uint8_t)(int)((double)dis * 254.);
    else a = (uint8_t)(int)(dis * dis * 254.f);
    if (a < 2) a = 2;
    w[(long)i * cols + j] = a;
}

void launch_weight8(hipStream_t s, uint8_t* w, int rows, int cols, int weight_type)
{
    const float xc = (float)(cols / 2), yc = (float)(rows / 2);
    const float dis_max = sqrtf(xc * xc + yc * yc);
    dim3 grid((cols + 255) / 256, rows), block(256);
    hipLaunchKernelGGL(k_weight8, grid, block, 0, s, w, rows, cols, xc, yc, dis_max, weight_type);
}

// MultiBandMap2DCPU's weightImage (MultiBandMap2DCPU.cpp:400-418): same float expression as the
// in-kernel analytic weight, evaluated once per frame size instead of once per warped pixel
__global__ __launch_bounds__(256) void k_weight32(float* __restrict__ w, int rows, int cols, float xc, float yc,
                                                   float dis_max, int weight_type)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= cols) return;
    const float dy = (float)i - yc, dx = (float)j - xc;
    float dis = dy * dy + dx * dx;
    dis = 1.f - sqrtf(dis) / dis_max;
    float wv = weight_type == 0 ? dis : dis * dis;
    if (wv <= 1e-5f) wv = 1e-5f;
    w[(long)i * cols + j] = wv;
}

void launch_weight32(hipStream_t s, float* w, int rows, int cols, int weight_type)
{
    const float xc = (float)(cols / 2), yc = (float)(rows / 2);
    const float dis_max = sqrtf(xc * xc + yc * yc);
    dim3 grid((cols + 255) / 256, rows), block(256);
    hipLaunchKernelGGL(k_weight32, grid, block, 0, s, w, rows, cols, xc, yc, dis_max, weight_type);
}

__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// (acc >> 15) saturated to 8 bits, four channels packed B | G<<8 | R<<16 | A<<24.  Each component is laundered through
// an asm operand before the OR: left to itself the compiler turns the first two into `v_ashr_pk_u8_i32`, treats that
// instruction's upper 16 result bits as zero, and on gfx950 they are not -- whatever the destination register held
// before shows up in the red channel.
__device__ __forceinline__ uint32_t pack_bgra(int accB, int accG, int accR, int accA)
{
    int b = sat_u8(accB >> 15), g = sat_u8(accG >> 15), r = sat_u8(accR >> 15), al = sat_u8(accA >> 15);
    asm volatile("" : "+v"(b), "+v"(g), "+v"(r), "+v"(al));
    return (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16) | ((uint32_t)al << 24);
}

// one thread = one canvas pixel; a wave = one 64-pixel OpenCV coordinate block row
__global__ __launch_bounds__(256) void k_single(const uint8_t* __restrict__ src, const uint8_t* __restrict__ w8, WarpArgs a,
                                                 const uint64_t* __restrict__ table, int tiles_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xb = a.x_off + blockIdx.x * 64;
    const int y  = a.y_off + blockIdx.y * 4 + wave;
    const int x  = xb + lane;
    uint64_t ent = table[(y >> 8) * tiles_x + (x >> 8)];
    if (!ent) return;
    if ((ent >> (48 + ((y >> 6) & 3) * 4 + ((xb >> 6) & 3))) & 1) return;      // the cull: see k_single2
    ent &= 0x0000ffffffffffffull;
    const double X0 = a.M[0] * xb + a.M[1] * y + a.M[2];
    const double Y0 = a.M[3] * xb + a.M[4] * y + a.M[5];
    const double W0 = a.M[6] * xb + a.M[7] * y + a.M[8];
    const double W  = W0 + a.M[6] * lane;
    const double Wl = (W ? 1. / W : 0) * 32.;                 // == 32./W bit for bit
    const int X = __double2int_rn((X0 + a.M[0] * lane) * Wl);  // cvt saturates == clamp + cvRound
    const int Y = __double2int_rn((Y0 + a.M[3] * lane) * Wl);
    auto ss = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
    const int sx = ss(X >> 5), sy = ss(Y >> 5);
    uint32_t out = 0;
    if (!(sx >= a.scols || sx + 1 < 0 || sy >= a.srows || sy + 1 < 0)) {
        // integer taps: saturate_cast<short>(w*32768); the products are exact on the 1/32 grid and sum to
        // 32768 except at (0,0) (32768 -> 32767, the missing 1 goes to another tap: no effect on outputs)
        const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
        int w[4] = { __float2int_rn((1.f - fy) * (1.f - fx) * 32768.f), __float2int_rn((1.f - fy) * fx * 32768.f),
                     __float2int_rn(fy * (1.f - fx) * 32768.f), __float2int_rn(fy * fx * 32768.f) };
        if (w[0] > 32767) w[0] = 32767;
        w[3] += 32768 - (w[0] + w[1] + w[2] + w[3]);
        const bool inx0 = sx >= 0, inx1 = sx + 1 < a.scols, iny0 = sy >= 0, iny1 = sy + 1 < a.srows;
        int accB = 1 << 14, accG = 1 << 14, accR = 1 << 14, accA = 1 << 14;
        auto tap = [&](bool in, int tx, int ty, int wt) {
            if (!in) return;                                   // BORDER_CONSTANT: the tap contributes cval = 0
            const int off = __mul24(ty, (int)a.sstep) + tx * a.src_cn;
            const int b = src[off], g = src[off + 1], r = src[off + 2], al = w8[__mul24(ty, a.scols) + tx];
            accB += __mul24(b, wt); accG += __mul24(g, wt); accR += __mul24(r, wt); accA += __mul24(al, wt);
        };
        tap(inx0 && iny0, sx, sy, w[0]);
        tap(inx1 && iny0, sx + 1, sy, w[1]);
        tap(inx0 && iny1, sx, sy + 1, w[2]);
        tap(inx1 && iny1, sx + 1, sy + 1, w[3]);
        out = pack_bgra(accB, accG, accR, accA);
    }
    uint32_t PF_GLOBAL* tile = (uint32_t PF_GLOBAL*)(ent & ~(uint64_t)1) + ((y & 255) * kElePixels + (x & 255));
    if (ent & 1) { *tile = (out >> 24) ? out : 0u; return; }   // fresh tile: zeros(...) then the select against alpha 0
    const uint32_t cur = *tile;
    if ((cur >> 24) < (out >> 24)) *tile = out;                // Map2DCPU.cpp:326-327
}

// ---------------------------------------------------------------------------------------------------------
// k_single2: the same result with the techniques of the multi-band warp (kernels.hip, warp_fetch):
//   * a thread keeps one canvas column and walks down 8 rows, two per step: the column terms of the fp64
//     coordinate chain are formed once, the second pixel's coordinates and loads overlap the first one's loads;
//   * correctly-rounded reciprocal without the scale/fixup steps and round-half-even by the 1.5*2^52 add when the
//     whole wave is in the tame range (host-checked homography, `plain`), the general forms otherwise;
//   * when the whole wave is strictly inside the frame: one unaligned 8-byte load per source row for both taps
//     (BGR or BGRA) and one 2-byte load per row of the weight-byte plane, instead of 16 byte loads;
//   * one tile-table entry per thread (8 rows never straddle a tile).
struct SingleTaps { uint32_t lo0, hi0, lo1, hi1, wa; int X, Y; bool fast; };

__device__ __forceinline__ double rcp_mid(double d)              // see kernels.hip rcp_mid_range
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(rem, r, r);
}

struct SingleCol { double m0xb, m3xb, m6xb, m0x1, m3x1, m6x1; };

__device__ __forceinline__ void single_coords(const WarpArgs& a, const SingleCol& col, int y, int plain, int& X, int& Y)
{
    const double X0 = col.m0xb + a.M[1] * y + a.M[2];
    const double Y0 = col.m3xb + a.M[4] * y + a.M[5];
    const double W0 = col.m6xb + a.M[7] * y + a.M[8];
    const double W  = W0 + col.m6x1;
    const double xn = X0 + col.m0x1, yn = Y0 + col.m3x1;
    const double Wn = rcp_mid(W);
    const double px = xn * Wn * 32., py = yn * Wn * 32.;        // (1/W)*32 == 32/W bit for bit; scaling by 32 commutes with rounding
    const bool tame = fabs(px) < 1.0e9 && fabs(py) < 1.0e9;
    if (plain && __builtin_amdgcn_ballot_w64(!tame) == 0) {
        constexpr double kMagic = 6755399441055744.0;
        X = (int)(uint32_t)(unsigned long long)__double_as_longlong(px + kMagic);
        Y = (int)(uint32_t)(unsigned long long)__double_as_longlong(py + kMagic);
    } else {
        const double Wl = (W ? 1. / W : 0) * 32.;
        X = __double2int_rn(xn * Wl);                            // cvt saturates == clamp + cvRound
        Y = __double2int_rn(yn * Wl);
    }
}

//   * (round 6) whole block ROWS of the canvas go to one XCD each, row r to XCD r mod 8: the hardware hands consecutive workgroup ids to the
//     8 XCDs round robin, so with a plain 2-D grid the horizontal neighbours of a block -- which read the same 128-byte frame lines and weight
//     bytes -- ran on seven other XCDs with L2s of their own (FETCH 171 MB per keyframe for 106 MB of reads, profiles/r05_single_band_traffic.txt).
//     (Contiguous BANDS of rows per XCD fetch as little, 104 MB, but leave the XCDs that hold the canvas's top and bottom -- mostly outside
//     the frame's footprint -- idle early: 65 us instead of 62, profiles/r06_single_band.md.)
__global__ __launch_bounds__(256) void k_single2(const uint8_t* __restrict__ src, const uint8_t* __restrict__ w8, WarpArgs a,
                                                  const uint64_t* __restrict__ table, int tiles_x, int plain, int nbx, int nblk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroup id b runs on XCD b & 7 and is the (b >> 3)-th workgroup there: block rows xcd, xcd + 8, xcd + 16, ... left to right
    const int i = (int)(blockIdx.x >> 3), ri = i / nbx, bx_ = i - ri * nbx, by_ = ri * 8 + (int)(blockIdx.x & 7);
    if (by_ * nbx >= nblk) return;
    const int xb = a.x_off + bx_ * 64;
    const int x  = xb + lane;
    const int yb = a.y_off + by_ * 32 + wave * 8;
    uint64_t ent = table[(yb >> 8) * tiles_x + (x >> 8)];
    if (!ent) return;
    // the cull (fusion_map.cpp build_tile_table): bits 48..63 of an entry flag the 64 x 64 cells of the tile in which this keyframe cannot
    // raise a stored alpha; a block (64 x 32) lies inside one cell
    if ((ent >> (48 + ((yb >> 6) & 3) * 4 + ((xb >> 6) & 3))) & 1) return;
    ent &= 0x0000ffffffffffffull;
    const SingleCol col = { a.M[0] * xb, a.M[3] * xb, a.M[6] * xb, a.M[0] * lane, a.M[3] * lane, a.M[6] * lane };
    const int cn = a.src_cn, step = (int)a.sstep;
    const uint32_t hisel = cn == 3 ? 0x06050403u : 0x07060504u;
    auto ss = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
    typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(1)));
    typedef unsigned short us1 __attribute__((aligned(1)));

    auto fetch = [&](int y) {
        SingleTaps t;
        single_coords(a, col, y, plain, t.X, t.Y);
        const int ux = t.X >> 5, uy = t.Y >> 5;
        const bool in = (unsigned)ux < (unsigned)(a.scols - 1) && (unsigned)uy < (unsigned)(a.srows < 2 ? 0 : a.srows - 2);
        t.fast = plain && __builtin_amdgcn_ballot_w64(!in) == 0;
        t.lo0 = t.hi0 = t.lo1 = t.hi1 = t.wa = 0;
        if (t.fast) {
            const uint32_t off = (uint32_t)(__mul24(uy, step) + __mul24(cn, ux)), woff = (uint32_t)(__mul24(uy, a.scols) + ux);
            const u2 b0 = *(const u2*)(src + off), b1 = *(const u2*)(src + off + (uint32_t)step);
            const uint32_t w0 = *(const us1*)(w8 + woff), w1 = *(const us1*)(w8 + woff + (uint32_t)a.scols);
            t.lo0 = b0.x; t.hi0 = b0.y; t.lo1 = b1.x; t.hi1 = b1.y; t.wa = w0 | (w1 << 16);
        }
        return t;
    };
    auto finish = [&](const SingleTaps& t, int y) {
        const int X = t.X, Y = t.Y;
        // integer taps: saturate_cast<short>(w*32768); the products are exact on the 1/32 grid and sum to
        // 32768 except at (0,0) (32768 -> 32767, the missing 1 goes to another tap: no effect on outputs)
        const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
        int w[4] = { __float2int_rn((1.f - fy) * (1.f - fx) * 32768.f), __float2int_rn((1.f - fy) * fx * 32768.f),
                     __float2int_rn(fy * (1.f - fx) * 32768.f), __float2int_rn(fy * fx * 32768.f) };
        if (w[0] > 32767) w[0] = 32767;
        w[3] += 32768 - (w[0] + w[1] + w[2] + w[3]);
        uint32_t out = 0;
        if (t.fast) {
            const uint32_t p00 = t.lo0, p01 = __builtin_amdgcn_perm(t.hi0, t.lo0, hisel);
            const uint32_t p10 = t.lo1, p11 = __builtin_amdgcn_perm(t.hi1, t.lo1, hisel);
            auto byte = [](uint32_t v, int k) { return (int)((v >> (8 * k)) & 0xffu); };
            const int accB = (1 << 14) + __mul24(byte(p00, 0), w[0]) + __mul24(byte(p01, 0), w[1]) + __mul24(byte(p10, 0), w[2]) + __mul24(byte(p11, 0), w[3]);
            const int accG = (1 << 14) + __mul24(byte(p00, 1), w[0]) + __mul24(byte(p01, 1), w[1]) + __mul24(byte(p10, 1), w[2]) + __mul24(byte(p11, 1), w[3]);
            const int accR = (1 << 14) + __mul24(byte(p00, 2), w[0]) + __mul24(byte(p01, 2), w[1]) + __mul24(byte(p10, 2), w[2]) + __mul24(byte(p11, 2), w[3]);
            const int accA = (1 << 14) + __mul24(byte(t.wa, 0), w[0]) + __mul24(byte(t.wa, 1), w[1]) + __mul24(byte(t.wa, 2), w[2]) + __mul24(byte(t.wa, 3), w[3]);
            out = pack_bgra(accB, accG, accR, accA);
        } else {
            const int sx = ss(X >> 5), sy = ss(Y >> 5);
            if (!(sx >= a.scols || sx + 1 < 0 || sy >= a.srows || sy + 1 < 0)) {
                const bool inx0 = sx >= 0, inx1 = sx + 1 < a.scols, iny0 = sy >= 0, iny1 = sy + 1 < a.srows;
                int accB = 1 << 14, accG = 1 << 14, accR = 1 << 14, accA = 1 << 14;
                auto tap = [&](bool in, int tx, int ty, int wt) {
                    if (!in) return;                                   // BORDER_CONSTANT: the tap contributes cval = 0
                    const int off = __mul24(ty, step) + tx * cn;
                    const int b = src[off], g = src[off + 1], r = src[off + 2], al = w8[__mul24(ty, a.scols) + tx];
                    accB += __mul24(b, wt); accG += __mul24(g, wt); accR += __mul24(r, wt); accA += __mul24(al, wt);
                };
                tap(inx0 && iny0, sx, sy, w[0]);
                tap(inx1 && iny0, sx + 1, sy, w[1]);
                tap(inx0 && iny1, sx, sy + 1, w[2]);
                tap(inx1 && iny1, sx + 1, sy + 1, w[3]);
                out = pack_bgra(accB, accG, accR, accA);
            }
        }
        uint32_t PF_GLOBAL* tile = (uint32_t PF_GLOBAL*)(ent & ~(uint64_t)1) + ((y & 255) * kElePixels + (x & 255));
        if (ent & 1) { *tile = (out >> 24) ? out : 0u; return; }   // fresh tile: zeros(...) then the select against alpha 0
        const uint32_t cur = *tile;
        if ((cur >> 24) < (out >> 24)) *tile = out;                // Map2DCPU.cpp:326-327
    };
#pragma unroll 1
    for (int k = 0; k < 8; k += 2) {
        const SingleTaps ta = fetch(yb + k), tb = fetch(yb + k + 1);
        finish(ta, yb + k); finish(tb, yb + k + 1);
    }
}

void launch_single(hipStream_t s, const uint8_t* src, const uint8_t* w8, const WarpArgs& a, const uint64_t* table, int tiles_x)
{
    static const bool old_kernel = exp_env("PF_SINGLE_OLD") != nullptr;     // experiments library: the one-pixel-per-thread form
    if (old_kernel) {
        dim3 grid(a.wcols / 64, a.wrows / 4), block(256);
        hipLaunchKernelGGL(k_single, grid, block, 0, s, src, w8, a, table, tiles_x);
        return;
    }
    // plain: see FusedWarp::plain (kernels.hip)
    int plain = a.srows <= 32767 && a.scols <= 32767 && !exp_env("PF_FORCE_GENERAL");
    for (int i = 0; i < 9; i++) if (!(std::fabs(a.M[i]) < 0x1p400)) plain = 0;
    const int nbx = a.wcols / 64, nblk = nbx * (a.wrows / 32);
    if (nblk <= 0) return;
    const int nby = a.wrows / 32;
    dim3 grid((unsigned)(((nby + 7) >> 3) << 3) * (unsigned)nbx), block(256);
    hipLaunchKernelGGL(k_single2, grid, block, 0, s, src, w8, a, table, tiles_x, plain, nbx, nblk);
}

}  // namespace pf
