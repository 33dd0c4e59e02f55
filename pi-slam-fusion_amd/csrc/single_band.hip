// single_band.hip -- Map2DCPU semantics on the GPU (reference
// Map2DFusion/Map2DCPU.cpp:150-334): one 8-bit BGRA tile per mosaic cell, alpha = weight byte
// (dis*254, floor 2), cv::warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) on 8UC4 through
// OpenCV's 15-bit fixed-point bilinear taps, select `if (ele.a < dst.a) ele = dst`.
#include "kernels.hpp"

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

// one thread = one canvas pixel; a wave = one 64-pixel OpenCV coordinate block row
__global__ __launch_bounds__(256) void k_single(const uint8_t* __restrict__ src, const uint8_t* __restrict__ w8, WarpArgs a,
                                                 const uint64_t* __restrict__ table, int tiles_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xb = a.x_off + blockIdx.x * 64;
    const int y  = a.y_off + blockIdx.y * 4 + wave;
    const int x  = xb + lane;
    const uint64_t ent = table[(y >> 8) * tiles_x + (x >> 8)];
    if (!ent) return;
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
        out = (uint32_t)sat_u8(accB >> 15) | ((uint32_t)sat_u8(accG >> 15) << 8) |
              ((uint32_t)sat_u8(accR >> 15) << 16) | ((uint32_t)sat_u8(accA >> 15) << 24);
    }
    uint32_t PF_GLOBAL* tile = (uint32_t PF_GLOBAL*)(ent & ~(uint64_t)1) + ((y & 255) * kElePixels + (x & 255));
    if (ent & 1) { *tile = (out >> 24) ? out : 0u; return; }   // fresh tile: zeros(...) then the select against alpha 0
    const uint32_t cur = *tile;
    if ((cur >> 24) < (out >> 24)) *tile = out;                // Map2DCPU.cpp:326-327
}

void launch_single(hipStream_t s, const uint8_t* src, const uint8_t* w8, const WarpArgs& a, const uint64_t* table, int tiles_x)
{
    dim3 grid(a.wcols / 64, a.wrows / 4), block(256);
    hipLaunchKernelGGL(k_single, grid, block, 0, s, src, w8, a, table, tiles_x);
}

}  // namespace pf
