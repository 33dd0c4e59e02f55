// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the multi-band
// fusion hot path.  Compiled with -ffp-contract=off: every fp32/fp64 operation
// below is evaluated in exactly the written order so that results are
// bit-identical to the reference's OpenCV 2.4.9 arithmetic.
//
// Path (reference: Map2DFusion/MultiBandMap2DCPU.cpp):
//   k_warp        :443-452  convertTo + warpPerspective(LINEAR,REFLECT) of the
//                           frame + warpPerspective(NEAREST,CONSTANT) of the
//                           radial weight map (:396-418, recomputed per pixel)
//   k_pyrdown     :469,:474 cv::pyrDown chain (image and weight)
//   k_lap_select  :469 + :476-555  pyrUp+subtract of createLaplacePyr fused
//                           with the per-tile max-weight select
//   Ele::blend (:77-146) and save (:806-840): collapse_fused.hip; their per-level form of rounds 1-5 (k_blend_* / k_collapse /
//   k_mosaic_gather / k_save_finish) is compiled into the experiments library only
#include "kernels.hpp"
#include "warp_index.hpp"
#include "env.hpp"
#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>

// PF_EXPERIMENTS=1 (a second library, libpifusion_exp.so: tests/test_gpu_variants.py and the A/B tools load it through PF_LIB) also
// compiles the forms of the level kernel that were built, measured and NOT adopted -- the wave-specialised rolling strips (strips.inc),
// the LDS-staged source patch (PATCH), 64x28 and 64x64 blocks, the fetch/finish row loops of rounds 2-4 (PF_A_ILP), the stamped
// instantiations and the timing-only ablation switches.  The product library carries the product instantiations only.
#ifndef PF_EXPERIMENTS
#define PF_EXPERIMENTS 0
#endif
#ifndef PF_ROWTAB            // A/B build switch: the row terms of the coordinates from a per-block LDS table (level3_block, deferred stage A)
#define PF_ROWTAB 0
#endif
#ifndef PF_HYBRID_W          // A/B build switch: see level3_block, deferred stage A
#define PF_HYBRID_W 0
#endif

namespace pf {

constexpr bool kExp = PF_EXPERIMENTS != 0;

// ---------------------------------------------------------------- helpers
// border maps, saturate_cast<short> and the warp's source addressing: warp_index.hpp (shared with the host-side check)
__device__ __forceinline__ double clamp_int_range(double v)
{
    v = (v < (double)INT_MAX) ? v : (double)INT_MAX;     // std::min((double)INT_MAX, v)
    v = ((double)INT_MIN < v) ? v : (double)INT_MIN;     // std::max((double)INT_MIN, v)
    return v;
}

// tile slots are reached through integers of the tile table: tell the compiler they are global
// memory, otherwise every tile access becomes a FLAT instruction
#define PF_GLOBAL __attribute__((address_space(1)))
// Build switches for the A/B of cache hints (does GW_1 of launch k survive in the 256 MiB Infinity Cache until
// launch k+1 reads it when the streams that are used once are marked non-temporal?  profiles/r04_ab.md):
//   PF_NT_STORES  Laplacian payload stores      PF_NT_W  stored-weight loads and stores      PF_NT_SRC  the warp's frame gathers
#ifdef PF_NT_STORES
#define PF_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define PF_STORE(p, v) (*(p) = (v))
#endif
#ifdef PF_NT_W
#define PF_STORE_W(p, v) __builtin_nontemporal_store((v), (p))
#define PF_LOAD_W(p) __builtin_nontemporal_load(p)
#else
#define PF_STORE_W(p, v) (*(p) = (v))
#define PF_LOAD_W(p) (*(p))
#endif
#ifdef PF_NT_SRC
#define PF_LOAD_SRC(p) __builtin_nontemporal_load(p)
#else
#define PF_LOAD_SRC(p) (*(p))
#endif

template <bool F32> struct Pix;
template <> struct Pix<false> { using T = short; using WT = int;   static constexpr int bytes = 2; };
template <> struct Pix<true>  { using T = float; using WT = float; static constexpr int bytes = 4; };

__device__ __forceinline__ short cast_down(int v)   { return (short)((v + 128) >> 8); }
__device__ __forceinline__ float cast_down(float v) { return v * (1.f / 256); }
__device__ __forceinline__ short cast_up(int v)     { return (short)((v + 32) >> 6); }
__device__ __forceinline__ float cast_up(float v)   { return v * (1.f / 64); }
__device__ __forceinline__ short sat_sub(short a, short b) { return (short)sat_short((int)a - (int)b); }
__device__ __forceinline__ float sat_sub(float a, float b) { return a - b; }
__device__ __forceinline__ short sat_add(short a, short b) { return (short)sat_short((int)a + (int)b); }
__device__ __forceinline__ float sat_add(float a, float b) { return a + b; }

// ------------------------------------------------------------------- warp
// One wave = one 64-pixel OpenCV coordinate block row (bw0 = 64): the block
// origin terms X0/Y0/W0 are wave-uniform, lane = x1.  4 waves = 4 rows.
template <bool F32>
__global__ __launch_bounds__(256) void k_warp(const uint8_t* __restrict__ src, WarpArgs a,
                                               void* __restrict__ g0v, float* __restrict__ w0)
{
    using T = typename Pix<F32>::T;
    __shared__ T stage[4][192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xb = a.x_off + blockIdx.x * 64;
    const int y  = a.y_off + blockIdx.y * 4 + wave;
    const int x1 = lane;

    const double X0 = a.M[0] * xb + a.M[1] * y + a.M[2];
    const double Y0 = a.M[3] * xb + a.M[4] * y + a.M[5];
    const double W0 = a.M[6] * xb + a.M[7] * y + a.M[8];
    const double W  = W0 + a.M[6] * x1;
    const double xn = X0 + a.M[0] * x1, yn = Y0 + a.M[3] * x1;

    // --- weight: INTER_NEAREST, BORDER_CONSTANT(0), analytic radial weight
    float wv = 0.f;
    {
        const double Wn = W ? 1. / W : 0;
        const int X = __double2int_rn(clamp_int_range(xn * Wn));
        const int Y = __double2int_rn(clamp_int_range(yn * Wn));
        const int sx = sat_short(X), sy = sat_short(Y);
        if ((unsigned)sx < (unsigned)a.scols && (unsigned)sy < (unsigned)a.srows) {
            const float dy = (float)sy - a.yc, dx = (float)sx - a.xc;
            float dis = dy * dy + dx * dx;
            dis = 1.f - sqrtf(dis) / a.dis_max;
            wv = a.weight_type == 0 ? dis : dis * dis;
            if ((double)wv <= 1e-5) wv = 1e-5f;
        }
    }
    // --- image: INTER_LINEAR (1/32 px), BORDER_REFLECT
    T out[3];
    {
        const double Wl = W ? 32. / W : 0;
        const int X = __double2int_rn(clamp_int_range(xn * Wl));
        const int Y = __double2int_rn(clamp_int_range(yn * Wl));
        const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
        const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
        const float c0 = (1.f - fy) * (1.f - fx), c1 = (1.f - fy) * fx, c2 = fy * (1.f - fx), c3 = fy * fx;
        int sx0 = sx, sx1 = sx + 1, sy0 = sy, sy1 = sy + 1;
        if (!((unsigned)sx < (unsigned)(a.scols - 1) && (unsigned)sy < (unsigned)(a.srows - 1))) {
            sx0 = border_reflect(sx, a.scols); sx1 = border_reflect(sx + 1, a.scols);
            sy0 = border_reflect(sy, a.srows); sy1 = border_reflect(sy + 1, a.srows);
        }
        const uint8_t* r0 = src + (long)sy0 * a.sstep;
        const uint8_t* r1 = src + (long)sy1 * a.sstep;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float v0 = (float)r0[sx0 * a.src_cn + k], v1 = (float)r0[sx1 * a.src_cn + k];
            float v2 = (float)r1[sx0 * a.src_cn + k], v3 = (float)r1[sx1 * a.src_cn + k];
            if (F32) {
                const float s = (float)(1. / 255.);
                v0 = v0 * s; v1 = v1 * s; v2 = v2 * s; v3 = v3 * s;
            }
            const float t = v0 * c0 + v1 * c1 + v2 * c2 + v3 * c3;
            if constexpr (F32) out[k] = t;
            else out[k] = (short)sat_short(__float2int_rn(t));
        }
    }
    // --- stores: weight is lane-contiguous; pixels go through LDS so that the
    //     wave writes whole dwords
    const long pix = (long)y * a.ccols + xb;
    w0[pix + lane] = wv;
    stage[wave][lane * 3 + 0] = out[0];
    stage[wave][lane * 3 + 1] = out[1];
    stage[wave][lane * 3 + 2] = out[2];
    __syncthreads();
    const uint32_t* st = reinterpret_cast<const uint32_t*>(stage[wave]);
    uint32_t* g = reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(g0v) + pix * 3);
    if constexpr (F32) {
        g[lane] = st[lane]; g[64 + lane] = st[64 + lane]; g[128 + lane] = st[128 + lane];
    } else {
        g[lane] = st[lane];
        if (lane < 32) g[64 + lane] = st[64 + lane];
    }
}

void launch_warp(hipStream_t s, bool f32, const uint8_t* src, const WarpArgs& a, void* g0, float* w0)
{
    dim3 grid(a.wcols / 64, a.wrows / 4), block(256);
    if (f32) hipLaunchKernelGGL(k_warp<true>, grid, block, 0, s, src, a, g0, w0);
    else     hipLaunchKernelGGL(k_warp<false>, grid, block, 0, s, src, a, g0, w0);
}

// ---------------------------------------------------------------- pyrDown
// dst tile TW x TH pixels per workgroup; horizontal 5-tap into LDS for the
// 2*TH+3 source rows, then the vertical 5-tap from LDS.
template <typename T, typename WT, int CN>
__global__ __launch_bounds__(256) void k_pyrdown(const T* __restrict__ src, int srows, int scols,
                                                  T* __restrict__ dst, int drows, int dcols,
                                                  int y0, int y1, int x0, int x1)
{
    constexpr int TW = 32, TH = 16, EW = TW * CN, SR = 2 * TH + 3;
    __shared__ WT hbuf[SR][EW];
    const int ox = x0 + blockIdx.x * TW, oy = y0 + blockIdx.y * TH;
    for (int idx = threadIdx.x; idx < SR * EW; idx += 256) {
        const int r = idx / EW, e = idx - r * EW;
        const int x = ox + e / CN, k = e % CN;
        if (x >= x1) continue;
        const int sy = border_reflect101(2 * oy - 2 + r, srows);
        const T* s = src + (long)sy * scols * CN + k;
        int i0 = 2 * x - 2, i1 = 2 * x - 1, i2 = 2 * x, i3 = 2 * x + 1, i4 = 2 * x + 2;
        if (i0 < 0 || i4 >= scols) {
            i0 = border_reflect101(i0, scols); i1 = border_reflect101(i1, scols); i2 = border_reflect101(i2, scols);
            i3 = border_reflect101(i3, scols); i4 = border_reflect101(i4, scols);
        }
        hbuf[r][e] = (WT)s[i2 * CN] * 6 + ((WT)s[i1 * CN] + (WT)s[i3 * CN]) * 4 + (WT)s[i0 * CN] + (WT)s[i4 * CN];
    }
    __syncthreads();
    const int vec_end = (dcols * CN / 8) * 8;      // PyrDownVec_32f covers full groups of 8 floats
    for (int idx = threadIdx.x; idx < TH * EW; idx += 256) {
        const int r = idx / EW, e = idx - r * EW;
        const int y = oy + r, x = ox + e / CN;
        if (y >= y1 || x >= x1) continue;
        const WT r0 = hbuf[2 * r][e], r1 = hbuf[2 * r + 1][e], r2 = hbuf[2 * r + 2][e], r3 = hbuf[2 * r + 3][e],
                 r4 = hbuf[2 * r + 4][e];
        const int ge = ox * CN + e;
        T o;
        if constexpr (sizeof(T) == 4) {
            if (ge < vec_end) {
                WT a = r0 + r4;
                WT b = (r1 + r3) + r2;
                a = a + (r2 + r2);
                o = (a + b * 4.f) * (1.f / 256);
            } else
                o = cast_down(r2 * 6 + (r1 + r3) * 4 + r0 + r4);
        } else
            o = cast_down(r2 * 6 + (r1 + r3) * 4 + r0 + r4);
        dst[(long)y * dcols * CN + ge] = o;
    }
}

void launch_pyrdown(hipStream_t s, int type, const void* src, int srows, int scols, void* dst,
                    int y0, int y1, int x0, int x1)
{
    const int drows = (srows + 1) / 2, dcols = (scols + 1) / 2;
    if (y1 <= y0 || x1 <= x0) return;
    dim3 grid((x1 - x0 + 31) / 32, (y1 - y0 + 15) / 16), block(256);
    if (type == 0)
        hipLaunchKernelGGL((k_pyrdown<short, int, 3>), grid, block, 0, s, (const short*)src, srows, scols, (short*)dst, drows, dcols, y0, y1, x0, x1);
    else if (type == 1)
        hipLaunchKernelGGL((k_pyrdown<float, float, 3>), grid, block, 0, s, (const float*)src, srows, scols, (float*)dst, drows, dcols, y0, y1, x0, x1);
    else
        hipLaunchKernelGGL((k_pyrdown<float, float, 1>), grid, block, 0, s, (const float*)src, srows, scols, (float*)dst, drows, dcols, y0, y1, x0, x1);
}

// ------------------------------------------------------------------ pyrUp
// value of pyrUp(src)[y][x][k] for a 2x destination (pyramids.cpp pyrUp_):
// horizontal  even: s[x-1] + s[x]*6 + s[x+1]   odd: (s[x]+s[x+1])*4
//   left edge even: s[0]*6 + s[1]*2 ; right edge even: s[n-2] + s[n-1]*7, odd: s[n-1]*8
// vertical    even: r0 + r1*6 + r2             odd: (r1+r2)*4   rows: -1 -> 1, n -> n-1
template <typename T, typename WT>
__device__ __forceinline__ WT up_h(const T* __restrict__ row, int x, int scols)
{
    const int sx = x >> 1;
    if (scols == 1) return (WT)row[0] * 8;
    if (x & 1) {
        if (sx == scols - 1) return (WT)row[sx * 3] * 8;
        return ((WT)row[sx * 3] + (WT)row[(sx + 1) * 3]) * 4;
    }
    if (sx == 0) return (WT)row[0] * 6 + (WT)row[3] * 2;
    if (sx == scols - 1) return (WT)row[(sx - 1) * 3] + (WT)row[sx * 3] * 7;
    return (WT)row[(sx - 1) * 3] + (WT)row[sx * 3] * 6 + (WT)row[(sx + 1) * 3];
}

template <typename T, typename WT>
__device__ __forceinline__ T pyr_up_at(const T* __restrict__ src, int srows, int scols, int y, int x, int k)
{
    const int sy = y >> 1;
    const T* r1 = src + (long)sy * scols * 3 + k;
    int syn = sy + 1; if (syn >= srows) syn = srows - 1;
    const T* r2 = src + (long)syn * scols * 3 + k;
    if (y & 1) return cast_up((up_h<T, WT>(r1, x, scols) + up_h<T, WT>(r2, x, scols)) * 4);
    int syp = sy - 1; if (syp < 0) syp = srows > 1 ? 1 : 0;
    const T* r0 = src + (long)syp * scols * 3 + k;
    return cast_up(up_h<T, WT>(r0, x, scols) + up_h<T, WT>(r1, x, scols) * 6 + up_h<T, WT>(r2, x, scols));
}

// ------------------------------------------------- Laplacian + tile select
template <bool F32>
__global__ __launch_bounds__(256) void k_lap_select(TileLayout lay, int level, const void* __restrict__ giv,
                                                     const void* __restrict__ gupv, const float* __restrict__ wi,
                                                     int rows, int cols, const uint64_t* __restrict__ table,
                                                     int tiles_x, int py0, int px0, int px1, int py1)
{
    using T = typename Pix<F32>::T; using WT = typename Pix<F32>::WT;
    const int x = px0 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = py0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= px1 || y >= py1) return;
    const int sh = 8 - level, ts = kElePixels >> level;
    const uint64_t ent = table[(y >> sh) * tiles_x + (x >> sh)];
    if (!ent) return;                                  // tile not owned by this shard
    char* slot = reinterpret_cast<char*>(ent & ~(uint64_t)1);
    const bool fresh = ent & 1;
    const int loc = (y & (ts - 1)) * ts + (x & (ts - 1));
    float* dw = reinterpret_cast<float*>(slot + lay.w_off[level]) + loc;
    const long pix = (long)y * cols + x;
    const float sw = wi[pix];
    if (!fresh && !(sw >= *dw)) return;
    const T* gi = reinterpret_cast<const T*>(giv) + pix * 3;
    T* dl = reinterpret_cast<T*>(slot + lay.lap_off[level]) + loc * 3;
    if (gupv) {
        const T* gup = reinterpret_cast<const T*>(gupv);
        const int srows = rows >> 1, scols = cols >> 1;
#pragma unroll
        for (int k = 0; k < 3; k++) dl[k] = sat_sub(gi[k], pyr_up_at<T, WT>(gup, srows, scols, y, x, k));
    } else {
        dl[0] = gi[0]; dl[1] = gi[1]; dl[2] = gi[2];
    }
    *dw = sw;
}

void launch_lap_select(hipStream_t s, const TileLayout& lay, int level, const void* g_i, const void* g_up,
                       const float* w_i, int rows, int cols, const uint64_t* tile_table, int tiles_x,
                       int ty0, int ty1, int tx0, int tx1)
{
    const int ts = kElePixels >> level;
    const int px0 = tx0 * ts, px1 = tx1 * ts, py0 = ty0 * ts, py1 = ty1 * ts;
    if (px1 <= px0 || py1 <= py0) return;
    dim3 grid((px1 - px0 + 63) / 64, (py1 - py0 + 3) / 4), block(256);
    if (lay.f32) hipLaunchKernelGGL(k_lap_select<true>, grid, block, 0, s, lay, level, g_i, g_up, w_i, rows, cols, tile_table, tiles_x, py0, px0, px1, py1);
    else         hipLaunchKernelGGL(k_lap_select<false>, grid, block, 0, s, lay, level, g_i, g_up, w_i, rows, cols, tile_table, tiles_x, py0, px0, px1, py1);
}

// ============================================================ fused level
// k_level: one launch per pyramid level i < L does, for a 64x32 block of level-i
// pixels, everything that needs G_i:
//   A  G_i / W_i on the block + halo (4 left/top, 3 right/bottom) into LDS --
//      level 0: computed by the warp (never written to HBM); level >= 1: read
//      from the packed GW_i buffer the previous launch wrote
//   H  horizontal 5-tap decimation of A                      (cv::pyrDown)
//   B  vertical 5-tap -> G_{i+1} / W_{i+1} on the half-size block + 1 halo;
//      the block's own part goes to GW_{i+1} (or straight into the top-level
//      select when i+1 == L)
//   D  L_i = G_i - pyrUp(G_{i+1}) from LDS, max-weight select into the tiles
// Out-of-canvas halo entries hold the BORDER_REFLECT_101 pixel, so H/B need
// no border logic; pyrUp's asymmetric edge rules are applied on global
// coordinates.  All arithmetic orders are those of the unfused kernels.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool F32> struct PxT;
template <> struct alignas(16) PxT<true>  { float c[3]; float w; };
template <> struct alignas(4)  PxT<false> { short c[3]; short pad; float w; };
template <bool F32> struct alignas(16) HxT { typename Pix<F32>::WT c[3]; float w; };

// int16 pyramids, stages B and D in packed 16-bit arithmetic (two channels per instruction).  Exact: Gaussian levels of an
// 8-bit frame stay in [0, 255], so the 5-tap sums (<= 255*16), their vertical sums (<= 255*256 = 65280, unsigned 16 bits
// hold it, +128 included), the pyrUp sums (<= 255*64 = 16320) and the Laplacian (|.| <= 255) never leave 16 bits -- the
// int32 arithmetic of pyramids.cpp gives the same bits ((v+128)>>8, (v+32)>>6, no saturation reached).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
struct __attribute__((may_alias)) alignas(4) PxP { us2 c01, c2p; float w; };       // PxT<false> seen as two channel pairs
struct __attribute__((may_alias)) alignas(4) BxP { ss2 c01, c2p; };                // Bx of the int16 build (c[4])

constexpr int LBW = 64, LAW = LBW + 7, LQW = LBW / 2 + 2;      // block width, staged width, half-size width (+halo)

// per-launch scalars kept small on purpose: the warp stage is SGPR-hungry, and whole TileLayout /
// WarpArgs structs as kernel arguments made the compiler re-load kernel arguments inside the pixel loop
struct LevelOffsets { uint32_t lap_off, w_off, top_lap_off, top_w_off; };   // byte offsets inside a tile slot
struct FusedWarp {
    double M[9];            // destination -> source map
    long   total;           // bytes of the frame that may be read (frame_bytes: the last row ends at its last pixel)
    const float* wmap;      // srows x scols radial weight plane (the reference's weightImage)
    int    srows, scols, sstep, cn;
    int    plain;           // host-checked: |M| entries < 2^400 and the frame is at most 32767 px on a side, so W
                            // cannot leave the mid range upwards and saturate_cast<short> never alters an
                            // in-frame coordinate (0: every pixel takes the general forms)
    // LDS-staged source patch (k_levels<..., PATCH = true>, see patch_plan): a level-0 block stages the part of the frame
    // and of the weight plane its 71x39 warped pixels read -- source rectangle [c - h, c + h + 1] around the image c of the
    // block's centre -- with coalesced 16-byte loads; the bilinear taps and the weight then come from LDS.
    // radial weight evaluated in the kernel instead of gathered from the plane (k_levels<..., WA = true>): weightImage's
    // formula (MultiBandMap2DCPU.cpp:403-417) with a correctly rounded square root and quotient, see radial_weight()
    float  wxc, wyc, wdmax, wrcp;   // x_center, y_center, dis_max, RN(1 / dis_max)
    int    wtype;                   // Map2D.WeightType
    int    seed_ok;                 // host-checked (seed_plan): W changes by at most 2^-15 of itself from one canvas row to the next, anywhere on the
                                    // canvas -- the reciprocal of a pixel's W may then start from the reciprocal of the pixel above it (rcp_seeded)
    float  Mf[9];           // M in fp32: only places the patch (a pixel outside it takes the global-memory path)
    int    phx, phy;        // half extents of the patch, source pixels
    int    pitch_i, pitch_w;// LDS row pitches of the frame patch and of the weight patch, bytes (multiples of 16)
};

struct LevelArgs {
    int level, rows, cols;        // level i and its canvas extent
    int cx0, cy0, cx1, cy1;       // compute region (block grid origin / extent)
    int tiles_x;
    int top_select;               // i+1 == L: select the top level from B
    int write_next;               // i+1 <  L: write GW_{i+1}
    int nbx, nby;                 // block grid (k_strips: strips x segments)
    int ablate;                   // diagnostics only (PF_ABLATE): bit0 skip A math, bit1 skip H/B, bit2 skip U/D
    int seg;                      // k_strips: rows of a strip segment
    unsigned inv_nbx;             // ceil(2^32 / nbx): block id -> (bx, by) by one multiply (a scalar division is ~20 dependent instructions, twice per workgroup)
    int pad_;
};
// by = b / nbx by one multiply: with inv = ceil(2^32 / nbx), floor(b * inv / 2^32) = floor(b / nbx) while b * (inv * nbx - 2^32) < 2^32, which
// nblk * nbx < 2^32 guarantees (level_inv_nbx); inv_nbx == 0: the division itself
template <class GA>      // LevelArgs in registers / on the stack, or where it lies in the kernel arguments (address space 4)
__device__ __forceinline__ void block_xy(const GA& g, int b, int& bx, int& by)
{
    if (g.inv_nbx) by = (int)__umulhi((unsigned)b, g.inv_nbx); else by = b / g.nbx;
    bx = b - by * g.nbx;
}

// 1/d exactly as the compiler's IEEE fp64 division computes it when no operand scaling is needed
// (v_div_scale_f64 and v_div_fixup_f64 are identities for 2^-500 < |d| < 2^500, v_div_fmas_f64 is a plain
// fma): v_rcp_f64, two Newton steps, quotient q = 1*r, one remainder correction -> correctly rounded.
__device__ __forceinline__ double rcp_mid_range(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(rem, r, r);
}

// The same reciprocal from a SEED instead of v_rcp_f64 (16 issue cycles): r0 = RN(1 / d0) for a d0 within 2^-15 of d (relative).  Two Newton steps
// take the seed's error 2^-15 to 2^-60, i.e. to the fma roundings' own 2^-53 -- exactly where v_rcp_f64 and two steps leave rcp_mid_range -- and
// the same remainder correction rounds it; whatever makes that sequence the correctly rounded reciprocal makes this one so (checked against exact
// rational arithmetic for seeds up to 2^-11 away: tests/test_oracle_ops.py::test_seeded_reciprocal_is_correctly_rounded).
__device__ __forceinline__ double rcp_seeded(double d, double r)
{
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(rem, r, r);
}

// the terms of the coordinate that depend on the canvas column only: xb = x & ~63 is OpenCV's 64-wide
// coordinate block, x1 = x & 63 the offset inside it
struct WarpCol { double m0xb, m3xb, m6xb, m0x1, m3x1, m6x1; };
__device__ __forceinline__ WarpCol warp_col(const FusedWarp& a, int x)
{
    const int xb = x & ~63, x1 = x & 63;
    return { a.M[0] * xb, a.M[3] * xb, a.M[6] * xb, a.M[0] * x1, a.M[3] * x1, a.M[6] * x1 };
}

// A warped pixel is produced in two steps so that a thread can have the loads of its next pixel in flight
// while it finishes the current one: warp_fetch (coordinates, border mapping, exactly three loads, no load
// under divergent control flow) and warp_finish (tap extraction, bilinear sum).
struct WarpTaps {
    uint32_t lo0, hi0, lo1, hi1;     // 8 bytes of each of the two source rows
    float    wraw;                   // weight plane entry (valid when kInb is set)
    int      X, Y;                   // 1/32-px source coordinate (the low 5 bits are the bilinear fractions)
    uint32_t flags;
};

// weightImage's entry for source pixel (sx, sy), computed: dis = (sy-yc)^2 + (sx-xc)^2; w = 1 - sqrt(dis) / dis_max
// (squared for WeightType 1), floored at (float)1e-5 -- every step rounded exactly as the reference's float code rounds it:
//   * sqrt: v_sqrt_f32 is within 1 ulp; the neighbour test (is the float below / above the better root?) by two exact
//     fma residuals makes it the correctly rounded root (the compiler's own IEEE sqrt sequence, without its denormal
//     scaling: dis is 0 or >= 1);
//   * quotient by the constant dis_max: q0 = s * RN(1/c), exact residual r = s - q0 * c, q = q0 + r * RN(1/c) -- correctly
//     rounded (Markstein) for q0 within 1 ulp;
//   * `if (*p <= 1e-5) *p = 1e-5` compares in double and stores (float)1e-5, the float just below 1e-5: that is max(w, 1e-5f).
// tests: every weight of every parity test is compared bit for bit with the oracle's weightImage gather.
__device__ __forceinline__ float radial_weight(const FusedWarp& a, int sx, int sy)
{
    const float dy = (float)sy - a.wyc, dx = (float)sx - a.wxc;
    const float dis = __builtin_fmaf(dx, dx, dy * dy);          // both products are exact (|d| < 2^12): one rounding, as dy*dy + dx*dx
    float s = __builtin_amdgcn_sqrtf(dis);
    {
        const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
        const float ed = __builtin_fmaf(-sd, s, dis), eu = __builtin_fmaf(-su, s, dis);
        s = ed <= 0.f ? sd : s;
        s = eu > 0.f ? su : s;
    }
    const float q0 = s * a.wrcp;
    const float q = __builtin_fmaf(__builtin_fmaf(-q0, a.wdmax, s), a.wrcp, q0);
    float w = 1.f - q;
    if (a.wtype != 0) w = w * w;
    return fmaxf(w, 1e-5f);
}

// X0, Y0, W0: the coordinate terms of the pixel's row and 64-wide coordinate block (M0*xb + M1*y + M2 ...), formed by the
// caller: per pixel (warp_fetch) or once per workgroup in a row table (level3_block, ILP 4)
// WA: the radial weight is computed (radial_weight) instead of gathered from the weight plane
template <bool WA = false>
__device__ __forceinline__ WarpTaps warp_fetch_pre(const uint8_t* __restrict__ src, const FusedWarp& a, const WarpCol& col,
                                                   const double X0, const double Y0, const double W0)
{
    const double W  = W0 + col.m6x1;
    const double xn = X0 + col.m0x1, yn = Y0 + col.m3x1;
    // nearest coordinate p = (X0+M0*x1)*(1/W); the 1/32-px coordinate (X0+M0*x1)*(32/W) equals 32*p
    // bit for bit (32/W == 32*(1/W) and scaling by a power of two commutes with rounding).
    // Common case, decided per wave: W in the mid range (reciprocal without the scale/fixup steps) and both
    // coordinates far inside the int range, where round-half-even comes from one fp64 add of 1.5*2^52 whose
    // low dword IS the integer.  Anything else (W == 0, degenerate homographies) takes the general forms;
    // v_cvt_i32_f64 saturates, which is exactly clamp-to-int-range followed by cvRound.
    int Xn, Yn, X, Y;
    {
        // |W| < 2^-500 (W == 0 included) ends in a huge or NaN product and fails the range test by itself;
        // the upper end is excluded on the host (a.plain)
        const double Wn = rcp_mid_range(W);
        const double pxn = xn * Wn, pyn = yn * Wn;
        const bool tame = fabs(pxn) < 3.0e7 && fabs(pyn) < 3.0e7;
        if (a.plain && __builtin_amdgcn_ballot_w64(!tame) == 0) {
            constexpr double kMagic = 6755399441055744.0;
            Xn = (int)(uint32_t)(unsigned long long)__double_as_longlong(pxn + kMagic);
            Yn = (int)(uint32_t)(unsigned long long)__double_as_longlong(pyn + kMagic);
            X  = (int)(uint32_t)(unsigned long long)__double_as_longlong(pxn * 32. + kMagic);
            Y  = (int)(uint32_t)(unsigned long long)__double_as_longlong(pyn * 32. + kMagic);
        } else {
            const double Wd = W ? 1. / W : 0;
            const double qx = xn * Wd, qy = yn * Wd;
            Xn = __double2int_rn(qx); Yn = __double2int_rn(qy);
            X = __double2int_rn(qx * 32.); Y = __double2int_rn(qy * 32.);
        }
    }
    WarpTaps t;
    uint32_t flags;
    typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(1)));
    {
        // weight: INTER_NEAREST, BORDER_CONSTANT 0 (out of bounds reads entry 0 and is zeroed in warp_finish).
        // remap's saturate_cast<short> of the coordinate only matters for frames wider than a short.
        int sx = Xn, sy = Yn;
        if (!a.plain) { sx = sat_short(sx); sy = sat_short(sy); }
        const bool inb = (unsigned)sx < (unsigned)a.scols && (unsigned)sy < (unsigned)a.srows;
        if constexpr (WA) t.wraw = radial_weight(a, sx, sy);
        else {
            const uint32_t woff = inb ? (uint32_t)(__mul24(sy, a.scols) + sx) << 2 : 0u;
            t.wraw = *(const float*)((const char*)a.wmap + woff);
        }
        flags = inb ? kInb : 0u;
    }
    // one unaligned 8-byte load per source row fetches both taps: pixel "lo" = bytes 0..2, "hi" = bytes cn..cn+2.
    // Frames are < 2 GiB and rows/steps fit 24 bits: 32-bit unsigned offsets from the frame base, full-rate
    // 24-bit multiplies.
    const int cn = a.cn;                                // 3 (BGR) or 4 (BGRA, alpha skipped)
    // strictly inside the frame and not on its last two rows: both 8-byte reads stay inside the buffer
    const int ux = X >> 5, uy = Y >> 5;
    TapAddr ta;
    if (a.plain && __builtin_amdgcn_ballot_w64(!tap_is_fast(ux, uy, a.srows, a.scols)) == 0)
        ta = tap_addr_fast(ux, uy, a.sstep, cn);
    else {
        // canvas pixels next to the frame, decided per wave: one reflection, no division
        const bool near = a.plain && __builtin_amdgcn_ballot_w64(!tap_is_near(ux, uy, a.srows, a.scols)) == 0;
        ta = tap_addr_border(ux, uy, near, a.srows, a.scols, a.sstep, cn, (uint32_t)a.total);
    }
    const uint32_t off0 = ta.off0, off1 = ta.off1;
    flags |= ta.flags;
    const u2 b0 = PF_LOAD_SRC((const u2*)(src + off0)), b1 = PF_LOAD_SRC((const u2*)(src + off1));
    t.lo0 = b0.x; t.hi0 = b0.y; t.lo1 = b1.x; t.hi1 = b1.y;
    t.X = X; t.Y = Y; t.flags = flags;
    return t;
}

template <bool WA = false>
__device__ __forceinline__ WarpTaps warp_fetch(const uint8_t* __restrict__ src, const FusedWarp& a, const WarpCol& col, int y)
{
    return warp_fetch_pre<WA>(src, a, col, col.m0xb + a.M[1] * y + a.M[2], col.m3xb + a.M[4] * y + a.M[5], col.m6xb + a.M[7] * y + a.M[8]);
}

// The bilinear sum of a pixel whose four taps lie strictly inside the frame (tap_is_fast): (lo, hi) pixels of the two 8-byte row loads,
// fractions fx, fy in 1/32 px, weight w -- remapBilinear's float form, evaluated left to right without contraction.
template <bool F32>
__device__ __forceinline__ PxT<F32> warp_finish_fast(uint32_t lo0, uint32_t hiw0, uint32_t lo1, uint32_t hiw1, int fxi, int fyi, float w, int cn)
{
    PxT<F32> o;
    o.w = w;
    const float fx = (float)fxi * (1.f / 32), fy = (float)fyi * (1.f / 32);
    const float c0 = (1.f - fy) * (1.f - fx), c1 = (1.f - fy) * fx, c2 = fy * (1.f - fx), c3 = fy * fx;
    const uint32_t hisel = cn == 3 ? 0x06050403u : 0x07060504u;      // v_perm_b32: bytes cn..cn+3 of the 8
    const uint32_t hi0 = __builtin_amdgcn_perm(hiw0, lo0, hisel), hi1 = __builtin_amdgcn_perm(hiw1, lo1, hisel);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v0 = (float)((lo0 >> (8 * k)) & 0xff), v1 = (float)((hi0 >> (8 * k)) & 0xff);
        float v2 = (float)((lo1 >> (8 * k)) & 0xff), v3 = (float)((hi1 >> (8 * k)) & 0xff);
        if constexpr (F32) {
            const float s = (float)(1. / 255.);
            v0 = v0 * s; v1 = v1 * s; v2 = v2 * s; v3 = v3 * s;
            o.c[k] = v0 * c0 + v1 * c1 + v2 * c2 + v3 * c3;
        } else {
            // every product and partial sum is a multiple of 2^-10 below 2^8: exact in fp32 whatever the association, fused or not
            const float tt = __builtin_fmaf(v3, c3, __builtin_fmaf(v2, c2, __builtin_fmaf(v1, c1, v0 * c0)));
            o.c[k] = (short)sat_short(__float2int_rn(tt));
        }
    }
    if constexpr (!F32) o.pad = 0;
    return o;
}

template <bool F32>
__device__ __forceinline__ PxT<F32> warp_finish(const WarpTaps& t, int cn)
{
    PxT<F32> o;
    const uint32_t flags = t.flags;
    o.w = (flags & kInb) ? t.wraw : 0.f;
    const float fx = (float)(t.X & 31) * (1.f / 32), fy = (float)(t.Y & 31) * (1.f / 32);
    const float c0 = (1.f - fy) * (1.f - fx), c1 = (1.f - fy) * fx, c2 = fy * (1.f - fx), c3 = fy * fx;
    const uint32_t hisel = cn == 3 ? 0x06050403u : 0x07060504u;      // v_perm_b32: bytes cn..cn+3 of the 8
#ifndef PF_INT_BILINEAR        // build switch, off: measured 2 % slower than the float form (profiles/r04_ab.md)
#define PF_INT_BILINEAR 0
#endif
    if constexpr (!F32 && PF_INT_BILINEAR != 0) {
        if (__builtin_amdgcn_ballot_w64((flags & ~kInb) != kT1Hi) == 0) {
            // int16, whole wave strictly inside the frame: the bilinear sum on integers (round 4).  With a = 32 - fx, b = 32 - fy
            // the reference's float sum is V / 1024 exactly, V = (S00*a + S01*fx) * b + (S10*a + S11*fx) * fy (every product and
            // partial sum of the float form is a multiple of 2^-10 below 2^8), and cvRound of it is round-half-even of V / 1024.
            // Taps of a channel are gathered into one word by byte permutes (7 for the three channels), the two horizontal sums are
            // v_dot4_u32_u8 against (a, fx, 0, 0) / (0, 0, a, fx), the vertical one two 24-bit multiply-adds.
            const uint32_t fx = (uint32_t)t.X & 31u, fy = (uint32_t)t.Y & 31u;
            const uint32_t B1 = (32u - fx) | fx << 8, B2 = B1 << 16, wy0 = 32u - fy;
            const uint32_t s01 = cn == 3 ? 0x04010300u : 0x05010400u;      // bytes (lo.c0, hi.c0, lo.c1, hi.c1) of a row's 8: hi pixel at byte cn
            const uint32_t s2 = cn == 3 ? 0x0c0c0502u : 0x0c0c0602u;       //       (lo.c2, hi.c2, 0, 0)
            const uint32_t r0a = __builtin_amdgcn_perm(t.hi0, t.lo0, s01), r0b = __builtin_amdgcn_perm(t.hi0, t.lo0, s2);
            const uint32_t r1a = __builtin_amdgcn_perm(t.hi1, t.lo1, s01), r1b = __builtin_amdgcn_perm(t.hi1, t.lo1, s2);
            const uint32_t A[3] = { __builtin_amdgcn_perm(r1a, r0a, 0x05040100u),       // (S00, S01, S10, S11) of channel 0
                                    __builtin_amdgcn_perm(r1a, r0a, 0x07060302u),       // channel 1
                                    __builtin_amdgcn_perm(r1b, r0b, 0x05040100u) };     // channel 2
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t h0 = __builtin_amdgcn_udot4(A[k], B1, 0u, false), h1 = __builtin_amdgcn_udot4(A[k], B2, 0u, false);
                const uint32_t V = __umul24(h0, wy0) + __umul24(h1, fy);
                o.c[k] = (short)((V + 511u + ((V >> 10) & 1u)) >> 10);                 // round half to even of V / 1024; 0..255: saturate_cast<short> is the identity
            }
            o.pad = 0;
            return o;
        }
    }
    float v[4][3];                           // taps (sy,sx) (sy,sx+1) (sy+1,sx) (sy+1,sx+1) after border mapping
    if (__builtin_amdgcn_ballot_w64((flags & ~kInb) != kT1Hi) == 0) {
        // whole wave strictly inside the frame: taps are (lo, hi) of each row
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t lo = j ? t.lo1 : t.lo0, hi = __builtin_amdgcn_perm(j ? t.hi1 : t.hi0, lo, hisel);
            v[2 * j][0] = (float)(lo & 0xff); v[2 * j][1] = (float)((lo >> 8) & 0xff); v[2 * j][2] = (float)((lo >> 16) & 0xff);
            v[2 * j + 1][0] = (float)(hi & 0xff); v[2 * j + 1][1] = (float)((hi >> 8) & 0xff); v[2 * j + 1][2] = (float)((hi >> 16) & 0xff);
        }
    } else {
        const bool t0hi = flags & kT0Hi, t1hi = flags & kT1Hi;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            uint32_t lo, hi;
            row_taps(j ? t.lo1 : t.lo0, j ? t.hi1 : t.hi0, (flags >> (j ? kBack1 : kBack0)) & 7, cn, lo, hi);
            const float l0 = (float)(lo & 0xff), l1 = (float)((lo >> 8) & 0xff), l2 = (float)((lo >> 16) & 0xff);
            const float h0 = (float)(hi & 0xff), h1 = (float)((hi >> 8) & 0xff), h2 = (float)((hi >> 16) & 0xff);
            v[2 * j][0] = t0hi ? h0 : l0; v[2 * j][1] = t0hi ? h1 : l1; v[2 * j][2] = t0hi ? h2 : l2;
            v[2 * j + 1][0] = t1hi ? h0 : l0; v[2 * j + 1][1] = t1hi ? h1 : l1; v[2 * j + 1][2] = t1hi ? h2 : l2;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v0 = v[0][k], v1 = v[1][k], v2 = v[2][k], v3 = v[3][k];
        if (F32) {
            const float s = (float)(1. / 255.);
            v0 = v0 * s; v1 = v1 * s; v2 = v2 * s; v3 = v3 * s;
        }
        if constexpr (F32) o.c[k] = v0 * c0 + v1 * c1 + v2 * c2 + v3 * c3;
        else {
            // int16: taps are integers <= 255 and the weights multiples of 2^-10, so every product and every partial sum is
            // a multiple of 2^-10 below 2^8 -- exact in fp32 whatever the association, fused or not.  Three fused
            // multiply-adds give the same bits as the reference's mul/add chain; only the final cvRound rounds.
            const float tt = __builtin_fmaf(v3, c3, __builtin_fmaf(v2, c2, __builtin_fmaf(v1, c1, v0 * c0)));
            o.c[k] = (short)sat_short(__float2int_rn(tt));
        }
    }
    if constexpr (!F32) o.pad = 0;
    return o;
}

typedef uint32_t park8 __attribute__((ext_vector_type(2), aligned(8)));      // what stage A parks in a pixel's LDS slot: fractions, weight
typedef uint32_t park4 __attribute__((ext_vector_type(2), aligned(4)));

// Does the canvas rectangle [x0, x0 + w) x [y0, y0 + h) map strictly inside the frame -- every pixel's four bilinear taps and its nearest
// pixel inside it, not on the last two rows (tap_is_fast), coordinates in the tame range (warp_fetch_pre)?  Decided once per workgroup, the
// same in every wave: lane l evaluates corner l & 3 exactly as the pixel there is evaluated.  W is affine, so one sign at the four corners is
// one sign on the rectangle, where the map is then projective and takes it to the convex quadrilateral of the corners' images; a
// quadrilateral whose corners keep ONE pixel of margin from the frame's fast region holds every pixel's computed coordinate (which differs from
// its exact image by rounding only: below 1e-8 px for frames of at most 32767 px, a.plain).
__device__ __forceinline__ bool block_maps_inside(const FusedWarp& a, int x0, int y0, int w, int h)
{
    const int lane = threadIdx.x & 3;
    const int x = x0 + ((lane & 1) ? w - 1 : 0), y = y0 + ((lane & 2) ? h - 1 : 0);
    const WarpCol col = warp_col(a, x);
    const double X0 = col.m0xb + a.M[1] * y + a.M[2], Y0 = col.m3xb + a.M[4] * y + a.M[5], W0 = col.m6xb + a.M[7] * y + a.M[8];
    const double W = W0 + col.m6x1, xn = X0 + col.m0x1, yn = Y0 + col.m3x1;
    const double Wn = rcp_mid_range(W);
    const double px = xn * Wn, py = yn * Wn;
    const bool ok = px >= 1.0 && px <= (double)(a.scols - 3) && py >= 1.0 && py <= (double)(a.srows - 4) && fabs(W) > 0x1p-400;
    const unsigned long long pos = __builtin_amdgcn_ballot_w64(W > 0.0);
    return __builtin_amdgcn_ballot_w64(!ok) == 0 && (pos == 0 || pos == __builtin_amdgcn_ballot_w64(true));
}

// one canvas pixel of the warp: image (LINEAR, REFLECT) + weight (NEAREST, CONSTANT 0)
template <bool F32>
__device__ __forceinline__ PxT<F32> warp_pixel(const uint8_t* __restrict__ src, const FusedWarp& a, const WarpCol& col, int y)
{
    return warp_finish<F32>(warp_fetch(src, a, col, y), a.cn);
}

// horizontal pyrUp term of one G_{i+1} row (pyramids.cpp pyrUp_): a,b,c = s[sx-1], s[sx], s[sx+1].
// EDGE = the block touches the canvas' left/right edge, where the fp32 forms differ from the
// interior ones (s[0]*6+s[1]*2, s[n-2]+s[n-1]*7, s[n-1]*8); for int16 they are the same integers
// as the interior form on reflected/replicated indices, which the caller supplies.
template <typename WT, bool EDGE>
__device__ __forceinline__ WT up_h_val(WT a, WT b, WT c, bool odd, bool le, bool re, bool one)
{
    if (odd) {
        WT t = (b + c) * 4;
        if (EDGE) { if (re || one) t = b * 8; }
        return t;
    }
    WT t = a + b * 6 + c;
    if (EDGE) {
        if (le) t = b * 6 + c * 2;
        if (re) t = a + b * 7;
        if (one) t = b * 8;
    }
    return t;
}

// A tile-table entry = slot address (256-byte aligned, below 2^48) | flags: bit 0 fresh (first write copies unconditionally), bits 48..63
// the CELLS of the tile (64 x 64 level-0 pixels each, bit 48 + 4 * row + column) in which this keyframe cannot win the max-weight select
// at any level (the cull of FusionMap::render_frame): pixels there are not looked at -- they may have been computed from input that was
// never produced.  A pixel of level i belongs to the cell that holds its level-0 origin (x << i, y << i).  Entries written without the
// cull carry no such bits.
constexpr uint64_t kEntFlags = 0xffff0000000000ffull, kEntCells = 0xffff000000000000ull, kEntLow = 0xffull;
__device__ __forceinline__ bool cell_culled(uint64_t ent, int x, int y, int ts, int sh)          // ts = 1 << sh: tile edge at this level
{
    const int c = ((((y & (ts - 1)) << 2) >> sh) << 2) | (((x & (ts - 1)) << 2) >> sh);
    return (((uint32_t)(ent >> 32) >> (16 + c)) & 1u) != 0;            // the high word alone: no 64-bit shift
}

// max-weight select of one pixel into its tile (Apply loop body, .cpp:496-551)
template <bool F32, typename TablePtr>
__device__ __forceinline__ void select_store(uint32_t lap_off, uint32_t w_off, int level, TablePtr table, int tiles_x,
                                             int x, int y, const typename Pix<F32>::T v[3], float sw)
{
    using T = typename Pix<F32>::T;
    const int sh = 8 - level, ts = kElePixels >> level;
    const uint64_t ent = table[(y >> sh) * tiles_x + (x >> sh)];
    if (!ent || cell_culled(ent, x, y, ts, sh)) return;
    const uint64_t slot = ent & ~kEntFlags;
    const int loc = (y & (ts - 1)) * ts + (x & (ts - 1));
    float PF_GLOBAL* dw = (float PF_GLOBAL*)(slot + w_off) + loc;
    if (!(ent & 1) && !(sw >= *dw)) return;
    T PF_GLOBAL* dl = (T PF_GLOBAL*)(slot + lap_off) + loc * 3;
    dl[0] = v[0]; dl[1] = v[1]; dl[2] = v[2];
    *dw = sw;
}

// workgroup barrier that waits for this wave's LDS traffic only: global loads issued earlier
// (the D-stage weight prefetch) stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// stage A for levels >= 1: GW_i block + halo from HBM into LDS.  Named pixels (not an array:
// arrays of this struct end up in scratch), four loads in flight per thread before their LDS stores.
constexpr int LAWH = (LAW + 1) / 2;                     // columns of one parity in the split layout

// SPLIT: A is stored as [row][column parity][column / 2] (see level3_block)
template <bool F32, int LAH, int LNT, bool SPLIT = false, int PITCH = 2 * LAWH>
__device__ __forceinline__ void stage_from_hbm(PxT<F32>* __restrict__ Aflat, const PxT<F32>* __restrict__ gin_,
                                               int ax0, int ay0, int rows, int cols, int tid, int begin = 0, bool gw8 = false)
{
    using Px = PxT<F32>;
    // gw8 (experiments library, PF_ABLATE bit 4096, int16, TIMING ONLY -- wrong weights): GW records of 8 bytes at an 8-byte pitch, the bound of
    // what a smaller int16 GW record (10 bytes in two planes) could gain (profiles/r06_ab.md)
    struct Gin {
        const PxT<F32>* p; bool g8;
        __device__ __forceinline__ Px operator[](long a) const {
            if constexpr (!F32 && kExp) {
                if (g8) { typedef uint32_t u2g __attribute__((ext_vector_type(2))); const u2g v = ((const u2g*)p)[a]; Px o; __builtin_memcpy(&o, &v, 8); o.w = 0.5f; return o; }
            }
            return p[a];
        }
    };
    const Gin gin{ gin_, gw8 };
    auto gaddr = [&](int idx) {
        idx = idx < LAH * LAW ? idx : tid;
        const int r = idx / LAW, c = idx - r * LAW;
        const int y = border_reflect101(ay0 + r, rows), x = border_reflect101(ax0 + c, cols);
        return (long)y * cols + x;
    };
    auto laddr = [&](int idx) {
        if (!SPLIT) return idx;
        const int r = idx / LAW, c = idx - r * LAW;
        return r * PITCH + (c & 1) * LAWH + (c >> 1);
    };
    if constexpr (SPLIT && LAH * LAW <= 6 * LNT) {
        // the whole tile in one round: six loads per thread in flight before the first LDS store.  These workgroups do
        // nothing but wait for GW_i (written by the previous launch, served from the Infinity Cache or HBM): two
        // dependent rounds of four loads held a workgroup slot for ~9000 cycles (tools/stamp_phases.py).
        const int i0 = tid, i1 = i0 + LNT, i2 = i1 + LNT, i3 = i2 + LNT, i4 = i3 + LNT, i5 = i4 + LNT;
        const Px t0 = gin[gaddr(i0)], t1 = gin[gaddr(i1)], t2 = gin[gaddr(i2)], t3 = gin[gaddr(i3)], t4 = gin[gaddr(i4)], t5 = gin[gaddr(i5)];
        Aflat[laddr(i0)] = t0; Aflat[laddr(i1)] = t1; Aflat[laddr(i2)] = t2; Aflat[laddr(i3)] = t3;
        if (i4 < LAH * LAW) Aflat[laddr(i4)] = t4;
        if (i5 < LAH * LAW) Aflat[laddr(i5)] = t5;
        static_assert(LAH * LAW > 4 * LNT, "i3 is always inside the tile");
        return;
    }
#pragma unroll
    for (int base = 0; base < LAH * LAW; base += 4 * LNT) {
        if (base + begin >= LAH * LAW) break;
        const int i0 = base + begin + tid, i1 = i0 + LNT, i2 = i1 + LNT, i3 = i2 + LNT;
        const Px t0 = gin[gaddr(i0)], t1 = gin[gaddr(i1)], t2 = gin[gaddr(i2)], t3 = gin[gaddr(i3)];
        if (i0 < LAH * LAW) Aflat[laddr(i0)] = t0;
        if (i1 < LAH * LAW) Aflat[laddr(i1)] = t1;
        if (i2 < LAH * LAW) Aflat[laddr(i2)] = t2;
        if (i3 < LAH * LAW) Aflat[laddr(i3)] = t3;
    }
}

template <bool F32, bool FROM_WARP, int LBH, int LNT, int LS>
__global__ __launch_bounds__(LNT) void k_level(LevelOffsets lay, LevelArgs g, FusedWarp wa, const uint8_t* __restrict__ src,
                                                const PxT<F32>* __restrict__ gw_in, PxT<F32>* __restrict__ gw_out,
                                                const uint64_t* __restrict__ table)
{
    using T = typename Pix<F32>::T; using WT = typename Pix<F32>::WT;
    using Px = PxT<F32>; using Hx = HxT<F32>;
    constexpr int LAH = LBH + 7, LQH = LBH / 2 + 2;
    static_assert((LBH * LBW) % LNT == 0, "D stage: whole passes");
    static_assert(LS == 1 || 7 * LAW <= LNT, "strip shift: one overlap pixel per thread");
    __shared__ Px A[LAH][LAW];
    __shared__ Hx Ht[LAH][LQW];
    __shared__ Px Bt[LQH][LQW];

    // XCD-aware block order: the 8 XCDs take workgroups round-robin, so give each XCD a
    // contiguous run of blocks (neighbouring blocks share halo pixels in that XCD's L2)
    const int nblk = g.nbx * g.nby;
    int b = blockIdx.x;
    {
        const int per = nblk >> 3;
        if (per > 0 && b < per * 8) b = (b & 7) * per + (b >> 3);
    }
    const int bx = b % g.nbx, by = b / g.nbx;
    const int x0 = g.cx0 + bx * LBW, ax0 = x0 - 4, bx0 = (x0 >> 1) - 1;
    const int nrows = g.rows >> 1, ncols = g.cols >> 1;         // level i+1 extent
    const int tid = threadIdx.x;
    // Rolling vertical strip: the workgroup walks LS consecutive LBH-row blocks; the 7 halo rows two
    // consecutive blocks share stay in LDS (shifted to the top), so only LBH new rows are staged per step.
#pragma unroll 1
    for (int st = 0; st < LS; st++) {
    const int y0 = g.cy0 + (by * LS + st) * LBH;
    if (y0 >= g.cy1 || y0 >= g.rows) break;
    const int ay0 = y0 - 4, by0 = (y0 >> 1) - 1;
    const int rb = st == 0 ? 0 : 7;                             // first staged row of this step

    // tile-table entries of this thread's D pixels: loaded now, used after stage A
    constexpr int ND = LBH * LBW / LNT;
    const int sh = 8 - g.level, ts = kElePixels >> g.level;
    uint64_t ents[ND];
    float dwv[ND];
#pragma unroll
    for (int it = 0; it < ND; it++) {
        const int idx = tid + it * LNT, y = y0 + (idx >> 6), x = x0 + (idx & 63);
        ents[it] = (x < g.cols && y < g.rows) ? table[(y >> sh) * g.tiles_x + (x >> sh)] : 0;
    }

    // ---- A
    if constexpr (FROM_WARP) {
        // blocks whose halo lies inside the canvas (all but the rim) skip the REFLECT_101 mapping
        const bool inner = ax0 >= 0 && ax0 + LAW <= g.cols && ay0 >= 0 && ay0 + LAH <= g.rows;
        int r = rb + tid / LAW, c = tid % LAW;
        for (int idx = rb * LAW + tid; idx < LAH * LAW; idx += LNT) {
            int y = ay0 + r, x = ax0 + c;
            if (!inner) { y = border_reflect101(y, g.rows); x = border_reflect101(x, g.cols); }
            if (kExp && (g.ablate & 1)) { Px z{}; z.w = (float)(x + y); (&A[0][0])[idx] = z; } else
            (&A[0][0])[idx] = warp_pixel<F32>(src, wa, warp_col(wa, x), y);
            c += LNT % LAW; r += LNT / LAW;
            if (c >= LAW) { c -= LAW; r++; }
        }
    } else {
        stage_from_hbm<F32, LAH, LNT>(&A[0][0], gw_in, ax0, ay0, g.rows, g.cols, tid, rb * LAW);
    }
    lds_barrier();
    if (kExp && (g.ablate & 2)) return;
    // stored weights of the D pixels: in flight during H / B / U (the barriers below wait for LDS only)
#pragma unroll
    for (int it = 0; it < ND; it++) {
        uint64_t ent = ents[it];
        // when this block's 64-pixel row segments lie inside one tile (always on an unsharded canvas; a shard's
        // halo-extended region may start 8 px left of a tile edge) make the slot address wave-uniform
        if ((x0 >> sh) == ((x0 + LBW - 1) >> sh))
            ent = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ent >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ent);
        ents[it] = ent;
        dwv[it] = -1.f;                                   // fresh tile: every weight (>= 0) wins
        if (ent && !(ent & 1)) {
            const int idx = tid + it * LNT, y = y0 + (idx >> 6), x = x0 + (idx & 63);
            dwv[it] = ((const float PF_GLOBAL*)((ent & ~(uint64_t)1) + lay.w_off))[(y & (ts - 1)) * ts + (x & (ts - 1))];
        }
    }
    // ---- H
    for (int idx = tid; idx < LAH * LQW; idx += LNT) {
        const int r = idx / LQW, q = idx - r * LQW;
        const Px a0 = A[r][2 * q], a1 = A[r][2 * q + 1], a2 = A[r][2 * q + 2], a3 = A[r][2 * q + 3], a4 = A[r][2 * q + 4];
        Hx h;
#pragma unroll
        for (int k = 0; k < 3; k++) h.c[k] = (WT)a2.c[k] * 6 + ((WT)a1.c[k] + (WT)a3.c[k]) * 4 + (WT)a0.c[k] + (WT)a4.c[k];
        h.w = a2.w * 6 + (a1.w + a3.w) * 4 + a0.w + a4.w;
        Ht[r][q] = h;
    }
    lds_barrier();
    // ---- B
    const int vec_img = (ncols * 3 / 8) * 8, vec_w = (ncols / 8) * 8;     // PyrDownVec_32f coverage
    for (int idx = tid; idx < LQH * LQW; idx += LNT) {
        const int p = idx / LQW, q = idx - p * LQW;
        const int Y = by0 + p, X = bx0 + q;
        if (X < 0 || X >= ncols || Y < 0 || Y >= nrows) continue;
        const Hx r0 = Ht[2 * p][q], r1 = Ht[2 * p + 1][q], r2 = Ht[2 * p + 2][q], r3 = Ht[2 * p + 3][q], r4 = Ht[2 * p + 4][q];
        Px o;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if constexpr (F32) {
                if (X * 3 + k < vec_img) {
                    float a = r0.c[k] + r4.c[k];
                    const float bb = (r1.c[k] + r3.c[k]) + r2.c[k];
                    a = a + (r2.c[k] + r2.c[k]);
                    o.c[k] = (a + bb * 4.f) * (1.f / 256);
                } else
                    o.c[k] = cast_down(r2.c[k] * 6 + (r1.c[k] + r3.c[k]) * 4 + r0.c[k] + r4.c[k]);
            } else
                o.c[k] = cast_down(r2.c[k] * 6 + (r1.c[k] + r3.c[k]) * 4 + r0.c[k] + r4.c[k]);
        }
        if (X < vec_w) {
            float a = r0.w + r4.w;
            const float bb = (r1.w + r3.w) + r2.w;
            a = a + (r2.w + r2.w);
            o.w = (a + bb * 4.f) * (1.f / 256);
        } else
            o.w = (r2.w * 6 + (r1.w + r3.w) * 4 + r0.w + r4.w) * (1.f / 256);
        if constexpr (!F32) o.pad = 0;
        Bt[p][q] = o;
        if (p >= 1 && p < LQH - 1 && q >= 1 && q < LQW - 1) {      // this block's own part of level i+1
            if (g.write_next) gw_out[(long)Y * ncols + X] = o;
            if (g.top_select) select_store<F32>(lay.top_lap_off, lay.top_w_off, g.level + 1, table, g.tiles_x, X, Y, o.c, o.w);
        }
    }
    lds_barrier();
    if (kExp && (g.ablate & 4)) return;
    // ---- U: horizontal pyrUp of the B rows into LDS (re-uses Ht, dead after B)
    Hx (*U)[LBW] = reinterpret_cast<Hx (*)[LBW]>(&Ht[0][0]);
    static_assert(sizeof(Hx) * LQH * LBW <= sizeof(Hx) * LAH * LQW, "U must fit in Ht");
    const bool edge = (x0 == 0) || (x0 + LBW >= g.cols) || (ncols == 1);
    for (int idx = tid; idx < LQH * LBW; idx += LNT) {
        const int p = idx >> 6, lx = idx & 63;
        const int x = x0 + lx, sx = x >> 1, j = (lx >> 1) + 1;
        Hx u;
        if (!edge) {
            const Px a = Bt[p][j - 1], b = Bt[p][j], c = Bt[p][j + 1];
#pragma unroll
            for (int k = 0; k < 3; k++) u.c[k] = up_h_val<WT, false>((WT)a.c[k], (WT)b.c[k], (WT)c.c[k], x & 1, false, false, false);
        } else {
            const bool one = ncols == 1, le = sx == 0, re = sx == ncols - 1;
            // int16: reflected (left) / replicated (right) neighbours; fp32: values unused where le/re select
            const Px a = Bt[p][(le && !F32) ? (one ? j : j + 1) : j - 1], b = Bt[p][j], c = Bt[p][(re && !F32) ? j : j + 1];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if constexpr (F32) u.c[k] = up_h_val<WT, true>(a.c[k], b.c[k], c.c[k], x & 1, le, re, one);
                else u.c[k] = up_h_val<WT, false>((WT)a.c[k], (WT)b.c[k], (WT)c.c[k], x & 1, false, false, false);
            }
        }
        u.w = 0.f;
        U[p][lx] = u;
    }
    lds_barrier();
    // ---- D: vertical pyrUp, Laplacian, max-weight select (one wave = one 64-pixel row)
#pragma unroll
    for (int it = 0; it < ND; it++) {
        const uint64_t ent = ents[it];
        if (!ent) continue;
        const int idx = tid + it * LNT, ly = idx >> 6, lx = idx & 63;
        const int y = y0 + ly, x = x0 + lx;
        const Px gpx = A[ly + 4][lx + 4];
        if (!(gpx.w >= dwv[it])) continue;
        const uint64_t slot = ent & ~(uint64_t)1;
        const int loc = (y & (ts - 1)) * ts + (x & (ts - 1));
        const int sy = y >> 1;
        int syn = sy + 1; if (syn >= nrows) syn = nrows - 1;
        int syp = sy - 1; if (syp < 0) syp = nrows > 1 ? 1 : 0;
        const Hx u1 = U[sy - by0][lx], u2 = U[syn - by0][lx];
        T out[3];
        if (y & 1) {
#pragma unroll
            for (int k = 0; k < 3; k++) out[k] = sat_sub(gpx.c[k], cast_up((u1.c[k] + u2.c[k]) * 4));
        } else {
            const Hx u0 = U[syp - by0][lx];
#pragma unroll
            for (int k = 0; k < 3; k++) out[k] = sat_sub(gpx.c[k], cast_up(u0.c[k] + u1.c[k] * 6 + u2.c[k]));
        }
        T PF_GLOBAL* dl = (T PF_GLOBAL*)(slot + lay.lap_off) + loc * 3;
        dl[0] = out[0]; dl[1] = out[1]; dl[2] = out[2];
        ((float PF_GLOBAL*)(slot + lay.w_off))[loc] = gpx.w;
    }
    if (st + 1 < LS) {
        lds_barrier();                                           // every read of A / U of this step is done
        Px keep;
        const bool mv = tid < 7 * LAW;
        if (mv) keep = (&A[0][0])[LBH * LAW + tid];             // rows LBH .. LBH+6 are the next step's rows 0 .. 6
        lds_barrier();
        if (mv) (&A[0][0])[tid] = keep;
    }
    }   // strip step
}

// --------------------------------------------------------------------------
// k_level3: the same computation as k_level in three stages and two barriers
//   A  as above
//   B  G_{i+1} / W_{i+1} straight from A: a thread forms the 7 horizontal 5-tap sums that
//      two vertically adjacent outputs share, then both vertical sums (no H tile in LDS)
//   D  one thread = one 2x2 output quad: the 3x3 neighbourhood of G_{i+1} is read once and
//      serves the four pyrUp parities; one tile-table entry and two 8-byte weight loads per
//      thread, prefetched after A
// LDS: A + B only (54 KB fp32 / 40.6 KB int16).
// ILP (stage A of a level-0 block): 0 = every row's coordinates, weight and loads first, every bilinear sum last, with the fractions and the
// weight parked in the pixel's LDS slot meanwhile (blocks that map strictly inside the frame; the fp32 product form since round 5);
// 2 / 3 = fetch, fetch(, fetch) | finish, finish(, finish) per step (the int16 product form: 2; every block that does not map inside).
// diagnostic build only (PF_STAMP=1, tools/stamp_phases.py): s_memtime at the phase boundaries of a workgroup, written by
// its first lane to a buffer nothing else reads.  The stamped kernel is a separate instantiation; the product kernel
// carries no stamp code.
__device__ __forceinline__ void phase_stamp(unsigned long long* st, int slot)
{
    if (st) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        if (threadIdx.x == 0) st[slot] = t;
    }
}

// diagnostic build only (PF_STAMP=1, tools/count_wins.py): per level, pixels stage D looked at / pixels that won the max-weight select --
// the USEFUL tile bytes of a launch are wins x (12 + 4) (fp32), whatever WRITE_SIZE says about sectors
__device__ unsigned long long g_select_seen[kMaxLevels], g_select_won[kMaxLevels];

// bytes of LDS a PATCH workgroup has behind A for the source patch (Bt lives there after stage A): two workgroups per CU
constexpr int kPatchBytes = 81920 - 44928;

template <bool F32, int LBH, int LNT, bool STAMP = false, int ILP = 2, bool PATCH = false, bool WA = false, class LO = LevelOffsets, class GA = LevelArgs>
__device__ __forceinline__ void level3_block(const bool FROM_WARP, const LO& lay, const GA& g, const FusedWarp& wa,
                                             const uint8_t* __restrict__ src, const PxT<F32>* __restrict__ gw_in,
                                             PxT<F32>* __restrict__ gw_out, const uint64_t* __restrict__ table_generic, const int b,
                                             unsigned long long* stamps = nullptr, const uint64_t* tab0 = nullptr)
{
    // tab0 (level-0 job of a pipelined launch): the same table inside the kernel arguments -- the copy in device memory is
    // written by this very launch, for the later ones.  Only the quad's entry is read from it (the top-level select of
    // stage B belongs to the last level job, which is never the job of the newest frame when there are two levels or more).
    // the table lies in device memory or in the kernel-argument segment: global memory either way (not a FLAT access)
    const uint64_t PF_GLOBAL* __restrict__ table = (const uint64_t PF_GLOBAL*)table_generic;
    if (STAMP) phase_stamp(stamps, 0);
    // stamped build only: stamp 5 ("stage D done, stores drained") when the function is left, whichever return ends it
    // (the guard lives at function scope: its destructor runs after stage D, not at the end of an inner block)
    struct StampExit { unsigned long long* st; __device__ StampExit(unsigned long long* p) : st(p) {} __device__ ~StampExit() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); phase_stamp(st, 5); } };
    struct NoExit { __device__ NoExit(unsigned long long*) {} };
    typename std::conditional<STAMP, StampExit, NoExit>::type at_exit(stamps);
    (void)at_exit;
    using T = typename Pix<F32>::T; using WT = typename Pix<F32>::WT;
    using Px = PxT<F32>;
    constexpr int LAH = LBH + 7, LQH = LBH / 2 + 2;
    static_assert(LNT >= 32 * (LBH / 2), "one thread per 2x2 output quad (threads beyond that help in stages A and B only)");
    struct Bx { T c[F32 ? 3 : 4]; };                            // stage D needs G_{i+1} only, not W_{i+1}
    // A is split by column parity, [row][parity][column / 2]: the pyrDown taps of neighbouring threads (2q .. 2q+4)
    // and the 2x2 quads of stage D are then 16 bytes apart per lane instead of 32 -- no LDS bank conflicts
    // (LBH == 24, experiments library: the odd columns follow the 36 even ones without the padding column -- 71 pixels per row, so that a 64 x 24
    // fp32 block takes 40 928 bytes with Bt and FOUR workgroups share a CU's 160 KB)
    constexpr int kPitch = LBH == 24 ? LAW : 2 * LAWH;
    __shared__ Px A[LAH][kPitch];
    // Bt (stage B -> D) and, in a PATCH kernel, the source patch of stage A share the LDS behind A: the patch is dead at
    // the barrier that ends stage A, Bt is first written after it
    constexpr int kTail = PATCH ? kPatchBytes : (int)(sizeof(Bx) * LQH * LQW);
    static_assert(kTail >= (int)(sizeof(Bx) * LQH * LQW), "Bt must fit");
    __shared__ __attribute__((aligned(16))) unsigned char tail[kTail];
    Bx (*Bt)[LQW] = reinterpret_cast<Bx (*)[LQW]>(tail);
    auto Aat = [&](int r, int c) -> Px& { return A[r][(c & 1) * LAWH + (c >> 1)]; };
    // LDS is allocated in 1280-byte granules on gfx950: fp32 must stay under 42 granules for three
    // workgroups per CU, int16 under 32 for four
    static_assert(LBH != 32 || PATCH || sizeof(A) + sizeof(tail) <= (F32 ? 42 : 32) * 1280, "LDS budget (64x32 blocks)");
    static_assert(LBH != 24 || sizeof(A) + sizeof(tail) <= 32 * 1280, "LDS budget (64x24 blocks: four workgroups per CU for both pyramid types)");
    static_assert(LBH != 32 || !PATCH || sizeof(A) + sizeof(tail) <= 64 * 1280, "LDS budget (64x32 blocks with a source patch: two per CU)");
    static_assert(LBH != 64 || sizeof(A) + sizeof(tail) <= (F32 ? 128 : 64) * 1280, "LDS budget (64x64 blocks of 1024 threads: int16 two per CU, fp32 one)");

    int bx, by; block_xy(g, b, bx, by);
    const int x0 = g.cx0 + bx * LBW, y0 = g.cy0 + by * LBH;
    const int ax0 = x0 - 4, ay0 = y0 - 4;
    const int nrows = g.rows >> 1, ncols = g.cols >> 1;         // level i+1 extent
    const int bx0 = (x0 >> 1) - 1, by0 = (y0 >> 1) - 1;
    const int tid = threadIdx.x;
    const int sh = 8 - g.level, ts = kElePixels >> g.level;
    if (kExp && (g.ablate & 512) && !FROM_WARP) __builtin_amdgcn_s_setprio(2);      // A/B: upper-level (latency-bound) workgroups first

    // this thread's output quad and its tile-table entry (a quad never straddles tiles)
    const int qx = tid & 31, qy = tid >> 5;
    const int dx0 = x0 + 2 * qx, dy0 = y0 + 2 * qy;
    uint64_t ent = 0;
    if (tid < 32 * (LBH / 2) && dx0 < g.cols && dy0 < g.rows) {
        const int ti = (dy0 >> sh) * g.tiles_x + (dx0 >> sh);
        if (tab0) ent = ((const uint64_t PF_GLOBAL*)tab0)[ti];
        else ent = table[ti];
        if (sh >= 3) { if (cell_culled(ent, dx0, dy0, ts, sh)) ent = 0; }       // cells of two pixels or more: the quad lies in one
        else {                                                                   // single-pixel cells (a 4-pixel tile): per pixel, below
            bool all = true;
#pragma unroll
            for (int k = 0; k < 4; k++) all = all && cell_culled(ent, dx0 + (k & 1), dy0 + (k >> 1), ts, sh);
            if (all) ent = 0;
        }
        ent &= ~kEntCells;                                                       // from here on: slot address | fresh bit
    }

    // ---- A
    // A/B (PF_ABLATE bits 1024 / 2048): static priority for the younger / the older half of the workgroup's waves during stage A (a SIMD serves
    // the older of its two waves of a workgroup first, and wave 0 waits ~3000 cycles at the barrier behind stage A for the younger ones)
    if (kExp && FROM_WARP && (g.ablate & 1024) && tid >= LNT / 2) __builtin_amdgcn_s_setprio(1);
    if (kExp && FROM_WARP && (g.ablate & 2048) && tid < LNT / 2) __builtin_amdgcn_s_setprio(1);
    if (FROM_WARP) {
        // a thread keeps one column of A and walks down it LNT / LAW rows at a time: the column terms of the
        // coordinate are formed once
        const bool inner = ax0 >= 0 && ax0 + LAW <= g.cols && ay0 >= 0 && ay0 + LAH <= g.rows;
        constexpr int RS = LNT / LAW;
        const int r0 = tid / LAW, c = tid - r0 * LAW;
        // halo coordinates lie in [-4, len + 3): one reflection unless the level is only a few pixels wide
        const bool near101 = g.rows >= 8 && g.cols >= 8;
        auto col_of = [&](int cc) {
            int x = ax0 + cc;
            if (!inner) x = near101 ? border_reflect101_near(x, g.cols) : border_reflect101(x, g.cols);
            return x;
        };
        auto row_of = [&](int r) {
            int y = ay0 + r;
            if (!inner) y = near101 ? border_reflect101_near(y, g.rows) : border_reflect101(y, g.rows);
            return y;
        };
        if constexpr (PATCH) {
            // ---- the block's source patch: frame bytes and weight plane around the image of the block's centre -> LDS
            // (placement only needs to be roughly right: a pixel whose taps fall outside takes the global-memory path)
            const float fxc = (float)(ax0 + LAW / 2), fyc = (float)(ay0 + LAH / 2);
            const float Wc = wa.Mf[6] * fxc + wa.Mf[7] * fyc + wa.Mf[8], iw = __builtin_amdgcn_rcpf(Wc);
            const float pcx = (wa.Mf[0] * fxc + wa.Mf[1] * fyc + wa.Mf[2]) * iw, pcy = (wa.Mf[3] * fxc + wa.Mf[4] * fyc + wa.Mf[5]) * iw;
            // clamp before the conversion: a wild centre gives an empty patch, not undefined behaviour
            const int icx = (int)__builtin_floorf(fminf(fmaxf(pcx, -1.0e6f), 1.0e6f)), icy = (int)__builtin_floorf(fminf(fmaxf(pcy, -1.0e6f), 1.0e6f));
            int px0 = __builtin_amdgcn_readfirstlane(icx) - wa.phx, py0 = __builtin_amdgcn_readfirstlane(icy) - wa.phy;
            int px1 = px0 + 2 * wa.phx + 2, py1 = py0 + 2 * wa.phy + 2;
            px0 = px0 > 0 ? px0 : 0; py0 = py0 > 0 ? py0 : 0;                   // clipped to the frame: inside it no tap is reflected
            px1 = px1 < wa.scols ? px1 : wa.scols; py1 = py1 < wa.srows ? py1 : wa.srows;
            const int pw = px1 - px0, ph = py1 - py0;                            // <= 0: no pixel of this block reads inside the frame
            const int cn = wa.cn, pitch_i = wa.pitch_i, pitch_w = wa.pitch_w;
            const int sb = (px0 * cn) & ~15, sbo = px0 * cn - sb;              // row bytes are staged from a 16-byte boundary of the row
            const int wb = (px0 * 4) & ~15, swo = px0 * 4 - wb;
            unsigned char* const pimg = tail;
            unsigned char* const pwgt = tail + (2 * wa.phy + 2) * pitch_i;
            if (pw > 1 && ph > 1) {
                // one wave = one patch row per round: lanes [0, nci) the frame chunks, [nci, nci + ncw) the weight chunks
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(1)));
                const int lane = tid & 63, wv = tid >> 6, nci = pitch_i >> 4;
                const bool isw = lane >= nci;
                const int ch = isw ? lane - nci : lane;
                // only the chunks that hold bytes a tap can read: [sbo, sbo + (pw-2)*cn + 8) of a frame row, [swo, swo + pw*4) of a weight row
                const bool on = ch * 16 < (isw ? swo + pw * 4 : sbo + (pw - 2) * cn + 8);
                constexpr int NW = LNT / 64, UNR = 4;
                for (int rb = wv; rb < ph; rb += NW * UNR) {
                    u4 v[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        const int r = rb + u * NW;
                        v[u] = u4{ 0, 0, 0, 0 };
                        if (on && r < ph) {
                            if (isw) {
                                // the weight plane carries 64 bytes of slack behind its last row (fusion_map.cpp)
                                const char* rowp = (const char*)(wa.wmap + (size_t)(py0 + r) * wa.scols) + wb;
                                v[u] = *(const u4u*)(rowp + ch * 16);
                            } else {
                                const long off = (long)(py0 + r) * wa.sstep + sb + ch * 16;
                                if (off + 16 <= wa.total) v[u] = *(const u4u*)(src + off);
                                else {                                            // the last bytes of the frame: never read past them
                                    uint32_t w4[4] = { 0, 0, 0, 0 };
                                    for (int k = 0; k < 16; k++) if (off + k < wa.total) w4[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
                                    v[u] = u4{ w4[0], w4[1], w4[2], w4[3] };
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        const int r = rb + u * NW;
                        if (on && r < ph) *(u4*)((isw ? pwgt + r * pitch_w : pimg + r * pitch_i) + ch * 16) = v[u];
                    }
                }
            }
            __syncthreads();
            if (r0 < RS) {
                const int x = col_of(c);
                const WarpCol col = warp_col(wa, x);
                const uint32_t hisel = cn == 3 ? 0x06050403u : 0x07060504u;
                typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(1)));
                struct Cd { int X, Y, Xn, Yn; bool ok; };
                // a pixel takes its taps from the patch when (ux, ux+1) x (uy, uy+1) lie inside it; an empty patch takes none
                const unsigned okw = pw > 1 && ph > 1 ? (unsigned)(pw - 1) : 0u, okh = pw > 1 && ph > 1 ? (unsigned)(ph - 1) : 0u;
                // coordinates exactly as warp_fetch_pre's common case (the host checked that every pixel of this canvas is
                // "tame": patch_plan / tame_canvas), then: do both taps' rows and columns lie inside the staged patch?
                auto coords = [&](int y) {
                    const double X0 = col.m0xb + wa.M[1] * y + wa.M[2], Y0 = col.m3xb + wa.M[4] * y + wa.M[5], W0 = col.m6xb + wa.M[7] * y + wa.M[8];
                    const double W = W0 + col.m6x1, xn = X0 + col.m0x1, yn = Y0 + col.m3x1;
                    const double Wn = rcp_mid_range(W);
                    const double pxn = xn * Wn, pyn = yn * Wn;
                    constexpr double kMagic = 6755399441055744.0;
                    Cd o;
                    o.Xn = (int)(uint32_t)(unsigned long long)__double_as_longlong(pxn + kMagic);
                    o.Yn = (int)(uint32_t)(unsigned long long)__double_as_longlong(pyn + kMagic);
                    o.X  = (int)(uint32_t)(unsigned long long)__double_as_longlong(pxn * 32. + kMagic);
                    o.Y  = (int)(uint32_t)(unsigned long long)__double_as_longlong(pyn * 32. + kMagic);
                    o.ok = (unsigned)((o.X >> 5) - px0) < okw && (unsigned)((o.Y >> 5) - py0) < okh;
                    return o;
                };
                auto fast = [&](const Cd& q) {
                    // taps (ux, ux+1) x (uy, uy+1) and the nearest pixel (Xn, Yn) -- one of those four -- are inside the frame
                    const int la = __mul24((q.Y >> 5) - py0, pitch_i) + __mul24(q.X >> 5, cn) - sb;
                    const u2 b0 = *(const u2*)(pimg + la), b1 = *(const u2*)(pimg + la + pitch_i);
                    const float wv_ = *(const float*)(pwgt + __mul24(q.Yn - py0, pitch_w) + q.Xn * 4 - wb);
                    PxT<F32> o;
                    o.w = wv_;
                    const float fx = (float)(q.X & 31) * (1.f / 32), fy = (float)(q.Y & 31) * (1.f / 32);
                    const float c0 = (1.f - fy) * (1.f - fx), c1 = (1.f - fy) * fx, c2 = fy * (1.f - fx), c3 = fy * fx;
                    const uint32_t h0 = __builtin_amdgcn_perm(b0.y, b0.x, hisel), h1 = __builtin_amdgcn_perm(b1.y, b1.x, hisel);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        float v0 = (float)((b0.x >> (8 * k)) & 0xff), v1 = (float)((h0 >> (8 * k)) & 0xff);
                        float v2 = (float)((b1.x >> (8 * k)) & 0xff), v3 = (float)((h1 >> (8 * k)) & 0xff);
                        if constexpr (F32) {
                            const float sc = (float)(1. / 255.);
                            v0 = v0 * sc; v1 = v1 * sc; v2 = v2 * sc; v3 = v3 * sc;
                            o.c[k] = v0 * c0 + v1 * c1 + v2 * c2 + v3 * c3;
                        } else {
                            // every partial sum is a multiple of 2^-10 below 2^8: exact in fp32 whatever the association, fused or not
                            const float tt = __builtin_fmaf(v3, c3, __builtin_fmaf(v2, c2, __builtin_fmaf(v1, c1, v0 * c0)));
                            o.c[k] = (short)sat_short(__float2int_rn(tt));
                        }
                    }
                    if constexpr (!F32) o.pad = 0;
                    return o;
                };
                for (int r = r0; r < LAH; r += 2 * RS) {
                    const int rb = r + RS;
                    const bool hasb = rb < LAH;
                    const int y = row_of(r), yb = row_of(hasb ? rb : r);
                    if (kExp && (g.ablate & 1)) { Px z{}; z.w = (float)(x + y); Aat(r, c) = z; if (hasb) Aat(rb, c) = z; continue; }
                    const Cd qa = coords(y), qb = coords(yb);
                    if (__builtin_amdgcn_ballot_w64(!(qa.ok && qb.ok)) == 0) {
                        Aat(r, c) = fast(qa);
                        const PxT<F32> pb = fast(qb);
                        if (hasb) Aat(rb, c) = pb;
                    } else {
                        Aat(r, c) = warp_pixel<F32>(src, wa, col, y);
                        if (hasb) Aat(rb, c) = warp_pixel<F32>(src, wa, col, yb);
                    }
                }
            }
        } else
        if (ILP == 0 && WA && inner && wa.plain && block_maps_inside(wa, ax0, ay0, LAW, LAH)) {
            // ---- the product's stage A for a block whose staged rectangle maps strictly inside the frame (nine in ten of the blocks that run
            // in the steady state): no pixel needs the range test, the border forms or the frame test, so the row loop is straight-line code.
            // A thread keeps one column and walks down it ONCE: per row the coordinates, the weight and the two row loads -- issued at once,
            // consumed last; what the bilinear sum needs later (the 1/32-px fractions and the weight) is PARKED in the pixel's own LDS slot,
            // which is idle until the finished pixel is written there (the same thread reads what it wrote: no barrier).  Then every row's
            // bilinear sum.  All 2 x 6 loads of a thread are in flight while it computes the coordinates of its later rows, and a wave
            // waits for memory once per block instead of once per step (rounds 2-4: fetch, fetch, fetch | finish, finish, finish per step;
            // the straight-line form of that loop was +22 % per wave but needed 94-123 VGPRs for the parked state, r02).
            // A/B (-DPF_ROWTAB=1): the terms of a pixel's coordinates that depend on its canvas row and 64-wide coordinate block only --
            // X0 = (M0 xb + M1 y) + M2, Y0, W0: nine fp64 operations per pixel -- from a table of 39 rows x 3 blocks x 3 terms that 351 threads
            // fill once per workgroup (in the LDS behind A, free until stage B), at the price of one more barrier and two LDS reads per pixel
            constexpr bool kRowTab = PF_ROWTAB != 0;
            double (*rowtab)[3][4] = reinterpret_cast<double (*)[3][4]>(tail);
            const int xb0 = ax0 & ~63;
            if constexpr (kRowTab) {
                static_assert(!kRowTab || sizeof(double) * LAH * 3 * 4 <= (size_t)kTail, "row table fits behind A");
                if (tid < LAH * 9) {
                    const int r = tid / 9, e = tid - r * 9, kx = e / 3, w = e - kx * 3;
                    const double xb = (double)(xb0 + 64 * kx), y = (double)(ay0 + r);
                    rowtab[r][kx][w] = wa.M[3 * w] * xb + wa.M[3 * w + 1] * y + wa.M[3 * w + 2];
                }
                lds_barrier();
            }
            if (r0 < RS) {
                constexpr int NR = (LAH + RS - 1) / RS;
                typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(1)));
                using p2 = typename std::conditional<F32, park8, park4>::type;       // an int16 slot is 12 bytes
                const WarpCol col = warp_col(wa, ax0 + c);
                const int cn = wa.cn, sstep = wa.sstep;
                // All NR rows of a thread at once (fp32), or in two halves (int16: twelve dwords of row data instead of twenty-four keep the kernel
                // inside the 64 VGPRs of four workgroups per CU)
                constexpr int CH = (F32 && LBH != 24) ? NR : (NR + 1) / 2;      // (64 x 24 fp32 blocks, experiments library: halves as well -- 64 VGPRs for a fourth workgroup per CU)
                u2 b0[CH], b1[CH];
                float wq[(CH + 1) / 2] = {};                       // gathered weights (A/B, see below): kept in registers until the finish
                // A thread's rows are CONSECUTIVE (r0 * NR + k).  (Experiments build, PF_SEED=1: the reciprocal of row k + 1 starts from that of
                // row k when the host found W to change slowly enough -- bit-exact, and no faster than one v_rcp_f64 per pixel: the row-to-row
                // dependency costs what the instruction saves.)  The last thread rows run past the tile: a wave none of whose lanes has row k skips it.
                const int rbase = r0 * NR;
                const bool seed = kExp && wa.seed_ok != 0;          // experiments build only (PF_SEED=1): measured, no gain (profiles/r05_ab.md section 4)
                double Wn = 0.0;
#pragma unroll
                for (int h = 0; h < NR; h += CH) {
#pragma unroll
                    for (int k = h; k < h + CH && k < NR; k++) {
                        const bool has = rbase + k < LAH;
                        if (k >= LAH - (RS - 1) * NR && __builtin_amdgcn_ballot_w64(has) == 0) continue;
                        const int r = has ? rbase + k : rbase;
                        const int y = ay0 + r;
                        double X0, Y0, W0;
                        if constexpr (kRowTab) {
                            const double* t = rowtab[r][(((ax0 + c) & ~63) - xb0) >> 6];
                            X0 = t[0]; Y0 = t[1]; W0 = t[2];
                        } else {
                            X0 = col.m0xb + wa.M[1] * y + wa.M[2]; Y0 = col.m3xb + wa.M[4] * y + wa.M[5]; W0 = col.m6xb + wa.M[7] * y + wa.M[8];
                        }
                        const double W = W0 + col.m6x1, xn = X0 + col.m0x1, yn = Y0 + col.m3x1;
                        // (a lane past the tile re-does its first row: its seed is then five rows off, and its result is not stored)
                        Wn = (k > 0 && seed) ? rcp_seeded(W, Wn) : rcp_mid_range(W);
                        const double pxn = xn * Wn, pyn = yn * Wn;
                        constexpr double kMagic = 6755399441055744.0;
                        const int Xn = (int)(uint32_t)(unsigned long long)__double_as_longlong(pxn + kMagic);
                        const int Yn = (int)(uint32_t)(unsigned long long)__double_as_longlong(pyn + kMagic);
                        // 32 * p is exact, so the fused form rounds once, exactly as p * 32 + magic does
                        const int X = (int)(uint32_t)(unsigned long long)__double_as_longlong(__builtin_fma(pxn, 32., kMagic));
                        const int Y = (int)(uint32_t)(unsigned long long)__double_as_longlong(__builtin_fma(pyn, 32., kMagic));
                        // A/B (build switch -DPF_HYBRID_W=1, tools/build_variant.sh): every other row GATHERS its weight from the plane instead of computing it -- the
                        // vector pipes (radial_weight: ~22 instructions) and the address unit (one more load) share the weights between them
                        constexpr bool kHyb = PF_HYBRID_W != 0; const bool gw_row = kHyb && (k & 1);
                        float wgt;
                        if (gw_row) wgt = *(const float*)((const char*)wa.wmap + ((uint32_t)(__mul24(Yn, wa.scols) + Xn) << 2));
                        else wgt = radial_weight(wa, Xn, Yn);
                        const uint32_t off0 = (uint32_t)(__mul24(Y >> 5, sstep) + __mul24(cn, X >> 5));
                        if (gw_row) wq[(k - h) >> 1] = wgt;
                        if (has) *reinterpret_cast<p2*>(&Aat(r, c)) = p2{ (uint32_t)((X & 31) | (Y & 31) << 5), __float_as_uint(gw_row ? 0.f : wgt) };
                        b0[k - h] = PF_LOAD_SRC((const u2*)(src + off0)); b1[k - h] = PF_LOAD_SRC((const u2*)(src + off0 + (uint32_t)sstep));
                    }
#pragma unroll
                    for (int k = h; k < h + CH && k < NR; k++) {
                        const bool has = rbase + k < LAH;
                        if (k >= LAH - (RS - 1) * NR && __builtin_amdgcn_ballot_w64(has) == 0) continue;
                        if (!has) continue;
                        const int r = rbase + k;
                        const p2 pk = *reinterpret_cast<const p2*>(&Aat(r, c));
                        constexpr bool kHyb = PF_HYBRID_W != 0; const bool gw_row = kHyb && (k & 1);
                        Aat(r, c) = warp_finish_fast<F32>(b0[k - h].x, b0[k - h].y, b1[k - h].x, b1[k - h].y, (int)(pk.x & 31u), (int)(pk.x >> 5), gw_row ? wq[(k - h) >> 1] : __uint_as_float(pk.y), cn);
                    }
                }
            }
        } else
        if (r0 < RS) {
            const int x = col_of(c);
            const WarpCol col = warp_col(wa, x);
            if constexpr (ILP == 3) {
                // three rows per step: nine loads of a thread in flight -- fewer waves are needed to keep the vector unit
                // fed while other workgroups of the CU sit in their memory-bound stages
                for (int r = r0; r < LAH; r += 3 * RS) {
                    const int rb = r + RS, rc = r + 2 * RS;
                    const bool hasb = rb < LAH, hasc = rc < LAH;
                    const int y = row_of(r), yb = row_of(hasb ? rb : r), yc = row_of(hasc ? rc : r);
                    if (kExp && (g.ablate & 1)) { Px z{}; z.w = (float)(x + y); Aat(r, c) = z; if (hasb) Aat(rb, c) = z; if (hasc) Aat(rc, c) = z; continue; }
                    const WarpTaps ta = warp_fetch<WA>(src, wa, col, y);
                    const WarpTaps tb = warp_fetch<WA>(src, wa, col, yb);
                    if (__builtin_amdgcn_ballot_w64(hasc) != 0) {
                        const WarpTaps tc = warp_fetch<WA>(src, wa, col, yc);
                        Aat(r, c) = warp_finish<F32>(ta, wa.cn);
                        if (hasb) Aat(rb, c) = warp_finish<F32>(tb, wa.cn);
                        if (hasc) Aat(rc, c) = warp_finish<F32>(tc, wa.cn);
                    } else {
                        Aat(r, c) = warp_finish<F32>(ta, wa.cn);
                        if (hasb) Aat(rb, c) = warp_finish<F32>(tb, wa.cn);
                    }
                }
            } else
            // two rows per step: the second pixel's coordinates and loads overlap the first one's loads
            for (int r = r0; r < LAH; r += 2 * RS) {
                const int rb = r + RS;
                const bool hasb = rb < LAH;
                int y = ay0 + r, yb = ay0 + rb;
                if (!inner) {
                    if (near101) { y = border_reflect101_near(y, g.rows); yb = border_reflect101_near(yb, g.rows); }
                    else { y = border_reflect101(y, g.rows); yb = border_reflect101(yb, g.rows); }
                }
                if (kExp && (g.ablate & 1)) { Px z{}; z.w = (float)(x + y); Aat(r, c) = z; if (hasb) Aat(rb, c) = z; continue; }
                const WarpTaps ta = warp_fetch<WA>(src, wa, col, y);
                if (__builtin_amdgcn_ballot_w64(hasb) != 0) {
                    const WarpTaps tb = warp_fetch<WA>(src, wa, col, hasb ? yb : y);
                    Aat(r, c) = warp_finish<F32>(ta, wa.cn);
                    if (hasb) Aat(rb, c) = warp_finish<F32>(tb, wa.cn);
                } else
                    Aat(r, c) = warp_finish<F32>(ta, wa.cn);
            }
        }
    } else {
        stage_from_hbm<F32, LAH, LNT, true, kPitch>(&A[0][0], gw_in, ax0, ay0, g.rows, g.cols, tid, 0, kExp && !F32 && (g.ablate & 4096));
    }
    if (STAMP) phase_stamp(stamps, 1);
    if (kExp && FROM_WARP && (g.ablate & (1024 | 2048))) __builtin_amdgcn_s_setprio(0);
    lds_barrier();
    if (STAMP) phase_stamp(stamps, 2);
    if (kExp && (g.ablate & 2)) return;
    // A never-taken exit (levels are >= 0).  With an exit here the compiler keeps stages B / D's address arithmetic below the barrier; without
    // one it hoists part of it above stage A, the fp32 kernel goes three VGPRs over its budget of 80, spills them to scratch, and the launch is
    // 3.6 % slower (same box, interleaved: profiles/r05_ab.md, "product against experiments build").  The experiments build has such an exit
    // anyway (PF_ABLATE, the line above), which is how the difference was found.
    if (!kExp && g.level < 0) return;
    if (kExp && (g.ablate & 256)) __builtin_amdgcn_s_setprio(2);           // A/B: the short memory-bound stages ahead of other workgroups' warp

    // stored weights of the quad: in flight during stage B
    float dwv[2][2] = { { -1.f, -1.f }, { -1.f, -1.f } };           // fresh tile: every weight (>= 0) wins
    const uint64_t slot = ent & ~kEntLow;
    const int loc0 = (dy0 & (ts - 1)) * ts + (dx0 & (ts - 1));
    if (ent && !(ent & 1)) {
        const float PF_GLOBAL* wp = (const float PF_GLOBAL*)(slot + lay.w_off) + loc0;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 v = PF_LOAD_W((const f2 PF_GLOBAL*)(wp + j * ts));
            dwv[j][0] = v.x; dwv[j][1] = v.y;
        }
    }
    if (sh < 3 && ent) {                                         // a quad across single-pixel cells (the entry read again: rare)
        const int ti = (dy0 >> sh) * g.tiles_x + (dx0 >> sh);
        const uint64_t e2 = tab0 ? ((const uint64_t PF_GLOBAL*)tab0)[ti] : table[ti];
#pragma unroll
        for (int k = 0; k < 4; k++)                               // no weight is >= NaN: those pixels never win
            if (cell_culled(e2, dx0 + (k & 1), dy0 + (k >> 1), ts, sh)) dwv[k >> 1][k & 1] = __builtin_nanf("");
    }

    // ---- B: two vertically adjacent outputs per thread
    const int vec_img = (ncols * 3 / 8) * 8, vec_w = (ncols / 8) * 8;     // PyrDownVec_32f coverage
    // the (LQH/2)*LQW work items go to the LAST threads of the workgroup: stage A gives the first waves one
    // more pixel per lane than the last ones, so this evens out the per-SIMD load
    static_assert((LQH / 2) * LQW <= LNT, "one B item per thread at most");
    for (int idx = tid - (LNT - (LQH / 2) * LQW); idx >= 0 && idx < (LQH / 2) * LQW; idx += LNT) {
        const int pp = idx / LQW, q = idx - pp * LQW;
        const int X = bx0 + q;
        if (X < 0 || X >= ncols) continue;
        // rows are consumed as they arrive (output e needs rows 2e .. 2e+4); each row's sums are pinned where they
        // are formed, otherwise the compiler sinks them into emit()'s range check and keeps the 35 raw taps live
        if constexpr (F32) {
            // packed fp32: a Px is two float2 halves (c0,c1) (c2,w); all four components take the same forms
            f2 hl[7], hh[7];
            const bool vec_all = X * 3 + 2 < vec_img && X < vec_w;          // every component in PyrDownVec_32f's range
            const bool uni = __builtin_amdgcn_ballot_w64(!vec_all) == 0;
            auto emit = [&](int e) {
                const int p = 2 * pp + e, Y = by0 + p;
                if (Y < 0 || Y >= nrows) return;
                Px o;
                if (uni) {
                    auto vsum = [](f2 r0, f2 r1, f2 r2, f2 r3, f2 r4) {
                        f2 a = r0 + r4;
                        const f2 bb = (r1 + r3) + r2;
                        a = a + (r2 + r2);
                        return (a + bb * 4.f) * (1.f / 256);
                    };
                    const f2 ol = vsum(hl[2 * e], hl[2 * e + 1], hl[2 * e + 2], hl[2 * e + 3], hl[2 * e + 4]);
                    const f2 oh = vsum(hh[2 * e], hh[2 * e + 1], hh[2 * e + 2], hh[2 * e + 3], hh[2 * e + 4]);
                    o.c[0] = ol.x; o.c[1] = ol.y; o.c[2] = oh.x; o.w = oh.y;
                } else {
                    float out[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        auto at = [&](int j) { return k < 2 ? hl[j][k] : hh[j][k - 2]; };
                        const float r0 = at(2 * e), r1 = at(2 * e + 1), r2 = at(2 * e + 2), r3 = at(2 * e + 3), r4 = at(2 * e + 4);
                        if (k < 3 ? X * 3 + k < vec_img : X < vec_w) {
                            float a = r0 + r4;
                            const float bb = (r1 + r3) + r2;
                            a = a + (r2 + r2);
                            out[k] = (a + bb * 4.f) * (1.f / 256);
                        } else
                            out[k] = (r2 * 6 + (r1 + r3) * 4 + r0 + r4) * (1.f / 256);
                    }
                    o.c[0] = out[0]; o.c[1] = out[1]; o.c[2] = out[2]; o.w = out[3];
                }
                {
                    Bx ob; ob.c[0] = o.c[0]; ob.c[1] = o.c[1]; ob.c[2] = o.c[2];
                    Bt[p][q] = ob;
                }
                if (p >= 1 && p < LQH - 1 && q >= 1 && q < LQW - 1) {      // this block's own part of level i+1
                    if (g.write_next) gw_out[(long)Y * ncols + X] = o;
                    if (g.top_select) select_store<F32>(lay.top_lap_off, lay.top_w_off, g.level + 1, table, g.tiles_x, X, Y, o.c, o.w);
                }
            };
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const f4* ev = (const f4*)&A[4 * pp + j][q]; const f4* od = (const f4*)&A[4 * pp + j][LAWH + q];
                const f4 a0 = ev[0], a1 = od[0], a2 = ev[1], a3 = od[1], a4 = ev[2];
                hl[j] = a2.xy * 6.f + (a1.xy + a3.xy) * 4.f + a0.xy + a4.xy;
                hh[j] = a2.zw * 6.f + (a1.zw + a3.zw) * 4.f + a0.zw + a4.zw;
                asm volatile("" : "+v"(hl[j]), "+v"(hh[j]) :: "memory");
                if (j == 4) emit(0);
                if (j == 6) emit(1);
            }
        } else {
            us2 h01[7], h2p[7]; float hw[7];
            auto emit = [&](int e) {
                const int p = 2 * pp + e, Y = by0 + p;
                if (Y < 0 || Y >= nrows) return;
                Px o;
                {
                    const us2 v01 = h01[2 * e + 2] * (us2)6 + (h01[2 * e + 1] + h01[2 * e + 3]) * (us2)4 + h01[2 * e] + h01[2 * e + 4];
                    const us2 v2p = h2p[2 * e + 2] * (us2)6 + (h2p[2 * e + 1] + h2p[2 * e + 3]) * (us2)4 + h2p[2 * e] + h2p[2 * e + 4];
                    const us2 o01 = (v01 + (us2)128) >> (us2)8, o2p = (v2p + (us2)128) >> (us2)8;
                    o.c[0] = (short)o01.x; o.c[1] = (short)o01.y; o.c[2] = (short)o2p.x;
                }
                {
                    const float r0 = hw[2 * e], r1 = hw[2 * e + 1], r2 = hw[2 * e + 2], r3 = hw[2 * e + 3], r4 = hw[2 * e + 4];
                    if (X < vec_w) {
                        float a = r0 + r4;
                        const float bb = (r1 + r3) + r2;
                        a = a + (r2 + r2);
                        o.w = (a + bb * 4.f) * (1.f / 256);
                    } else
                        o.w = (r2 * 6 + (r1 + r3) * 4 + r0 + r4) * (1.f / 256);
                }
                o.pad = 0;
                {
                    Bx ob; ob.c[0] = o.c[0]; ob.c[1] = o.c[1]; ob.c[2] = o.c[2]; ob.c[3] = 0;
                    Bt[p][q] = ob;
                }
                if (p >= 1 && p < LQH - 1 && q >= 1 && q < LQW - 1) {      // this block's own part of level i+1
                    if (kExp && (g.ablate & 4096)) {                          // timing only: 8-byte GW records (see stage_from_hbm)
                        typedef uint32_t u2g __attribute__((ext_vector_type(2)));
                        u2g v; __builtin_memcpy(&v, &o, 8);
                        if (g.write_next) ((u2g*)gw_out)[(long)Y * ncols + X] = v;
                    } else
                    if (g.write_next) gw_out[(long)Y * ncols + X] = o;
                    if (g.top_select) select_store<F32>(lay.top_lap_off, lay.top_w_off, g.level + 1, table, g.tiles_x, X, Y, o.c, o.w);
                }
            };
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const PxP* ev = reinterpret_cast<const PxP*>(&A[4 * pp + j][q]); const PxP* od = reinterpret_cast<const PxP*>(&A[4 * pp + j][LAWH + q]);
                const PxP a0 = ev[0], a1 = od[0], a2 = ev[1], a3 = od[1], a4 = ev[2];
                h01[j] = a2.c01 * (us2)6 + (a1.c01 + a3.c01) * (us2)4 + a0.c01 + a4.c01;
                h2p[j] = a2.c2p * (us2)6 + (a1.c2p + a3.c2p) * (us2)4 + a0.c2p + a4.c2p;
                hw[j] = a2.w * 6 + (a1.w + a3.w) * 4 + a0.w + a4.w;
                asm volatile("" : "+v"(h01[j]), "+v"(h2p[j]), "+v"(hw[j]) :: "memory");
                if (j == 4) emit(0);
                if (j == 6) emit(1);
            }
        }
    }
    if (STAMP) phase_stamp(stamps, 3);
    lds_barrier();
    if (STAMP) phase_stamp(stamps, 4);
    if (kExp && (g.ablate & 4)) return;

    // ---- D: 2x2 quad, Laplacian + max-weight select
    if (!ent) return;
    int ry = dy0;                                                // the quad's row in A from dy0 (live anyway), not from a register kept since the top
    asm volatile("" : "+v"(ry));
    ry -= y0;
    const Px g00 = A[ry + 4][qx + 2], g01 = A[ry + 4][LAWH + qx + 2];
    const Px g10 = A[ry + 5][qx + 2], g11 = A[ry + 5][LAWH + qx + 2];
    const bool in01 = dx0 + 1 < g.cols, in10 = dy0 + 1 < g.rows;
    const bool s00 = g00.w >= dwv[0][0], s01 = in01 && g01.w >= dwv[0][1];
    const bool s10 = in10 && g10.w >= dwv[1][0], s11 = in10 && in01 && g11.w >= dwv[1][1];
#ifdef PF_COUNT_WINS        // tools/count_wins.py (a build of its own: tools/build_variant.sh wins -DPF_EXPERIMENTS=1 -DPF_COUNT_WINS=1): two atomics per thread distort the stamps
    if constexpr (STAMP) {
        atomicAdd(&g_select_seen[g.level], (unsigned long long)(1 + (int)in01 + (int)in10 + (int)(in10 && in01)));
        atomicAdd(&g_select_won[g.level], (unsigned long long)((int)s00 + (int)s01 + (int)s10 + (int)s11));
    }
#endif
    if (!(s00 || s01 || s10 || s11)) return;
    const int sy = dy0 >> 1, sx = dx0 >> 1;
    int syn = sy + 1; if (syn >= nrows) syn = nrows - 1;
    int syp = sy - 1; if (syp < 0) syp = nrows > 1 ? 1 : 0;
    const bool one = ncols == 1, le = sx == 0, re = sx == ncols - 1;
    const bool edge = le || re;
    const int j = qx + 1;                                    // B column of sx
    // int16: edge forms are the interior form on reflected (left) / replicated (right) neighbours
    const int ja = (le && !F32) ? (one ? j : j + 1) : j - 1, jc = (re && !F32) ? j : j + 1;
    const int rows3[3] = { syp - by0, sy - by0, syn - by0 };
    T PF_GLOBAL* dl = (T PF_GLOBAL*)(slot + lay.lap_off) + loc0 * 3;
    float PF_GLOBAL* dw = (float PF_GLOBAL*)(slot + lay.w_off) + loc0;
    if constexpr (F32) {
        if (__builtin_amdgcn_ballot_w64(edge) == 0) {
            // interior columns, packed fp32: channels (0,1) as a float2, channel 2 scalar
            f2 heL[3], hoL[3]; float he2[3], ho2[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const float* pa = Bt[rows3[r]][j - 1].c; const float* pb = Bt[rows3[r]][j].c; const float* pc = Bt[rows3[r]][j + 1].c;
                const f2 aL = { pa[0], pa[1] }, bL = { pb[0], pb[1] }, cL = { pc[0], pc[1] };
                const float a2 = pa[2], b2 = pb[2], c2 = pc[2];
                heL[r] = aL + bL * 6.f + cL; hoL[r] = (bL + cL) * 4.f;
                he2[r] = a2 + b2 * 6 + c2;   ho2[r] = (b2 + c2) * 4;
            }
            const f2 g00L = { g00.c[0], g00.c[1] }, g01L = { g01.c[0], g01.c[1] }, g10L = { g10.c[0], g10.c[1] }, g11L = { g11.c[0], g11.c[1] };
            const f2 o00L = g00L - (heL[0] + heL[1] * 6.f + heL[2]) * (1.f / 64);
            const f2 o01L = g01L - (hoL[0] + hoL[1] * 6.f + hoL[2]) * (1.f / 64);
            const f2 o10L = g10L - ((heL[1] + heL[2]) * 4.f) * (1.f / 64);
            const f2 o11L = g11L - ((hoL[1] + hoL[2]) * 4.f) * (1.f / 64);
            const float o00c = g00.c[2] - (he2[0] + he2[1] * 6 + he2[2]) * (1.f / 64);
            const float o01c = g01.c[2] - (ho2[0] + ho2[1] * 6 + ho2[2]) * (1.f / 64);
            const float o10c = g10.c[2] - ((he2[1] + he2[2]) * 4) * (1.f / 64);
            const float o11c = g11.c[2] - ((ho2[1] + ho2[2]) * 4) * (1.f / 64);
            // Laplacian payload: written once, read again only by blend/save -- non-temporal when PF_NT_STORES is defined (A/B)
            auto st3 = [&](T PF_GLOBAL* p, float a, float b, float c2) { PF_STORE(p, a); PF_STORE(p + 1, b); PF_STORE(p + 2, c2); };
            if (s00) { st3(dl, o00L.x, o00L.y, o00c); PF_STORE_W(&dw[0], g00.w); }
            if (s01) { st3(dl + 3, o01L.x, o01L.y, o01c); PF_STORE_W(&dw[1], g01.w); }
            if (s10) { st3(dl + 3 * ts, o10L.x, o10L.y, o10c); PF_STORE_W(&dw[ts], g10.w); }
            if (s11) { st3(dl + 3 * ts + 3, o11L.x, o11L.y, o11c); PF_STORE_W(&dw[ts + 1], g11.w); }
            return;
        }
    }
    if constexpr (!F32) {
        // int16: packed 16-bit arithmetic, channels (0,1) and (2,pad); the edge forms are the interior form on reflected
        // (left) / replicated (right) neighbours, which ja / jc already name
        ss2 he01[3], ho01[3], he2p[3], ho2p[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const BxP a = *reinterpret_cast<const BxP*>(&Bt[rows3[r]][ja]), bq = *reinterpret_cast<const BxP*>(&Bt[rows3[r]][j]),
                      c = *reinterpret_cast<const BxP*>(&Bt[rows3[r]][jc]);
            he01[r] = a.c01 + bq.c01 * (ss2)6 + c.c01; ho01[r] = (bq.c01 + c.c01) * (ss2)4;
            he2p[r] = a.c2p + bq.c2p * (ss2)6 + c.c2p; ho2p[r] = (bq.c2p + c.c2p) * (ss2)4;
        }
        auto up_e = [](ss2 u0, ss2 u1, ss2 u2) { return (u0 + u1 * (ss2)6 + u2 + (ss2)32) >> (ss2)6; };
        auto up_o = [](ss2 u1, ss2 u2) { return ((u1 + u2) * (ss2)4 + (ss2)32) >> (ss2)6; };
        auto px2 = [](const Px& q, ss2& c01, ss2& c2p) { const PxP& v = reinterpret_cast<const PxP&>(q); c01 = (ss2)v.c01; c2p = (ss2)v.c2p; };
        ss2 a01, a2p;
        auto st3p = [&](T PF_GLOBAL* p, ss2 c01, ss2 c2p) { PF_STORE(p, c01.x); PF_STORE(p + 1, c01.y); PF_STORE(p + 2, c2p.x); };
        if (s00) { px2(g00, a01, a2p); st3p(dl, a01 - up_e(he01[0], he01[1], he01[2]), a2p - up_e(he2p[0], he2p[1], he2p[2])); PF_STORE_W(&dw[0], g00.w); }
        if (s01) { px2(g01, a01, a2p); st3p(dl + 3, a01 - up_e(ho01[0], ho01[1], ho01[2]), a2p - up_e(ho2p[0], ho2p[1], ho2p[2])); PF_STORE_W(&dw[1], g01.w); }
        if (s10) { px2(g10, a01, a2p); st3p(dl + 3 * ts, a01 - up_o(he01[1], he01[2]), a2p - up_o(he2p[1], he2p[2])); PF_STORE_W(&dw[ts], g10.w); }
        if (s11) { px2(g11, a01, a2p); st3p(dl + 3 * ts + 3, a01 - up_o(ho01[1], ho01[2]), a2p - up_o(ho2p[1], ho2p[2])); PF_STORE_W(&dw[ts + 1], g11.w); }
        return;
    }
    WT he[3][3], ho[3][3];                                   // [row][channel]: even / odd output column
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const Bx a = Bt[rows3[r]][ja], bq = Bt[rows3[r]][j], c = Bt[rows3[r]][jc];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (F32 && edge) {
                he[r][k] = up_h_val<WT, true>((WT)a.c[k], (WT)bq.c[k], (WT)c.c[k], false, le, re, one);
                ho[r][k] = up_h_val<WT, true>((WT)a.c[k], (WT)bq.c[k], (WT)c.c[k], true, le, re, one);
            } else {
                he[r][k] = up_h_val<WT, false>((WT)a.c[k], (WT)bq.c[k], (WT)c.c[k], false, false, false, false);
                ho[r][k] = up_h_val<WT, false>((WT)a.c[k], (WT)bq.c[k], (WT)c.c[k], true, false, false, false);
            }
        }
    }
    T o00[3], o01[3], o10[3], o11[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o00[k] = sat_sub(g00.c[k], cast_up(he[0][k] + he[1][k] * 6 + he[2][k]));
        o01[k] = sat_sub(g01.c[k], cast_up(ho[0][k] + ho[1][k] * 6 + ho[2][k]));
        o10[k] = sat_sub(g10.c[k], cast_up((he[1][k] + he[2][k]) * 4));
        o11[k] = sat_sub(g11.c[k], cast_up((ho[1][k] + ho[2][k]) * 4));
    }
    auto st3g = [&](T PF_GLOBAL* p, const T* v) { PF_STORE(p, v[0]); PF_STORE(p + 1, v[1]); PF_STORE(p + 2, v[2]); };
    if (s00) { st3g(dl, o00); PF_STORE_W(&dw[0], g00.w); }
    if (s01) { st3g(dl + 3, o01); PF_STORE_W(&dw[1], g01.w); }
    if (s10) { st3g(dl + 3 * ts, o10); PF_STORE_W(&dw[ts], g10.w); }
    if (s11) { st3g(dl + 3 * ts + 3, o11); PF_STORE_W(&dw[ts + 1], g11.w); }
}

// XCD-aware block order: consecutive block ids go round-robin to the 8 XCDs, so block id b is mapped
// to (b & 7) * per + (b >> 3): each XCD (own L2) works on a contiguous run of blocks
__device__ __forceinline__ int xcd_order(int b, int nblk)
{
    const int per = nblk >> 3;
    return (per > 0 && b < per * 8) ? (b & 7) * per + (b >> 3) : b;
}

// ... in runs of C consecutive blocks (a run = C neighbours along a row of the job's grid): round robin over the XCDs run by run
__device__ __forceinline__ int xcd_chunks(int b, int nblk, int C)
{
    const int full = (nblk / (8 * C)) * (8 * C);
    if (b >= full) return b;
    const int k = b >> 3;
    return ((k / C) * 8 + (b & 7)) * C + k % C;
}

// one pyramid level of one frame per launch (pf_options.fused = 3)
template <bool F32, bool FROM_WARP, int LBH, int LNT>
__global__ __launch_bounds__(LNT, 4) void k_level3(LevelOffsets lay, LevelArgs g, FusedWarp wa, const uint8_t* __restrict__ src,
                                                 const PxT<F32>* __restrict__ gw_in, PxT<F32>* __restrict__ gw_out,
                                                 const uint64_t* __restrict__ table)
{
    level3_block<F32, LBH, LNT>(FROM_WARP, lay, g, wa, src, gw_in, gw_out, table, xcd_order(blockIdx.x, g.nbx * g.nby));
}

// k_levels: one launch per keyframe carrying level 0 of frame f, level 1 of frame f-1, ... level L-1 of
// frame f-L+1 (pf_options.fused = 1).  The jobs of a launch are independent -- level i of a frame needs
// GW_i, written by that frame's level i-1 job one launch earlier -- so the small upper levels fill the
// chip alongside level 0 inside ONE grid, with no stream/queue multiplexing involved.  Jobs are laid out
// level 0 first (the short upper-level blocks fill the tail); each starts at a block id that is a multiple of 8.
// What a workgroup needs before its first pixel comes first -- 20 dwords, loaded in one round (k_levels); the rest is read where it is used.
struct LevelJob {
    LevelArgs    g;            // 16 dwords
    int first;                 // first block id
    int from_warp;             // level 0: stage A is the warp of the launch's frame
    int nrect;                 // see LevelLaunch::rect
    int bits_off;              // first word of the job's need bitmap in LevelBatch::need_bits, -1: none (the rectangles decide)
    LevelOffsets lay;
    const void*  gw_in;
    void*        gw_out;
    const uint64_t* table;     // the tile table of the job's frame (device memory)
    BlockRect rect[kMaxRectsUpper];   // job 0 of a launch: LevelBatch::rect0 instead (up to kMaxRects)
};
static_assert(sizeof(LevelArgs) == 64 && offsetof(LevelJob, first) == 64 && offsetof(LevelJob, lay) == 80 && sizeof(LevelJob) == 152, "LevelJob layout (k_levels loads its first 20 dwords by hand)");
// job[k].first, k >= 1: block offset among the upper-level jobs.  tab0: the tile table of job 0's frame when it travels
// in the kernel arguments (tab0_n entries; 0: job 0 reads job[0].table like the others)
// need_r0 (job 0, tile table in the arguments): instead of the need rectangles, a level-0 block decides for itself whether anything
// rendered depends on it -- whether a cell that is rendered (entry != 0, cell flag clear) lies within need_r0 = 3 * 2^L - 2
// pixels of it, the reach of the pyramid (`need` recursion of FusionMap::render_frame: pyrDown reads [2p-2, 2p+2], pyrUp +-1).
// Exact at cell granularity, where eight bounding boxes are not (profiles/r04_ab.md).
// kernel-argument words by explicit scalar loads (s_load_dwordx16 / x4 through the constant address space), issued together
#define PF_CONST __attribute__((address_space(4)))
typedef uint32_t su16 __attribute__((ext_vector_type(16)));
typedef uint32_t su8 __attribute__((ext_vector_type(8)));
typedef uint32_t su4 __attribute__((ext_vector_type(4)));
typedef uint32_t su2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void load_words(const char PF_CONST* p, uint32_t (&w)[N])
{
    static_assert(N % 4 == 0, "whole 16-byte groups");
    int i = 0;
#pragma unroll
    for (; i + 16 <= N; i += 16) { const su16 v = *(const su16 PF_CONST*)(p + 4 * i); for (int k = 0; k < 16; k++) w[i + k] = v[k]; }
#pragma unroll
    for (; i + 4 <= N; i += 4) { const su4 v = *(const su4 PF_CONST*)(p + 4 * i); for (int k = 0; k < 4; k++) w[i + k] = v[k]; }
}

// compact0 (job 0 with need rectangles that cover well under its whole grid -- a shard's scattered hash cells above all): the launch carries
// one workgroup per block INSIDE the rectangles instead of one per block of the grid; rect_first[k] = blocks of rectangles 0 .. k-1,
// compact0 = their total, rect_inv[k] = ceil(2^32 / width of rectangle k) (0: width 1).  A block that lies in two rectangles runs in the first.
struct LevelBatch { int njobs, upper_groups, total_groups, sequential, tab0_n, rect_runs, need_r0, compact0; LevelJob job[kMaxLevels - 1]; BlockRect rect0[kMaxRects];
                    int rect_first[kMaxRects]; unsigned rect_inv[kMaxRects]; uint32_t need_bits[kNeedWords]; uint64_t tab0[kArgTable]; };
static_assert(sizeof(LevelBatch) + sizeof(FusedWarp) + 2 * sizeof(void*) <= 4096, "kernel arguments of k_levels: 4 KB");
static_assert(sizeof(LevelBatch) + sizeof(FusedWarp) + 16 <= 4096, "kernel arguments are limited to 4 KB");

template <bool F32, int LBH, int LNT, bool STAMP = false, int ILP = 2, bool PATCH = false, bool WA = false>
__global__ __launch_bounds__(LNT, (LNT == 512 && !PATCH) ? ((LBH == 24 || (!F32 && (ILP == 2 || ILP == 0))) ? 8 : 6) : 4) void k_levels(LevelBatch batch, FusedWarp wa, const uint8_t* __restrict__ src, unsigned long long* stamps)
{
    // Block ids are dealt in groups of 8 (one per XCD).  Group g belongs to the FIRST job (level 0 of the newest frame
    // when there is one) or to the upper-level jobs: last in the grid by default, or (PF_INTERLEAVE_JOBS, diagnostics)
    // spread evenly between the first job's groups.
    unsigned long long t_entry = 0;
    if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory");      // stamped build: kernel entry
    // The prologue is a chain of scalar loads from the kernel-argument segment, and a workgroup holds its slot (80 VGPRs x 512 threads,
    // 52 KB of LDS) while it waits for them: written as `B.job[j].field` wherever a field is needed, the compiler emits about ten
    // DEPENDENT load / s_waitcnt rounds before the first pixel (measured with the stamped build, profiles/r05_ab.md: 3600 cycles median
    // of a level-0 workgroup's 24 700).  So: ONE round for everything whose address does not depend on data -- the header, every job's first
    // block, job 0 whole (speculative: three in four workgroups are level-0 ones) -- then at most one more for an upper-level job.
    const char PF_CONST* const ka = (const char PF_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
    const LevelBatch PF_CONST& B = *(const LevelBatch PF_CONST*)ka;          // `batch` as it lies in the kernel-argument segment: every access a scalar load
    (void)batch;
    static_assert(offsetof(LevelBatch, job) == 32, "the header is one s_load_dwordx8");
    const su8 hdr = *(const su8 PF_CONST*)ka;                                // njobs, upper_groups, total_groups, sequential, tab0_n, rect_runs, need_r0, compact0
    su16 ja = *(const su16 PF_CONST*)(ka + 32); su4 jb = *(const su4 PF_CONST*)(ka + 32 + 64);      // job 0: LevelArgs | first, from_warp, nrect, bits_off
    int first[kMaxLevels];
#pragma unroll
    for (int k = 1; k < kMaxLevels - 1; k++) first[k] = *(const int PF_CONST*)(ka + offsetof(LevelBatch, job) + k * sizeof(LevelJob) + offsetof(LevelJob, first));
    // (the empty statements pin the loads here: left alone, the compiler sinks each one to its first use and the chain is back)
    asm volatile("" :: "s"(hdr), "s"(ja), "s"(jb));
    asm volatile("" :: "s"(first[1]), "s"(first[2]), "s"(first[3]), "s"(first[4]), "s"(first[5]), "s"(first[6]), "s"(first[7]));
    static_assert(kMaxLevels == 9, "first[1..7]: a launch carries at most kMaxLevels - 1 level jobs");
    const int njobs = (int)hdr[0], ug = (int)hdr[1], tg = (int)hdr[2], seq = (int)hdr[3], tab0_n = (int)hdr[4], rect_runs = (int)hdr[5], need_r0 = (int)hdr[6], compact0 = (int)hdr[7];
    const int g = (int)blockIdx.x >> 3, lane8 = (int)blockIdx.x & 7;
    int u0, u1;
    if (!kExp || seq == 1) { u0 = g < tg - ug ? 0 : g - (tg - ug); u1 = g < tg - ug ? 0 : u0 + 1; }      // upper levels last (default; the product's only order)
    else if (seq == 2) { u0 = g < ug ? g : ug; u1 = g < ug ? g + 1 : ug; }                          // upper levels first (PF_UPPER_FIRST)
    else { u0 = (int)(((long)g * ug) / tg); u1 = (int)(((long)(g + 1) * ug) / tg); }                 // dealt between the first job's groups (PF_INTERLEAVE_JOBS)
    int j = 0, b;
    if (u1 > u0) {                                            // an upper-level group
        b = u0 * 8 + lane8;
        j = 1;
        int fj = first[1];
#pragma unroll
        for (int k = 2; k < kMaxLevels - 1; k++) if (k < njobs && b >= first[k]) { j = k; fj = first[k]; }
        b -= fj;
        j = __builtin_amdgcn_readfirstlane(j); b = __builtin_amdgcn_readfirstlane(b);      // uniform by construction: keep the job in scalar registers
        const char PF_CONST* jp = ka + offsetof(LevelBatch, job) + j * sizeof(LevelJob);
        ja = *(const su16 PF_CONST*)jp; jb = *(const su4 PF_CONST*)(jp + 64);
    } else
        b = (g - u0) * 8 + lane8;
    // the job: its first 20 dwords from the registers just loaded, the rest (level offsets, GW buffers, table, rectangles) read from the
    // kernel arguments where it is used
    struct JobHead { LevelArgs g; int first, from_warp, nrect, bits_off; } J;
    {
        uint32_t jw[20];
#pragma unroll
        for (int k = 0; k < 16; k++) jw[k] = ja[k];
#pragma unroll
        for (int k = 0; k < 4; k++) jw[16 + k] = jb[k];
        __builtin_memcpy(&J, jw, sizeof J);
    }
    const LevelJob PF_CONST& Jc = B.job[j];
    const int nblk = J.g.nbx * J.g.nby;
    int bb;
    if (kExp && j == 0 && compact0) {          // experiments build, PF_COMPACT=1: measured, no change for a shard of 8 (profiles/r05_predicted_scaling.md)
        // one workgroup per block inside the need rectangles: b -> rectangle k -> (bx, by); everything in one round of loads
        if (b >= compact0) return;                             // padding up to the next multiple of 8
        static_assert(kMaxRects == 8 && sizeof(BlockRect) == 8, "rect0 is one s_load_dwordx16, rect_first / rect_inv one s_load_dwordx8 each");
        const su16 rw = *(const su16 PF_CONST*)(ka + offsetof(LevelBatch, rect0));
        const su8 rf = *(const su8 PF_CONST*)(ka + offsetof(LevelBatch, rect_first)), ri = *(const su8 PF_CONST*)(ka + offsetof(LevelBatch, rect_inv));
        asm volatile("" :: "s"(rw), "s"(rf), "s"(ri));
        int k = 0;
#pragma unroll
        for (int i = 1; i < kMaxRects; i++) if (i < J.nrect && b >= (int)rf[i]) k = i;
        uint32_t lo = rw[0], hi = rw[1], fk = rf[0], ik = ri[0];
#pragma unroll
        for (int i = 1; i < kMaxRects; i++) if (k == i) { lo = rw[2 * i]; hi = rw[2 * i + 1]; fk = rf[i]; ik = ri[i]; }
        const int rx0 = (short)(lo & 0xffff), ry0 = (short)(lo >> 16), rx1 = (short)(hi & 0xffff);
        const unsigned local = (unsigned)b - fk, ly = ik ? __umulhi(local, ik) : local;
        const int bx = rx0 + (int)(local - ly * (unsigned)(rx1 - rx0)), by = ry0 + (int)ly;
        bool earlier = false;
#pragma unroll
        for (int i = 0; i < kMaxRects - 1; i++) {
            const int ax0 = (short)(rw[2 * i] & 0xffff), ay0 = (short)(rw[2 * i] >> 16), ax1 = (short)(rw[2 * i + 1] & 0xffff), ay1 = (short)(rw[2 * i + 1] >> 16);
            earlier = earlier || (i < k && bx >= ax0 && bx < ax1 && by >= ay0 && by < ay1);
        }
        if (earlier) return;                                   // this block runs in an earlier rectangle
        bb = by * J.g.nbx + bx;
    } else {
        if (b >= nblk) return;                                 // padding up to the next multiple of 8
        // with need rectangles (a shard, or tiles culled) whole bands of the grid exit at once: contiguous runs per XCD would leave some
        // XCDs without work, so the blocks are dealt round robin instead (PF_RECT_ORDER=1 keeps the runs, for A/B)
        const int rr = kExp ? rect_runs : 0;                  // PF_RECT_ORDER (A/B): 1 XCD runs even with rectangles, 2 round robin always, >= 3 runs of 4 / 8 / 16 blocks
        bb = rr >= 3 ? xcd_chunks(b, nblk, 1 << (rr >= 3 ? rr - 1 : 0)) : ((J.nrect && !rr) || rr == 2) ? b : xcd_order(b, nblk);
    }
    // the newest frame's tile table arrived in the kernel arguments: one workgroup stores it where the launches that
    // carry this frame's upper levels will read it (kernel boundaries order that)
    // (addressed through the kernel-argument segment pointer: taking the address of the by-value member costs registers)
    const uint64_t* tab0 = nullptr;
    if (j == 0 && tab0_n) {
        tab0 = (const uint64_t*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LevelBatch, tab0));
        if (b == 0) for (int i = threadIdx.x; i < tab0_n; i += LNT) const_cast<uint64_t*>(Jc.table)[i] = tab0[i];
    }
    if (j == 0 && tab0 && need_r0 && J.nrect) {
        // job 0: is a rendered cell within the pyramid's reach of this block?  Tile by tile: the cells of the tile inside the reach
        // as a 16-bit mask against the entry's culled cells
        int bx, by; block_xy(J.g, bb, bx, by); const int r0 = need_r0;
        int x0 = J.g.cx0 + bx * LBW - r0, x1 = J.g.cx0 + bx * LBW + LBW - 1 + r0, y0 = J.g.cy0 + by * LBH - r0, y1 = J.g.cy0 + by * LBH + LBH - 1 + r0;
        x0 = x0 > 0 ? x0 : 0; y0 = y0 > 0 ? y0 : 0; x1 = x1 < J.g.cols - 1 ? x1 : J.g.cols - 1; y1 = y1 < J.g.rows - 1 ? y1 : J.g.rows - 1;
        // need_r0 <= 96: the reach spans two tiles a side at most -- four entries, the cells of each inside the reach as a 16-bit mask
        const int cx0 = x0 >> 6, cx1 = x1 >> 6, cy0 = y0 >> 6, cy1 = y1 >> 6;
        const int tx0 = cx0 >> 2, tx1 = cx1 >> 2, ty0 = cy0 >> 2, ty1 = cy1 >> 2;
        auto cols = [](int a, int b) { return ((0xfu << a) & (0xfu >> (3 - b))) * 0x1111u; };
        auto rows = [](int a, int b) { return (0xffffu << (4 * a)) & (0xffffu >> (4 * (3 - b))); };
        const unsigned mx0 = cols(cx0 & 3, tx1 > tx0 ? 3 : cx1 & 3), mx1 = cols(tx1 > tx0 ? 0 : cx0 & 3, cx1 & 3);
        const unsigned my0 = rows(cy0 & 3, ty1 > ty0 ? 3 : cy1 & 3), my1 = rows(ty1 > ty0 ? 0 : cy0 & 3, cy1 & 3);
        const uint64_t PF_CONST* t0c = (const uint64_t PF_CONST*)(ka + offsetof(LevelBatch, tab0));
        const uint64_t e00 = t0c[ty0 * J.g.tiles_x + tx0], e01 = t0c[ty0 * J.g.tiles_x + tx1];
        const uint64_t e10 = t0c[ty1 * J.g.tiles_x + tx0], e11 = t0c[ty1 * J.g.tiles_x + tx1];
        asm volatile("" :: "s"(e00), "s"(e01), "s"(e10), "s"(e11));               // one round of four loads
        auto rendered = [](uint64_t e, unsigned m) { return (int)(e != 0 && (~((uint32_t)(e >> 32) >> 16) & m) != 0); };
        const bool hit = (rendered(e00, my0 & mx0) | rendered(e01, my0 & mx1) | rendered(e10, my1 & mx0) | rendered(e11, my1 & mx1)) != 0;
        if (!hit) return;
    } else
    if (J.bits_off >= 0) {                                     // an upper-level job with its need bitmap in the kernel arguments
        if (!((B.need_bits[J.bits_off + (bb >> 5)] >> (bb & 31)) & 1u)) return;
    } else
    if (J.nrect && !(kExp && j == 0 && compact0)) {                    // a shard: does any tile of this rank depend on the block?  (compact0: it does by construction)
        int bx, by; block_xy(J.g, bb, bx, by);
        bool hit = false;
        for (int k = 0; k < J.nrect; k++) {
            // (read where they lie in the kernel arguments: indexing J.rect would put the job on the stack)
            const char PF_CONST* rp = j == 0 ? ka + offsetof(LevelBatch, rect0) + k * sizeof(BlockRect)
                                             : ka + offsetof(LevelBatch, job) + j * sizeof(LevelJob) + offsetof(LevelJob, rect) + (k < kMaxRectsUpper ? k : 0) * sizeof(BlockRect);
            const uint64_t rw = *(const uint64_t PF_CONST*)rp;
            BlockRect r; r.x0 = (short)(rw & 0xffff); r.y0 = (short)((rw >> 16) & 0xffff); r.x1 = (short)((rw >> 32) & 0xffff); r.y1 = (short)(rw >> 48);
            hit = hit || (bx >= r.x0 && bx < r.x1 && by >= r.y0 && by < r.y1);
        }
        if (!hit) return;
    }
    unsigned long long* st = nullptr;
    if (STAMP) { st = stamps + (size_t)blockIdx.x * 8; if (threadIdx.x == 0) { st[6] = (unsigned long long)j | t_entry << 4; st[7] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 20) /* HW_REG_XCC_ID */; } }
    // The block itself reads the job's fields where they lie in the kernel arguments, as it needs them (a scalar load behind other waves' work):
    // kept in registers from the prologue on, the sixteen words of LevelArgs push the kernel over its SGPR budget and the spills cost VGPRs.  The
    // laundered pointer keeps the compiler from forwarding the prologue's copies.
    const char PF_CONST* kb = ka;
    asm volatile("" : "+s"(kb));
    const LevelJob PF_CONST& Jl = *(const LevelJob PF_CONST*)(kb + offsetof(LevelBatch, job) + j * sizeof(LevelJob));
    level3_block<F32, LBH, LNT, STAMP, ILP, PATCH, WA>(J.from_warp != 0, Jl.lay, Jl.g, wa, src, (const PxT<F32>*)Jl.gw_in, (PxT<F32>*)Jl.gw_out,
                                                       Jl.table, bb, st, tab0);
}

// radial_weight() forms dis = fma(dx, dx, dy * dy): one rounding, equal to the reference's RN(RN(dy^2) + RN(dx^2)) only while
// both squares are exact, i.e. |dx|, |dy| < 2^12.  Frames with a side of 8192 px or more (half extent >= 4096) gather the
// weight from the plane instead (the reference's own weightImage, built by launch_weight32).
static bool radial_weight_exact(const WarpArgs& wa) { return wa.srows < 8192 && wa.scols < 8192; }

#if PF_EXPERIMENTS
#include "strips.inc"
#endif

// FusedWarp::plain: see there.  Experiments library, PF_FORCE_GENERAL=1 (tests): every pixel through the general forms.
static int plain_homography(const WarpArgs& wa)
{
    static const bool force_general = exp_env("PF_FORCE_GENERAL") != nullptr;      // experiments library
    if (force_general || wa.srows > 32767 || wa.scols > 32767) return 0;
    for (int i = 0; i < 9; i++) if (!(std::fabs(wa.M[i]) < 0x1p400)) return 0;
    return 1;
}

#if PF_EXPERIMENTS
// Can this frame's level 0 run with LDS-staged source patches (k_levels<..., PATCH>)?  Needs, over the whole canvas plus
// the 4-pixel halo of the blocks: W of one sign and every source coordinate far inside the int range (then no pixel
// needs warp_fetch_pre's range test), and a patch -- the source rectangle a 71x39 block of canvas pixels maps into,
// bounded from the map's behaviour at the canvas corners and centre, two pixels of margin -- that fits the LDS budget.
// Fills w.Mf / phx / phy / pitch_*.  A pixel that falls outside its block's patch all the same (the bound is an estimate
// for strongly projective maps) takes the global-memory path inside the kernel: results never depend on this plan.
// Over the whole canvas plus the 4-pixel halo of the blocks: W of one sign, away from zero, and every source coordinate
// far inside the int range?  Numerators and W are affine in (x, y), so their extremes over the rectangle are at its corners.
// 1e7 leaves the kernel's own per-pixel bound (3e7) a wide margin for the block-relative evaluation order.
static bool tame_canvas(const double* M, int crows, int ccols)
{
    const double xs[2] = { -4.0, ccols + 3.0 }, ys[2] = { -4.0, crows + 3.0 };
    double wmin = 0, nmax = 0; int sign = 0;
    for (int i = 0; i < 4; i++) {
        const double x = xs[i & 1], y = ys[i >> 1], W = M[6] * x + M[7] * y + M[8];
        if (!(std::fabs(W) > 0x1p-400)) return false;
        const int sg = W > 0 ? 1 : -1;
        if (sign && sg != sign) return false;
        sign = sg;
        wmin = i ? std::min(wmin, std::fabs(W)) : std::fabs(W);
        nmax = std::max(nmax, std::max(std::fabs(M[0] * x + M[1] * y + M[2]), std::fabs(M[3] * x + M[4] * y + M[5])));
    }
    return nmax / wmin < 1.0e7;
}

static bool patch_plan(const WarpArgs& wa, int crows, int ccols, int block_rows, FusedWarp& w)
{
    // Measured on MI355X (profiles/r03_patch_stage_a.md): the staged form needs 2 workgroups per CU instead of 3 and is
    // slower than the global gather on cfg-A -- it runs on request only (PF_PATCH=1; parity-tested like the default path)
    static const bool on = getenv("PF_PATCH") != nullptr;
    if (!on || !w.plain || block_rows != 32) return false;
    const double* M = wa.M;
    auto map = [&](double x, double y, double& px, double& py, double& W) {
        W = M[6] * x + M[7] * y + M[8];
        px = (M[0] * x + M[1] * y + M[2]) / W; py = (M[3] * x + M[4] * y + M[5]) / W;
    };
    if (!tame_canvas(M, crows, ccols)) return false;
    double ex = 0, ey = 0;
    const double cx[5] = { 32.0, ccols - 32.0, 32.0, ccols - 32.0, ccols * 0.5 }, cy[5] = { 16.0, 16.0, crows - 16.0, crows - 16.0, crows * 0.5 };
    for (int k = 0; k < 5; k++) {
        double p0x, p0y, W;
        map(cx[k], cy[k], p0x, p0y, W);
        for (int i = 0; i < 4; i++) {
            double qx, qy;
            map(cx[k] + ((i & 1) ? 36.0 : -36.0), cy[k] + ((i >> 1) ? 20.0 : -20.0), qx, qy, W);
            ex = std::max(ex, std::fabs(qx - p0x)); ey = std::max(ey, std::fabs(qy - p0y));
        }
    }
    if (!(ex < 4096 && ey < 4096)) return false;
    const int hx = (int)std::ceil(ex) + 2, hy = (int)std::ceil(ey) + 2, pw = 2 * hx + 2, ph = 2 * hy + 2, cn = wa.src_cn;
    const int pitch_i = (15 + (pw - 2) * cn + 8 + 15) & ~15, pitch_w = (15 + pw * 4 + 15) & ~15;
    if ((pitch_i >> 4) + (pitch_w >> 4) > 64) return false;                    // one wave stages one patch row per round
    if ((long)ph * (pitch_i + pitch_w) > kPatchBytes) return false;
    for (int i = 0; i < 9; i++) w.Mf[i] = (float)M[i];
    w.phx = hx; w.phy = hy; w.pitch_i = pitch_i; w.pitch_w = pitch_w;
    return true;
}

#endif   // PF_EXPERIMENTS

static unsigned level_inv_nbx(int nbx, int nby)          // LevelArgs::inv_nbx, see block_xy
{
    if (nbx <= 1 || (unsigned long long)nbx * nby * nbx >= 0x100000000ull) return 0;
    return (unsigned)((0x100000000ull + (unsigned)nbx - 1) / (unsigned)nbx);
}

// FusedWarp::seed_ok: over the canvas and the blocks' 4-pixel halo W = M6 x + M7 y + M8 keeps one sign (affine: its extremes are at the corners)
// and one canvas row changes it by at most 2^-16 of its smallest magnitude -- half the 2^-15 rcp_seeded is argued for, which itself sits
// 2^3.5 inside the seed error at which its two Newton steps stop reaching the fma roundings' level.
static int seed_plan(const double* M, int rows, int cols)
{
    static const bool on = kExp && getenv("PF_SEED") != nullptr;  // experiments build, on request: bit-exact and no faster than v_rcp_f64 per pixel
    if (!on) return 0;
    const double xs[2] = { -4.0, cols + 3.0 }, ys[2] = { -4.0, rows + 3.0 };
    double wmin = 0; int sign = 0;
    for (int i = 0; i < 4; i++) {
        const double W = M[6] * xs[i & 1] + M[7] * ys[i >> 1] + M[8];
        if (!(std::fabs(W) > 0x1p-400) || !std::isfinite(W)) return 0;
        const int sg = W > 0 ? 1 : -1;
        if (sign && sg != sign) return 0;
        sign = sg;
        wmin = i ? std::min(wmin, std::fabs(W)) : std::fabs(W);
    }
    return std::fabs(M[7]) <= 0x1p-16 * wmin;
}

size_t level_px_bytes(bool f32) { return f32 ? sizeof(PxT<true>) : sizeof(PxT<false>); }

void launch_level(hipStream_t s, const TileLayout& lay, int level, int rows, int cols, int cx0, int cy0, int cx1, int cy1,
                  int tiles_x, bool top_select, bool write_next, const WarpArgs* wa, const uint8_t* src,
                  const void* gw_in, void* gw_out, const uint64_t* table, int shape)
{
    LevelArgs g{};
    g.level = level; g.rows = rows; g.cols = cols; g.cx0 = cx0; g.cy0 = cy0; g.cx1 = cx1; g.cy1 = cy1;
    g.tiles_x = tiles_x; g.top_select = top_select; g.write_next = write_next;
    // Two code shapes of the same computation: k_level (4 barriers, H tile in LDS, 64x16 blocks) and
    // k_level3 (2 barriers, no H tile, 2x2 output quads, 64x32 blocks; 73 VGPRs / 51.6 KB LDS fp32 ->
    // 3 workgroups per CU, 56 VGPRs / 38.1 KB int16 -> 4).  Measured on MI355X (cfg-A) k_level3 is
    // the faster one for both pyramid types; shape 2 (pf_options.fused = 2) or PF_KLEVEL=4 selects k_level.
    static const int force = exp_env_int("PF_KLEVEL", 0);                      // experiments library
    static const int ablate = kExp && getenv("PF_ABLATE") ? atoi(getenv("PF_ABLATE")) : 0;
    const bool use4 = force == 4 || (force != 3 && shape == 2);
    const int BH = use4 ? 16 : 32;
    const int LSTEPS = 1;     // rolling strips (k_level<..., LS>1>) measured slower on MI355X: kept at one block per workgroup
    g.nbx = (cx1 - cx0 + LBW - 1) / LBW; g.nby = (cy1 - cy0 + BH * LSTEPS - 1) / (BH * LSTEPS);
    g.inv_nbx = level_inv_nbx(g.nbx, g.nby);
    g.ablate = ablate;
    if (g.nbx <= 0 || g.nby <= 0) return;
    dim3 grid(g.nbx * g.nby);
    FusedWarp w{};
    if (wa) {
        for (int i = 0; i < 9; i++) w.M[i] = wa->M[i];
        w.total = frame_bytes(wa->srows, wa->scols, wa->sstep, wa->src_cn); w.wmap = wa->wmap;
        w.srows = wa->srows; w.scols = wa->scols; w.sstep = (int)wa->sstep; w.cn = wa->src_cn;
        w.plain = plain_homography(*wa);
    }
    LevelOffsets lo{ lay.lap_off[level], lay.w_off[level], lay.lap_off[level + 1], lay.w_off[level + 1] };
#define PF_LAUNCH(K, F, W, ...) hipLaunchKernelGGL((K<F, W, __VA_ARGS__>), grid, dim3(512), 0, s, lo, g, w, src, (const PxT<F>*)gw_in, (PxT<F>*)gw_out, table)
    if (use4) {
        if (lay.f32) { if (wa) PF_LAUNCH(k_level, true, true, 16, 512, 1); else PF_LAUNCH(k_level, true, false, 16, 512, 1); }
        else         { if (wa) PF_LAUNCH(k_level, false, true, 16, 512, 1); else PF_LAUNCH(k_level, false, false, 16, 512, 1); }
    } else {
        if (lay.f32) { if (wa) PF_LAUNCH(k_level3, true, true, 32, 512); else PF_LAUNCH(k_level3, true, false, 32, 512); }
        else         { if (wa) PF_LAUNCH(k_level3, false, true, 32, 512); else PF_LAUNCH(k_level3, false, false, 32, 512); }
    }
#undef PF_LAUNCH
}


static unsigned long long* g_stamp_buf = nullptr;
static int g_stamp_blocks = 0;
[[maybe_unused]] constexpr int kStampBlocks = 1 << 16;

// launches of the pipelined level kernel by form since the library was loaded (tests/test_gpu_variants.py asserts that the form a
// switch asks for really ran): 0 block form, computed weight (default); 1 block form, weight plane gather; 2 LDS-staged source patch;
// 3 rolling strips; 4 64x64 blocks; 5 64x28 blocks; 6 stamped instantiation; 7 tile table in the kernel arguments
static long long g_form_counts[8] = {};
void read_form_counts(long long out[8]) { for (int i = 0; i < 8; i++) out[i] = g_form_counts[i]; }
static long long g_compact_launches = 0;                          // pipelined launches whose level-0 job ran on the compact grid (LevelBatch::compact0)
long long read_compact_launches() { return g_compact_launches; }

#if PF_EXPERIMENTS
// The wave-specialised rolling-strip form of the pipelined launch (strips.inc).  Returns false when this launch has to take
// the block form (diagnostic builds, the gathered weight plane).
static bool launch_strips(hipStream_t s, const TileLayout& lay, const LevelLaunch* jobs, int njobs, const WarpArgs* wa, const uint8_t* src)
{
    static const int on = getenv("PF_STRIPS") ? atoi(getenv("PF_STRIPS")) : 0;      // default off until it beats the block form (profiles/r04_strips.md)
    static const int sablate = getenv("PF_SABLATE") ? atoi(getenv("PF_SABLATE")) : 0;      // timing only: 1 no warp, 2 no pyrDown, 4 no select, 8 no weight prefetch
    static const bool stamp = getenv("PF_STAMP") != nullptr;
    static const bool other = getenv("PF_WEIGHT_PLANE") || getenv("PF_PATCH") || getenv("PF_BLOCK64") || getenv("PF_ABLATE") || getenv("PF_A_ILP");
    static const int order = getenv("PF_UPPER_FIRST") ? 2 : (getenv("PF_INTERLEAVE_JOBS") ? 0 : 1);
    if (!on || other) return false;
    if (wa && !radial_weight_exact(*wa)) return false;
    constexpr int R = 4;
    static const int seg_env = getenv("PF_STRIP_SEG") ? atoi(getenv("PF_STRIP_SEG")) : 0;
    const int seg = seg_env >= R ? (seg_env / R) * R : 64;
    LevelBatch batch{};
    int first_blocks = 0, upper_blocks = 0;
    for (int k = 0; k < njobs; k++) {
        const LevelLaunch& q = jobs[k];
        LevelJob& J = batch.job[batch.njobs];
        J.g.level = q.level; J.g.rows = q.rows; J.g.cols = q.cols; J.g.cx0 = q.cx0; J.g.cy0 = q.cy0; J.g.cx1 = q.cx1; J.g.cy1 = q.cy1;
        J.g.tiles_x = q.tiles_x; J.g.top_select = q.top_select; J.g.write_next = q.write_next; J.g.ablate = sablate;
        J.g.seg = seg;
        J.g.nbx = (q.cx1 - q.cx0 + strips::KW - 1) / strips::KW; J.g.nby = (q.cy1 - q.cy0 + seg - 1) / seg;
        if (J.g.nbx <= 0 || J.g.nby <= 0) continue;
        if ((q.cx0 | q.cy0 | q.cy1) & 1) return false;          // strips start on even columns / rows (they always do: regions are tile- or 2x-aligned)
        J.lay = LevelOffsets{ lay.lap_off[q.level], lay.w_off[q.level], lay.lap_off[q.level + 1], lay.w_off[q.level + 1] };
        J.gw_in = q.gw_in; J.gw_out = q.gw_out; J.table = q.table; J.from_warp = q.from_warp;
        // job 0 keeps its rectangles in LevelBatch::rect0 (up to kMaxRects), the others in their own (kMaxRectsUpper; more than that -- the
        // host does not produce them -- become their common bounding box: a superset is always right)
        {
            const int cap = batch.njobs == 0 ? kMaxRects : kMaxRectsUpper;
            J.nrect = q.nrect;
            if (q.nrect > cap) {
                BlockRect u = q.rect[0];
                for (int r = 1; r < q.nrect && r < kMaxRects; r++) { u.x0 = std::min(u.x0, q.rect[r].x0); u.y0 = std::min(u.y0, q.rect[r].y0); u.x1 = std::max(u.x1, q.rect[r].x1); u.y1 = std::max(u.y1, q.rect[r].y1); }
                J.nrect = 1;
                if (batch.njobs == 0) batch.rect0[0] = u; else J.rect[0] = u;
            } else
                for (int r = 0; r < q.nrect; r++) { if (batch.njobs == 0) batch.rect0[r] = q.rect[r]; else J.rect[r] = q.rect[r]; }
        }
        J.bits_off = -1;
        if (batch.njobs == 0 && q.table_args && q.table_n > 0 && q.table_n <= kArgTable) {
            batch.tab0_n = q.table_n;
            for (int i = 0; i < q.table_n; i++) batch.tab0[i] = q.table_args[i];
        }
        const int padded = (J.g.nbx * J.g.nby + 7) & ~7;
        if (batch.njobs == 0) { J.first = 0; first_blocks = padded; }
        else { J.first = upper_blocks; upper_blocks += padded; }
        batch.njobs++;
    }
    if (!batch.njobs) return true;
    const int nblocks = first_blocks + upper_blocks;
    batch.upper_groups = upper_blocks / 8; batch.total_groups = nblocks / 8; batch.sequential = order;
    FusedWarp w{};
    if (wa) {
        for (int i = 0; i < 9; i++) w.M[i] = wa->M[i];
        w.total = frame_bytes(wa->srows, wa->scols, wa->sstep, wa->src_cn); w.wmap = wa->wmap;
        w.srows = wa->srows; w.scols = wa->scols; w.sstep = (int)wa->sstep; w.cn = wa->src_cn;
        w.plain = plain_homography(*wa);
        w.wxc = wa->xc; w.wyc = wa->yc; w.wdmax = wa->dis_max; w.wrcp = (float)(1.0L / (long double)wa->dis_max); w.wtype = wa->weight_type;
    }
    constexpr int P = PF_S_P;
    g_form_counts[3]++;
    if (batch.tab0_n) g_form_counts[7]++;
    if (stamp) {
        // diagnostic build (tools/strip_roles.py): 32 u64 per workgroup -- per wave {cycles waiting at the period barriers, lifetime}, [30] job, [31] periods
        if (!g_stamp_buf) { if (hipMalloc((void**)&g_stamp_buf, (size_t)kStampBlocks * 64) != hipSuccess) g_stamp_buf = nullptr; }
        if (g_stamp_buf && nblocks * 4 <= kStampBlocks) {
            (void)hipMemsetAsync(g_stamp_buf, 0, (size_t)kStampBlocks * 64, s);
            g_stamp_blocks = nblocks * 4;
            if (lay.f32) hipLaunchKernelGGL((k_strips<true, R, P, true, true>), dim3(nblocks), dim3((P + 4) * 64), 0, s, batch, w, src, g_stamp_buf);
            else         hipLaunchKernelGGL((k_strips<false, R, P, true, true>), dim3(nblocks), dim3((P + 4) * 64), 0, s, batch, w, src, g_stamp_buf);
            return true;
        }
    }
    if (lay.f32) hipLaunchKernelGGL((k_strips<true, R, P, true>), dim3(nblocks), dim3((P + 4) * 64), 0, s, batch, w, src, (unsigned long long*)nullptr);
    else         hipLaunchKernelGGL((k_strips<false, R, P, true>), dim3(nblocks), dim3((P + 4) * 64), 0, s, batch, w, src, (unsigned long long*)nullptr);
    return true;
}

#endif   // PF_EXPERIMENTS

int level0_need_reach(const TileLayout& lay, int table_n, int nrect0)
{
    static const bool off = kExp && getenv("PF_NO_NEED_R0") != nullptr;                                 // A/B: the need rectangles for job 0 too
    static const bool strips = kExp && getenv("PF_STRIPS") && atoi(getenv("PF_STRIPS")) != 0;             // the strip form keeps the rectangles
    const int bh = level_block_rows(lay.f32 != 0);
    if (off || strips || table_n <= 0 || table_n > kArgTable || nrect0 <= 0 || !(bh == 32 || (kExp && bh == 24))) return 0;
    const int r0 = 3 * (1 << (lay.nlev - 1)) - 2;
    return r0 <= 96 ? r0 : 0;                                     // beyond five bands the reach spans more than two tiles: the rectangles
}

void launch_levels(hipStream_t s, const TileLayout& lay, const LevelLaunch* jobs, int njobs, const WarpArgs* wa, const uint8_t* src)
{
#if PF_EXPERIMENTS
    if (level_block_rows(lay.f32 != 0) == 32 && launch_strips(s, lay, jobs, njobs, wa, src)) return;
#endif
    static const int ablate = kExp && getenv("PF_ABLATE") ? atoi(getenv("PF_ABLATE")) : 0;
    const int BH = level_block_rows(lay.f32 != 0);
    LevelBatch batch{};
    int first_blocks = 0, upper_blocks = 0, bits_words = 0;
    static const bool use_bits = !(kExp && getenv("PF_NO_NEED_BITS"));       // A/B: the rectangles for the upper levels
    for (int k = 0; k < njobs; k++) {
        const LevelLaunch& q = jobs[k];
        LevelJob& J = batch.job[batch.njobs];
        J.g.level = q.level; J.g.rows = q.rows; J.g.cols = q.cols; J.g.cx0 = q.cx0; J.g.cy0 = q.cy0; J.g.cx1 = q.cx1; J.g.cy1 = q.cy1;
        J.g.tiles_x = q.tiles_x; J.g.top_select = q.top_select; J.g.write_next = q.write_next; J.g.ablate = ablate;
        J.g.nbx = (q.cx1 - q.cx0 + LBW - 1) / LBW; J.g.nby = (q.cy1 - q.cy0 + BH - 1) / BH;
        if (J.g.nbx <= 0 || J.g.nby <= 0) continue;
        J.g.inv_nbx = level_inv_nbx(J.g.nbx, J.g.nby);
        J.lay = LevelOffsets{ lay.lap_off[q.level], lay.w_off[q.level], lay.lap_off[q.level + 1], lay.w_off[q.level + 1] };
        J.gw_in = q.gw_in; J.gw_out = q.gw_out; J.table = q.table; J.from_warp = q.from_warp;
        // job 0 keeps its rectangles in LevelBatch::rect0 (up to kMaxRects), the others in their own (kMaxRectsUpper; more than that -- the
        // host does not produce them -- become their common bounding box: a superset is always right)
        {
            const int cap = batch.njobs == 0 ? kMaxRects : kMaxRectsUpper;
            J.nrect = q.nrect;
            if (q.nrect > cap) {
                BlockRect u = q.rect[0];
                for (int r = 1; r < q.nrect && r < kMaxRects; r++) { u.x0 = std::min(u.x0, q.rect[r].x0); u.y0 = std::min(u.y0, q.rect[r].y0); u.x1 = std::max(u.x1, q.rect[r].x1); u.y1 = std::max(u.y1, q.rect[r].y1); }
                J.nrect = 1;
                if (batch.njobs == 0) batch.rect0[0] = u; else J.rect[0] = u;
            } else
                for (int r = 0; r < q.nrect; r++) { if (batch.njobs == 0) batch.rect0[r] = q.rect[r]; else J.rect[r] = q.rect[r]; }
        }
        J.bits_off = -1;
        if (use_bits && !q.from_warp && q.need_bits && q.need_n == J.g.nbx * J.g.nby && bits_words + (q.need_n + 31) / 32 <= kNeedWords) {
            J.bits_off = bits_words;
            for (int w = 0; w < (q.need_n + 31) / 32; w++) batch.need_bits[bits_words + w] = q.need_bits[w];
            bits_words += (q.need_n + 31) / 32;
        }
        if (batch.njobs == 0 && q.table_args && q.table_n > 0 && q.table_n <= kArgTable) {
            batch.tab0_n = q.table_n;
            for (int i = 0; i < q.table_n; i++) batch.tab0[i] = q.table_args[i];
        }
        // job 0 has its own block numbering, jobs 1.. share one (see k_levels)
        const int padded = (J.g.nbx * J.g.nby + 7) & ~7;
        if (batch.njobs == 0) { J.first = 0; first_blocks = padded; }
        else { J.first = upper_blocks; upper_blocks += padded; }
        batch.njobs++;
    }
    // job 0 with rectangles that cover at most three quarters of its grid: one workgroup per block inside them (LevelBatch::compact0).  A rank
    // of 8 launches the whole canvas' 7072 level-0 blocks to run ~900 of them -- and yet the compact grid does not shorten its launches (49.3 k
    // against 50.7 k predicted keyframes/s at 8 ranks, same box: the empty workgroups were not what a shard waits for).  Measured, not adopted.
    static const bool compact = kExp && getenv("PF_COMPACT") != nullptr;            // experiments build, on request
    if (compact && batch.njobs && batch.job[0].from_warp && batch.job[0].nrect > 0 && BH == 32) {
        const LevelJob& J0 = batch.job[0];
        long total = 0; bool ok = true;
        for (int r = 0; r < J0.nrect; r++) {
            const BlockRect& q = batch.rect0[r];
            const int w = q.x1 - q.x0, h = q.y1 - q.y0;
            if (w < 0 || h < 0 || q.x0 < 0 || q.y0 < 0 || q.x1 > J0.g.nbx || q.y1 > J0.g.nby) { ok = false; break; }
            batch.rect_first[r] = (int)total;
            batch.rect_inv[r] = w > 1 ? (unsigned)((0x100000000ull + (unsigned)w - 1) / (unsigned)w) : 0u;
            total += (long)w * h;
        }
        // (total > 0: workgroup 0 of job 0 also stores the frame's tile table for the later launches)
        if (ok && total > 0 && total * 4 <= (long)J0.g.nbx * J0.g.nby * 3 && total < (1l << 24)) {
            batch.compact0 = (int)total;
            first_blocks = (int)((total + 7) & ~7l);
            g_compact_launches++;
        }
    }
    const int nblocks = first_blocks + upper_blocks;
    batch.upper_groups = upper_blocks / 8; batch.total_groups = nblocks / 8;
    // measured on MI355X (tools/ab.sh, cfg-A fp32): upper levels dealt between the level-0 groups 156 us per launch, upper
    // levels last 151 us -- a latency-bound workgroup in a slot costs the level-0 phase more than the tail costs
    batch.sequential = 1;
    if (kExp) {
        static const bool interleave = getenv("PF_INTERLEAVE_JOBS") != nullptr;
        static const bool upper_first = getenv("PF_UPPER_FIRST") != nullptr;
        batch.sequential = upper_first ? 2 : !interleave;
        static const int rect_runs = getenv("PF_RECT_ORDER") ? atoi(getenv("PF_RECT_ORDER")) : 0;      // 1: XCD runs even with rectangles; 2: round robin always (A/B)
        batch.rect_runs = rect_runs;
    }
    if (batch.njobs && batch.job[0].from_warp && batch.job[0].g.level == 0) batch.need_r0 = level0_need_reach(lay, batch.tab0_n, batch.job[0].nrect);
#if PF_EXPERIMENTS
    // A/B (PF_FORCE_NEED_TEST=1, with PF_CULL=0): every level-0 block runs its need test although nothing is culled -- the test is then always
    // true and the work the same as without it: what the test costs a launch, i.e. the most a host-side level-0 bitmap could save (profiles/r06_ab.md)
    static const bool force_test = getenv("PF_FORCE_NEED_TEST") != nullptr;
    if (force_test && batch.njobs && batch.job[0].from_warp && batch.job[0].g.level == 0 && batch.tab0_n > 0 && batch.job[0].nrect == 0 && !batch.need_r0) {
        batch.need_r0 = level0_need_reach(lay, batch.tab0_n, 1);
        if (batch.need_r0) { batch.job[0].nrect = 1; batch.rect0[0] = BlockRect{ 0, 0, 32767, 32767 }; }
    }
#endif
    if (!batch.njobs) return;
    FusedWarp w{};
    if (wa) {
        for (int i = 0; i < 9; i++) w.M[i] = wa->M[i];
        w.total = frame_bytes(wa->srows, wa->scols, wa->sstep, wa->src_cn); w.wmap = wa->wmap;
        w.srows = wa->srows; w.scols = wa->scols; w.sstep = (int)wa->sstep; w.cn = wa->src_cn;
        w.plain = plain_homography(*wa);
        w.wxc = wa->xc; w.wyc = wa->yc; w.wdmax = wa->dis_max; w.wrcp = (float)(1.0L / (long double)wa->dis_max); w.wtype = wa->weight_type;
        w.seed_ok = (w.plain && batch.job[0].from_warp) ? seed_plan(wa->M, batch.job[0].g.rows, batch.job[0].g.cols) : 0;
    }
    if (batch.tab0_n) g_form_counts[7]++;
    // The radial weight is computed in the kernel (radial_weight; two gathers per warped pixel instead of three: fp32 +1.8 %, int16 +-0
    // against the weight plane gather, profiles/r03_ab.md) unless the frame is too large for its exactness argument -- those frames, and
    // PF_WEIGHT_PLANE=1 (tests), gather the reference's weightImage plane as fused = 0/2/3 do.
    static const bool wplane_env = exp_env("PF_WEIGHT_PLANE") != nullptr;      // experiments library
    const bool wplane = wplane_env || (wa && !radial_weight_exact(*wa));
    unsigned long long* st = nullptr;
    // Stage A of a level-0 block (level3_block): fp32 -- every row's coordinates, weight and row loads first, every bilinear sum last, the
    // fractions and the weight parked in the pixel's own LDS slot meanwhile (ILP 0; blocks that do not map strictly inside
    // the frame, and frames that gather the weight plane, walk two / three rows per step as before).  int16 -- two rows per step (ILP 2): that form
    // fits 64 VGPRs, so four workgroups share a CU; the deferred form spills there and is 10 % slower.  Measured, same box, interleaved
    // (profiles/r05_ab.md): fp32 +5 % over round 4's three rows per step once the prologue's scalar loads are batched, int16 +4 % from the
    // prologue alone.
#if PF_EXPERIMENTS
    static const bool stamp = getenv("PF_STAMP") != nullptr;
    static const int ilp_env = getenv("PF_A_ILP") ? atoi(getenv("PF_A_ILP")) : -1;       // 0: deferred finishes; 2 / 3: rows per step
    if (stamp && BH == 32) {
        if (!g_stamp_buf) { if (hipMalloc((void**)&g_stamp_buf, kStampBlocks * 64) != hipSuccess) g_stamp_buf = nullptr; }
        if (g_stamp_buf && nblocks <= kStampBlocks) {
            (void)hipMemsetAsync(g_stamp_buf, 0, (size_t)kStampBlocks * 64, s);
            g_stamp_blocks = nblocks;
            st = g_stamp_buf;
        }
    }
#define PF_GO(F, H, T, S, I, P, W) hipLaunchKernelGGL((k_levels<F, H, T, S, I, P, W>), dim3(nblocks), dim3(T), 0, s, batch, w, src, st)
    if (st) {                                          // stamped instantiations (tools/stamp_phases.py, tools/count_wins.py) of the product's forms, or of the deferred one
        g_form_counts[6]++;
        if (wplane) { if (lay.f32) PF_GO(true, 32, 512, true, 3, false, false); else PF_GO(false, 32, 512, true, 2, false, false); }
        else        { if (lay.f32) PF_GO(true, 32, 512, true, 0, false, true);  else PF_GO(false, 32, 512, true, 2, false, true); }
        return;
    }
    // level 0 with LDS-staged source patches when the frame's map allows it (patch_plan)
    if (wa && batch.job[0].from_warp && patch_plan(*wa, batch.job[0].g.rows, batch.job[0].g.cols, BH, w)) {
        g_form_counts[2]++;
        if (lay.f32) PF_GO(true, 32, 512, false, 2, true, false); else PF_GO(false, 32, 512, false, 2, true, false);
        return;
    }
    if (BH == 24) {                                   // PF_BLOCK24 (A/B, profiles/r06_ab.md): 64 x 24 blocks, 40 928 B of LDS and 64 VGPRs -> four workgroups per CU for fp32 too
        if (wa) g_form_counts[5]++;
        if (lay.f32) PF_GO(true, 24, 512, false, 0, false, true); else PF_GO(false, 24, 512, false, 2, false, true);
        return;
    }
    if (BH == 28) {                                   // PF_BLOCK28 (A/B): 64x28 blocks stage 35 rows = five whole passes of the 7 rows 512 threads warp at a time
        if (wa) g_form_counts[5]++;
        if (wplane) { if (lay.f32) PF_GO(true, 28, 512, false, 3, false, false); else PF_GO(false, 28, 512, false, 2, false, false); }
        else        { if (lay.f32) PF_GO(true, 28, 512, false, 3, false, true);  else PF_GO(false, 28, 512, false, 2, false, true); }
        return;
    }
    if (BH == 64) {                                   // PF_BLOCK64 (A/B, see level_block_rows)
        if (wa) g_form_counts[4]++;
        if (lay.f32) PF_GO(true, 64, 1024, false, 3, false, false); else PF_GO(false, 64, 1024, false, 2, false, false);
        return;
    }
    if (ilp_env == 0 && !wplane && !lay.f32) {        // int16 with deferred finishes
        if (wa) g_form_counts[0]++;
        PF_GO(false, 32, 512, false, 0, false, true);
        return;
    }
    if (ilp_env == 2 || ilp_env == 3) {               // the row loops of rounds 2-4
        if (wa) g_form_counts[wplane ? 1 : 0]++;
        if (wplane) {
            if (lay.f32) { if (ilp_env == 3) PF_GO(true, 32, 512, false, 3, false, false); else PF_GO(true, 32, 512, false, 2, false, false); }
            else         { if (ilp_env == 3) PF_GO(false, 32, 512, false, 3, false, false); else PF_GO(false, 32, 512, false, 2, false, false); }
        } else {
            if (lay.f32) { if (ilp_env == 3) PF_GO(true, 32, 512, false, 3, false, true); else PF_GO(true, 32, 512, false, 2, false, true); }
            else         { if (ilp_env == 3) PF_GO(false, 32, 512, false, 3, false, true); else PF_GO(false, 32, 512, false, 2, false, true); }
        }
        return;
    }
#undef PF_GO
#endif
    // the product's forms: weight computed or gathered
    if (wa) g_form_counts[wplane ? 1 : 0]++;
    if (wplane) {
        if (lay.f32) hipLaunchKernelGGL((k_levels<true, 32, 512, false, 3, false, false>), dim3(nblocks), dim3(512), 0, s, batch, w, src, st);
        else         hipLaunchKernelGGL((k_levels<false, 32, 512, false, 2, false, false>), dim3(nblocks), dim3(512), 0, s, batch, w, src, st);
    } else {
        if (lay.f32) hipLaunchKernelGGL((k_levels<true, 32, 512, false, 0, false, true>), dim3(nblocks), dim3(512), 0, s, batch, w, src, st);
        else         hipLaunchKernelGGL((k_levels<false, 32, 512, false, 2, false, true>), dim3(nblocks), dim3(512), 0, s, batch, w, src, st);
    }
}

// rows of a level-kernel block: 32 (512 threads, three workgroups per CU).  PF_BLOCK64=1 (A/B): 64 rows x 1024 threads, halo
// recompute 1.23 instead of 1.35 -- int16 (12-byte LDS pixels) fits two such workgroups per CU by LDS, fp32 (96 KB) one.
// Measured on MI355X, cfg-A: int16 -23 %, fp32 -18 %
int level_block_rows(bool f32)
{
    (void)f32;
#if PF_EXPERIMENTS
    static const bool b64 = getenv("PF_BLOCK64") != nullptr, b28 = getenv("PF_BLOCK28") != nullptr, b24 = getenv("PF_BLOCK24") != nullptr;
    return b64 ? 64 : (b28 ? 28 : (b24 ? 24 : 32));
#else
    return 32;
#endif
}

// diagnostics (PF_STAMP=1): pixels seen / won by stage D per level since the last reset; out[2 * level] = seen, [2 * level + 1] = won
int read_select_counts(unsigned long long* out, int reset)
{
    unsigned long long seen[kMaxLevels], won[kMaxLevels];
    if (hipMemcpyFromSymbol(seen, HIP_SYMBOL(g_select_seen), sizeof seen) != hipSuccess) return 0;
    if (hipMemcpyFromSymbol(won, HIP_SYMBOL(g_select_won), sizeof won) != hipSuccess) return 0;
    for (int i = 0; i < kMaxLevels; i++) { out[2 * i] = seen[i]; out[2 * i + 1] = won[i]; }
    if (reset) {
        unsigned long long z[kMaxLevels] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_select_seen), z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_select_won), z, sizeof z);
    }
    return kMaxLevels;
}

// diagnostics: the stamps of the most recent PF_STAMP launch (8 u64 per workgroup: start, A done, barrier 1 passed, B done,
// barrier 2 passed, D done + stores drained, job index, XCC id); returns the workgroup count
int read_phase_stamps(unsigned long long* out, int cap_blocks)
{
    if (!g_stamp_buf || !g_stamp_blocks) return 0;
    const int n = g_stamp_blocks < cap_blocks ? g_stamp_blocks : cap_blocks;
    if (out && hipMemcpy(out, g_stamp_buf, (size_t)n * 64, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------ blend
// strip-set geometry of a neighbour at (dx,dy): level i holds h_i x w_i pixels
__host__ __device__ inline void strip_dims(int nlev, int level, int dx, int dy, int& w, int& h)
{
    const int ts = kElePixels >> level, b = 1 << (nlev - 1 - level);
    w = dx == 0 ? ts : b; h = dy == 0 ? ts : b;
}

size_t halo_bytes(const TileLayout& lay, int dx, int dy)
{
    size_t n = 0;
    for (int i = 0; i < lay.nlev; i++) { int w, h; strip_dims(lay.nlev, i, dx, dy, w, h); n += (size_t)w * h; }
    return n * 3 * (lay.f32 ? 4 : 2);
}

#if PF_EXPERIMENTS      // the per-level output side of rounds 1-5: A/B partner and second opinion of collapse_fused.hip (PF_BLEND_PER_LEVEL=1)
// Ele::blend's 3x3 assembly (.cpp:93-117): padded level image of side ts+2b.
template <bool F32>
__global__ __launch_bounds__(256) void k_blend_gather(TileLayout lay, int level, int border, const BlendSrc* __restrict__ srcs,
                                                       char* __restrict__ dst, size_t dst_stride)
{
    using T = typename Pix<F32>::T;
    const int ts = kElePixels >> level, side = ts + 2 * border;
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= side || py >= side) return;
    const int z = blockIdx.z;
    int rx, sx, ry, sy;
    if (px < border) { rx = 0; sx = ts - border + px; } else if (px < border + ts) { rx = 1; sx = px - border; } else { rx = 2; sx = px - border - ts; }
    if (py < border) { ry = 0; sy = ts - border + py; } else if (py < border + ts) { ry = 1; sy = py - border; } else { ry = 2; sy = py - border - ts; }
    const BlendSrc bs = srcs[z * 9 + ry * 3 + rx];
    T* d = reinterpret_cast<T*>(dst + z * dst_stride) + ((long)py * side + px) * 3;
    const T* s;
    if (!bs.is_strip) {
        s = reinterpret_cast<const T*>(reinterpret_cast<const char*>(bs.base) + lay.lap_off[level]) + ((long)sy * ts + sx) * 3;
    } else {
        // packed strips: levels concatenated, each h x w row-major, origin at the strip's corner
        const int dx = rx - 1, dy = ry - 1;
        long off = 0;
        for (int i = 0; i < level; i++) { int w, h; strip_dims(lay.nlev, i, dx, dy, w, h); off += (long)w * h; }
        int w, h; strip_dims(lay.nlev, level, dx, dy, w, h);
        const int lx = (rx == 0) ? sx - (ts - border) : sx, ly = (ry == 0) ? sy - (ts - border) : sy;
        s = reinterpret_cast<const T*>(bs.base) + (off + (long)ly * w + lx) * 3;
    }
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

void launch_blend_gather(hipStream_t s, const TileLayout& lay, int level, int border, const BlendSrc* srcs,
                         void* dst, size_t dst_stride_bytes, int batch)
{
    const int side = (kElePixels >> level) + 2 * border;
    dim3 grid((side + 63) / 64, (side + 3) / 4, batch), block(256);
    if (lay.f32) hipLaunchKernelGGL(k_blend_gather<true>, grid, block, 0, s, lay, level, border, srcs, (char*)dst, dst_stride_bytes);
    else         hipLaunchKernelGGL(k_blend_gather<false>, grid, block, 0, s, lay, level, border, srcs, (char*)dst, dst_stride_bytes);
}

// restoreImageFromLaplacePyr step: dst = pyrUp(src) + dst  (saturating for 16S)
template <bool F32>
__global__ __launch_bounds__(256) void k_collapse(char* __restrict__ dst, size_t dst_stride, const char* __restrict__ src,
                                                   size_t src_stride, int rows, int cols)
{
    using T = typename Pix<F32>::T; using WT = typename Pix<F32>::WT;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    T* d = reinterpret_cast<T*>(dst + blockIdx.z * dst_stride) + ((long)y * cols + x) * 3;
    const T* s = reinterpret_cast<const T*>(src + blockIdx.z * src_stride);
#pragma unroll
    for (int k = 0; k < 3; k++) d[k] = sat_add(pyr_up_at<T, WT>(s, rows >> 1, cols >> 1, y, x, k), d[k]);
}

void launch_collapse(hipStream_t s, bool f32, void* dst, size_t dst_stride_bytes, const void* src,
                     size_t src_stride_bytes, int rows, int cols, int batch)
{
    dim3 grid((cols + 63) / 64, (rows + 3) / 4, batch), block(256);
    if (f32) hipLaunchKernelGGL(k_collapse<true>, grid, block, 0, s, (char*)dst, dst_stride_bytes, (const char*)src, src_stride_bytes, rows, cols);
    else     hipLaunchKernelGGL(k_collapse<false>, grid, block, 0, s, (char*)dst, dst_stride_bytes, (const char*)src, src_stride_bytes, rows, cols);
}

__device__ __forceinline__ uint8_t sat_uchar(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// crop the centre 256^2, zero where weights[0]==0 (.cpp:121-126,144), optional 8U view (.cpp:156)
template <bool F32>
__global__ __launch_bounds__(256) void k_blend_finish(TileLayout lay, const char* __restrict__ lvl0, size_t stride, int border,
                                                       const BlendSrc* __restrict__ srcs, char* __restrict__ raw, uint8_t* __restrict__ bgr,
                                                       const int* __restrict__ out_idx)
{
    using T = typename Pix<F32>::T;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), z = blockIdx.z;
    const int zo = out_idx ? out_idx[z] : z;                 // where this tile goes in the output (batches are cut by blend mode)
    const int side = kElePixels + 2 * border;
    const T* s = reinterpret_cast<const T*>(lvl0 + z * stride) + ((long)(y + border) * side + x + border) * 3;
    const float* w0 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(srcs[z * 9 + 4].base) + lay.w_off[0]);
    const bool zero = w0[y * kElePixels + x] == 0.f;
    const long o = ((long)zo * kElePixels * kElePixels + y * kElePixels + x) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const T v = zero ? (T)0 : s[k];
        if (raw) reinterpret_cast<T*>(raw)[o + k] = v;
        if (bgr) {
            if constexpr (F32) bgr[o + k] = sat_uchar(__float2int_rn(v * 255.f));
            else bgr[o + k] = sat_uchar(v);
        }
    }
}

void launch_blend_finish(hipStream_t s, const TileLayout& lay, const void* lvl0, size_t stride_bytes, int border,
                         const BlendSrc* srcs, void* raw_out, uint8_t* bgr_out, int batch, const int* out_idx)
{
    dim3 grid(kElePixels / 64, kElePixels / 4, batch), block(256);
    if (lay.f32) hipLaunchKernelGGL(k_blend_finish<true>, grid, block, 0, s, lay, (const char*)lvl0, stride_bytes, border, srcs, (char*)raw_out, bgr_out, out_idx);
    else         hipLaunchKernelGGL(k_blend_finish<false>, grid, block, 0, s, lay, (const char*)lvl0, stride_bytes, border, srcs, (char*)raw_out, bgr_out, out_idx);
}

// ------------------------------------------------------------------- save
template <bool F32>
__global__ __launch_bounds__(256) void k_mosaic_gather(TileLayout lay, int level, const uint64_t* __restrict__ table, int wx, int wy,
                                                        char* __restrict__ dst)
{
    using T = typename Pix<F32>::T;
    const int ts = kElePixels >> level, cols = wx * ts, rows = wy * ts;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint64_t ent = table[(y / ts) * wx + x / ts];
    T* d = reinterpret_cast<T*>(dst) + ((long)y * cols + x) * 3;
    if (!ent) { d[0] = d[1] = d[2] = (T)0; return; }
    const T* s = reinterpret_cast<const T*>(reinterpret_cast<const char*>(ent) + lay.lap_off[level]) + ((long)(y % ts) * ts + x % ts) * 3;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

void launch_mosaic_gather(hipStream_t s, const TileLayout& lay, int level, const uint64_t* table, int wx, int wy, void* dst)
{
    const int ts = kElePixels >> level;
    dim3 grid((wx * ts + 63) / 64, (wy * ts + 3) / 4), block(256);
    if (lay.f32) hipLaunchKernelGGL(k_mosaic_gather<true>, grid, block, 0, s, lay, level, table, wx, wy, (char*)dst);
    else         hipLaunchKernelGGL(k_mosaic_gather<false>, grid, block, 0, s, lay, level, table, wx, wy, (char*)dst);
}

// 16S -> 8U saturate, background where level-0 weight == 0 (.cpp:838-840)
template <bool F32>
__global__ __launch_bounds__(256) void k_save_finish(TileLayout lay, const char* __restrict__ lvl0, const uint64_t* __restrict__ table,
                                                      int wx, int wy, int bg, uint8_t* __restrict__ bgr)
{
    using T = typename Pix<F32>::T;
    const int cols = wx * kElePixels, rows = wy * kElePixels;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint64_t ent = table[(y >> 8) * wx + (x >> 8)];
    float w = 0.f;
    if (ent) w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ent) + lay.w_off[0])[(y & 255) * kElePixels + (x & 255)];
    const long o = ((long)y * cols + x) * 3;
    const T* s = reinterpret_cast<const T*>(lvl0) + o;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint8_t v;
        if (w == 0.f) v = sat_uchar(bg);
        else if constexpr (F32) v = sat_uchar(__float2int_rn(s[k] * 255.f));
        else v = sat_uchar(s[k]);
        bgr[o + k] = v;
    }
}

void launch_save_finish(hipStream_t s, const TileLayout& lay, const void* lvl0, const uint64_t* table, int wx, int wy,
                        int bg, uint8_t* bgr)
{
    dim3 grid((wx * kElePixels + 63) / 64, (wy * kElePixels + 3) / 4), block(256);
    if (lay.f32) hipLaunchKernelGGL(k_save_finish<true>, grid, block, 0, s, lay, (const char*)lvl0, table, wx, wy, bg, bgr);
    else         hipLaunchKernelGGL(k_save_finish<false>, grid, block, 0, s, lay, (const char*)lvl0, table, wx, wy, bg, bgr);
}

#endif  // PF_EXPERIMENTS

// ------------------------------------------------------------- halo pack
template <bool F32>
__global__ __launch_bounds__(256) void k_halo_pack(TileLayout lay, const char* __restrict__ slot, int dx, int dy, char* __restrict__ out)
{
    using T = typename Pix<F32>::T;
    // one launch covers all levels: blockIdx.y = level
    const int level = blockIdx.y;
    const int ts = kElePixels >> level, b = 1 << (lay.nlev - 1 - level);
    int w, h; strip_dims(lay.nlev, level, dx, dy, w, h);
    long off = 0;
    for (int i = 0; i < level; i++) { int ww, hh; strip_dims(lay.nlev, i, dx, dy, ww, hh); off += (long)ww * hh; }
    // the neighbour at (dx,dy) of the requesting tile hands over the edge facing it:
    // dx=-1 (left neighbour) -> its right-most b columns, dx=+1 -> its left-most b columns
    const int sx0 = dx < 0 ? ts - b : 0, sy0 = dy < 0 ? ts - b : 0;
    const T* s = reinterpret_cast<const T*>(slot + lay.lap_off[level]);
    T* d = reinterpret_cast<T*>(out) + off * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int ly = i / w, lx = i - ly * w;
        const long sp = ((long)(sy0 + ly) * ts + sx0 + lx) * 3;
        d[(long)i * 3] = s[sp]; d[(long)i * 3 + 1] = s[sp + 1]; d[(long)i * 3 + 2] = s[sp + 2];
    }
}

void launch_halo_pack(hipStream_t s, const TileLayout& lay, const void* slot, int dx, int dy, void* out)
{
    dim3 grid(32, lay.nlev), block(256);
    if (lay.f32) hipLaunchKernelGGL(k_halo_pack<true>, grid, block, 0, s, lay, (const char*)slot, dx, dy, (char*)out);
    else         hipLaunchKernelGGL(k_halo_pack<false>, grid, block, 0, s, lay, (const char*)slot, dx, dy, (char*)out);
}

// every strip set of an exchange in one launch: blockIdx.z = strip set, blockIdx.y = level
template <bool F32>
__global__ __launch_bounds__(256) void k_halo_pack_batch(TileLayout lay, const StripDesc* __restrict__ descs, char* __restrict__ out)
{
    using T = typename Pix<F32>::T;
    const StripDesc d = descs[blockIdx.z];
    const int level = blockIdx.y, dx = d.dx, dy = d.dy;
    const int ts = kElePixels >> level, b = 1 << (lay.nlev - 1 - level);
    int w, h; strip_dims(lay.nlev, level, dx, dy, w, h);
    long off = 0;
    for (int i = 0; i < level; i++) { int ww, hh; strip_dims(lay.nlev, i, dx, dy, ww, hh); off += (long)ww * hh; }
    const int sx0 = dx < 0 ? ts - b : 0, sy0 = dy < 0 ? ts - b : 0;
    const T* s = reinterpret_cast<const T*>(reinterpret_cast<const char*>(d.slot) + lay.lap_off[level]);
    T* o = reinterpret_cast<T*>(out + d.out_off) + off * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int ly = i / w, lx = i - ly * w;
        const long sp = ((long)(sy0 + ly) * ts + sx0 + lx) * 3;
        o[(long)i * 3] = s[sp]; o[(long)i * 3 + 1] = s[sp + 1]; o[(long)i * 3 + 2] = s[sp + 2];
    }
}

void launch_halo_pack_batch(hipStream_t s, const TileLayout& lay, const StripDesc* descs_dev, int n, void* out)
{
    if (n <= 0) return;
    for (int z0 = 0; z0 < n; z0 += 65535) {                    // gridDim.z limit
        const int nz = n - z0 < 65535 ? n - z0 : 65535;
        dim3 grid(8, lay.nlev, nz), block(256);
        if (lay.f32) hipLaunchKernelGGL(k_halo_pack_batch<true>, grid, block, 0, s, lay, descs_dev + z0, (char*)out);
        else         hipLaunchKernelGGL(k_halo_pack_batch<false>, grid, block, 0, s, lay, descs_dev + z0, (char*)out);
    }
}

// ------------------------------------------------------------------ misc
TileLayout make_layout(int band_num, bool f32)
{
    TileLayout l{};
    l.nlev = band_num + 1; l.f32 = f32 ? 1 : 0;
    uint32_t off = 0;
    const int es = f32 ? 4 : 2;
    for (int i = 0; i < l.nlev; i++) { l.lap_off[i] = off; const uint32_t n = (kElePixels >> i) * (kElePixels >> i); off += ((n * 3 * es + 255) / 256) * 256; }
    for (int i = 0; i < l.nlev; i++) { l.w_off[i] = off;   const uint32_t n = (kElePixels >> i) * (kElePixels >> i); off += ((n * 4 + 255) / 256) * 256; }
    l.slot_bytes = off;
    return l;
}

const char* kernel_name(int id)
{
    static const char* n[K_COUNT] = { "warp", "pyrdown_img", "pyrdown_w", "lap_select", "blend_gather", "collapse",
                                      "blend_finish", "mosaic_gather", "save_finish", "level0_fused", "level_fused", "single_band", "blend_fused", "save_fused" };
    return (id >= 0 && id < K_COUNT) ? n[id] : "?";
}

}  // namespace pf
