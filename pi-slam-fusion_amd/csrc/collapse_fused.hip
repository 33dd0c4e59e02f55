// collapse_fused.hip -- the output side of the path as ONE launch per request (gfx950, wave64):
//   Ele::blend        MultiBandMap2DCPU.cpp:77-146   3x3 assembly (:93-117), restoreImageFromLaplacePyr (:119, :142),
//                                                     crop + weights[0]==0 mask (:121-126, :144)
//   updateTexture     :149-188                        the 8U view (convertTo CV_8UC3, :156)
//   save              :806-840                        paste per level, one whole-mosaic collapse, 8U, background
//
// Rounds 1-5 ran this as the reference writes it -- a padded square per level in HBM (k_blend_gather), one k_collapse launch
// per level, a finish launch: 12 launches per chunk and every level through HBM twice.  Here a workgroup owns a 128 x 32
// block of the level-0 result (a row segment of 384 bytes of BGR8 = three whole cache lines) and collapses the part of the
// pyramid that block depends on inside LDS:
//
//   pyrUp is a 3-tap filter, so level i-1 rows [lo, hi] need level i rows [(lo-1)>>1, (hi>>1)+1]: one pixel of halo per
//   level (66 x 18, 35 x 11, 20 x 8, 12 x 6, 8 x 5 ... pixels for levels 1, 2, ...: 23 KB of LDS for any band count).
//   1. the Laplacian regions of levels 1..L go from the tile slots / packed halo strips / mosaic tiles straight into LDS;
//   2. levels L-1 ... 1 are restored in place in LDS, one thread per 2 x 2 destination quad (pyrUp + add in the reference's
//      operation order, with its C cast and saturation and the borders of the padded square / the mosaic -- the edge forms
//      of pyrUp are not index reflections in fp32);
//   3. level 0: a thread takes 2 rows x 4 columns, reads its 3 x 4 neighbourhood of level 1 from LDS once, adds the tile's
//      own Laplacian (16-byte loads), applies the weight mask and the 8U view and writes 12 bytes per row.
// Nothing of a level >= 1 ever returns to HBM, level 0 is read once and the result written once.
//
// Bit-exactness: compiled with -ffp-contract=off; every sum below is written in the association order of OpenCV 2.4.9's
// pyrUp_.  One rewriting is used: pyrUp's odd-column sum (s[x] + s[x+1]) * 4 is carried as O = s[x] + s[x+1]
// and the factor 4 folded into the final cast -- multiplying by a power of two commutes with fp32 rounding (no overflow or
// underflow is reachable: the operands are sums of pixel values), and for 16S ((4 v + 32) >> 6) == ((v + 8) >> 4) exactly.
// Checked against the oracle by the blend / save / dist tests and against the per-level kernels of rounds 1-5 (experiments library).
#include "kernels.hpp"
#include "warp_index.hpp"

namespace pf {
namespace {

#define PF_GLOBAL __attribute__((address_space(1)))

constexpr int kBW = 128, kBH = 32;                    // level-0 block of a workgroup
constexpr int kCT = 256;                              // threads
#ifndef PF_CF_WAVES
#define PF_CF_WAVES 5
#endif
#ifndef PF_CF_ABLATE         // timing-only A/B builds (tools/build_variant.sh): 1 no region loads, 2 no restore of levels L-1..1, 4 no level-1 sums in the level-0 pass
#define PF_CF_ABLATE 0
#endif
constexpr int kCFWaves = PF_CF_WAVES;                 // waves per SIMD the register budget is cut for (5: 96 VGPRs, five workgroups per CU; 6 spilled and was 30 % slower, profiles/r06_blend_ab.md)
constexpr int region_edge(int s, int level) { for (int i = 0; i < level; i++) s = ((s + 1) >> 1) + 2; return s; }
constexpr int region_px(int level) { return region_edge(kBW, level) * region_edge(kBH, level); }
constexpr int region_px_total(int from) { int n = 0; for (int i = from; i < kMaxLevels; i++) n += region_px(i); return n; }
constexpr int kLdsPx = region_px_total(1);            // 1925 pixels = 23 100 B of 3 x 4-byte components
static_assert(region_edge(128, 1) == 66 && region_edge(32, 1) == 18 && region_edge(66, 1) == 35, "pyrUp dependence regions");

typedef float    f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(4)));

template <bool F32> struct Px;
template <> struct Px<false> { using T = short; using WT = int; };
template <> struct Px<true>  { using T = float; using WT = float; };

// pyrUp_'s vertical step + cast for the four parities of a destination pixel.  E = even-column horizontal sum, O = odd-column sum / 4
// (see the header); rows 0 / 1 / 2 = source rows sy-1 / sy / sy+1.
__device__ __forceinline__ int   up_ee(int e0, int e1, int e2)       { return (int)(short)((e0 + e1 * 6 + e2 + 32) >> 6); }   // (short) C cast: wraps
__device__ __forceinline__ int   up_eo(int o0, int o1, int o2)       { return (int)(short)((o0 + o1 * 6 + o2 + 8) >> 4); }
__device__ __forceinline__ int   up_oe(int e1, int e2)               { return (int)(short)((e1 + e2 + 8) >> 4); }
__device__ __forceinline__ int   up_oo(int o1, int o2)               { return (int)(short)((o1 + o2 + 2) >> 2); }
__device__ __forceinline__ float up_ee(float e0, float e1, float e2) { return (e0 + e1 * 6 + e2) * (1.f / 64); }
__device__ __forceinline__ float up_eo(float o0, float o1, float o2) { return (o0 + o1 * 6 + o2) * (1.f / 16); }
__device__ __forceinline__ float up_oe(float e1, float e2)           { return (e1 + e2) * (1.f / 16); }
__device__ __forceinline__ float up_oo(float o1, float o2)           { return (o1 + o2) * (1.f / 4); }
__device__ __forceinline__ int   add_sat(int up, int lap)     { return sat_short(up + lap); }  // cv::add on 16S saturates
__device__ __forceinline__ float add_sat(float up, float lap) { return up + lap; }
__device__ __forceinline__ uint32_t sat_u8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// region of a level held in LDS: rows [y0, y0 + h) x cols [x0, x0 + w) of the level's image (rows x cols); the regions of levels
// 1, 2, ... lie back to back, pixel p of the flat list at lds[3 p]
struct Reg {
    int y0, x0, h, w;
    int poff, rows, cols;                // first pixel in the flat list; extent of the level's image
};
// what the flat loader needs of it, per level, in LDS
struct RegL { int poff, y0, x0, w; float rcp_w; int lap_off, pad0_, pad1_; };

struct Shared {
    RegL reg[kMaxLevels];
    BlendJob job;
};

// the recurrence above from the level-0 block; every input is workgroup-uniform, so this is scalar code
__device__ __forceinline__ Reg level_region(int level, int Y0, int X0, int rows0, int cols0)
{
    int ylo = Y0, yhi = Y0 + kBH - 1, xlo = X0, xhi = X0 + kBW - 1, poff = 0;
    Reg r{};
    for (int i = 1; i <= level; i++) {
        const int rows = rows0 >> i, cols = cols0 >> i;
        ylo = (ylo - 1) >> 1; if (ylo < 0) ylo = 0;
        xlo = (xlo - 1) >> 1; if (xlo < 0) xlo = 0;
        yhi = (yhi >> 1) + 1; if (yhi > rows - 1) yhi = rows - 1;
        xhi = (xhi >> 1) + 1; if (xhi > cols - 1) xhi = cols - 1;
        r.y0 = ylo; r.x0 = xlo; r.h = yhi - ylo + 1; r.w = xhi - xlo + 1; r.poff = poff; r.rows = rows; r.cols = cols;
        poff += r.h * r.w;
    }
    return r;
}

// idx / w for 0 <= idx < 4096, 1 <= w <= 128, rcp = 1.f / w: (idx + 0.5) / w is at least 1 / 256 away from an integer, the float error is below 2^-10
__device__ __forceinline__ int div_small(int idx, float rcp) { return (int)(((float)idx + 0.5f) * rcp); }

// three components of a pixel.  int16 pixels (6 bytes, 2-byte aligned) of a TILE SLOT are read as one 8-byte load: the two bytes behind a
// pixel are the next pixel's or the level's alignment padding inside the slot (levels are 256-byte aligned, the weights follow the last one).
// Packed halo strips keep three 2-byte loads: their last pixel may be the last bytes of the exchange buffer.
template <bool F32, bool SLOT>
__device__ __forceinline__ void load_px(const PF_GLOBAL typename Px<F32>::T* s, typename Px<F32>::WT out[3])
{
    using WT = typename Px<F32>::WT;
    if constexpr (!F32 && SLOT) {
        typedef uint32_t u2u __attribute__((ext_vector_type(2), aligned(1)));
        const u2u v = *(const PF_GLOBAL u2u*)s;
        out[0] = (int)(short)(v.x & 0xffffu); out[1] = (int)v.x >> 16; out[2] = (int)(short)(v.y & 0xffffu);
    } else { out[0] = (WT)s[0]; out[1] = (WT)s[1]; out[2] = (WT)s[2]; }
}

__device__ __forceinline__ void strip_dims_d(int nlev, int level, int dx, int dy, int& w, int& h)
{
    const int ts = kElePixels >> level, b = 1 << (nlev - 1 - level);
    w = dx == 0 ? ts : b; h = dy == 0 ? ts : b;
}

// Laplacian pixel (py, px) of level `level` of a tile's padded square (Ele::blend's assembly, .cpp:93-117)
template <bool F32>
__device__ __forceinline__ void fetch_blend(const BlendJob& job, int nlev, int level, int lap_off, int py, int px, typename Px<F32>::WT out[3])
{
    using T = typename Px<F32>::T;
    const int ts = kElePixels >> level, b = job.border ? 1 << (nlev - 1 - level) : 0;
    int rx = 1, sx = px - b, ry = 1, sy = py - b;
    if (sx < 0) { rx = 0; sx += ts; } else if (sx >= ts) { rx = 2; sx -= ts; }
    if (sy < 0) { ry = 0; sy += ts; } else if (sy >= ts) { ry = 2; sy -= ts; }
    const int j = ry * 3 + rx;
    const PF_GLOBAL T* s;
    if (!((job.strip_mask >> j) & 1)) {
        s = (const PF_GLOBAL T*)((const PF_GLOBAL char*)job.src[j] + lap_off) + (sy * ts + sx) * 3;
        load_px<F32, true>(s, out);
        return;
    } else {                                              // packed strips: levels concatenated, each h x w row-major from the strip's corner
        const int dx = rx - 1, dy = ry - 1;
        int off = 0;
        for (int i = 0; i < level; i++) { int w, h; strip_dims_d(nlev, i, dx, dy, w, h); off += w * h; }
        int w, h; strip_dims_d(nlev, level, dx, dy, w, h);
        const int lx = rx == 0 ? sx - (ts - b) : sx, ly = ry == 0 ? sy - (ts - b) : sy;
        s = (const PF_GLOBAL T*)job.src[j] + (off + ly * w + lx) * 3;
    }
    load_px<F32, false>(s, out);
}

// ... of the pasted mosaic (save, .cpp:806-834): absent tiles are zero
template <bool F32>
__device__ __forceinline__ void fetch_mosaic(const uint64_t* __restrict__ table, int wx, int level, int lap_off, int py, int px, typename Px<F32>::WT out[3])
{
    using T = typename Px<F32>::T; using WT = typename Px<F32>::WT;
    const int sh = 8 - level, ts = kElePixels >> level;
    const uint64_t ent = table[(py >> sh) * wx + (px >> sh)];
    if (!ent) { out[0] = out[1] = out[2] = (WT)0; return; }
    const PF_GLOBAL T* s = (const PF_GLOBAL T*)((const PF_GLOBAL char*)ent + lap_off) + ((py & (ts - 1)) * ts + (px & (ts - 1))) * 3;
    load_px<F32, true>(s, out);
}

// 4 consecutive level-0 pixels of a row (12 components): loads and stores by the widest aligned pieces
template <bool F32> struct Row4;
template <> struct Row4<true> {
    float v[12];
    __device__ __forceinline__ void load(const PF_GLOBAL float* p) {
        const PF_GLOBAL f4* q = (const PF_GLOBAL f4*)p;
        const f4 a = q[0], b = q[1], c = q[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    }
    __device__ __forceinline__ float in(int e) const { return v[e]; }
    __device__ __forceinline__ void store(float* p) const {
        f4* q = (f4*)p;
        q[0] = f4{ v[0], v[1], v[2], v[3] }; q[1] = f4{ v[4], v[5], v[6], v[7] }; q[2] = f4{ v[8], v[9], v[10], v[11] };
    }
};
template <> struct Row4<false> {
    int v[12];
    uint32_t u[6];                                                     // as loaded: unpacked where the values are first used (12 fewer live registers while the level-1 sums are formed)
    __device__ __forceinline__ void load(const PF_GLOBAL short* p) {
        const PF_GLOBAL u2* q = (const PF_GLOBAL u2*)p;                // 24 bytes, 8-byte aligned
        const u2 a = q[0], b = q[1], c = q[2];
        u[0] = a.x; u[1] = a.y; u[2] = b.x; u[3] = b.y; u[4] = c.x; u[5] = c.y;
    }
    __device__ __forceinline__ int in(int e) const { return (e & 1) ? (int)u[e >> 1] >> 16 : (int)(short)(u[e >> 1] & 0xffffu); }
    __device__ __forceinline__ void store(short* p) const {
        uint32_t w[6];
#pragma unroll
        for (int i = 0; i < 6; i++) w[i] = ((uint32_t)v[2 * i] & 0xffffu) | ((uint32_t)v[2 * i + 1] << 16);
        u2* q = (u2*)p;
        q[0] = u2{ w[0], w[1] }; q[1] = u2{ w[2], w[3] }; q[2] = u2{ w[4], w[5] };
    }
};

// One workgroup = one 128 x 32 block of a level-0 result.
//   MOSAIC = false: block (blockIdx % 16) of tile job[blockIdx / 16] (Ele::blend); results to raw / bgr at tile index job.out
//   MOSAIC = true : block of the wx*2 x wy*8 block grid of the pasted mosaic (save); result to bgr (rows x cols x 3)
// Workgroup ids are dealt so that the blocks of one tile / of neighbouring tiles run on one XCD (they share the upper levels in its L2).
template <bool F32, bool MOSAIC>
__global__ __launch_bounds__(kCT, kCFWaves) void k_collapse_fused(TileLayout lay, const BlendJob* __restrict__ jobs, const uint64_t* __restrict__ table,
                                                                   int wx, int wy, int bg, char* __restrict__ raw, uint8_t* __restrict__ bgr, int nblocks)
{
    using T = typename Px<F32>::T; using WT = typename Px<F32>::WT;
    __shared__ WT lds[kLdsPx * 3];
    __shared__ Shared sh;
    const int tid = threadIdx.x, L = lay.nlev - 1;
    // XCD-aware order: hardware deals consecutive workgroup ids round robin over the 8 XCDs; give XCD k the k-th eighth of the logical blocks
    int wg = blockIdx.x;
    if ((nblocks & 7) == 0) wg = (wg & 7) * (nblocks >> 3) + (wg >> 3);

    int Y0, X0, rows0, cols0;                         // block origin and extent of the (padded) level-0 image
    const PF_GLOBAL char* self = nullptr;             // the tile whose level 0 this block restores
    int ly0, lx0;                                      // block origin inside that tile
    int out_tile = 0;
    if constexpr (MOSAIC) {
        const int nbx = wx * (kElePixels / kBW);
        const int by = wg / nbx, bx = wg - by * nbx;
        Y0 = by * kBH; X0 = bx * kBW; rows0 = wy * kElePixels; cols0 = wx * kElePixels;
        self = (const PF_GLOBAL char*)table[(Y0 >> 8) * wx + (X0 >> 8)];
        ly0 = Y0 & 255; lx0 = X0 & 255;
        if (!self) {                                   // no tile here: background (.cpp:840, weights never pasted stay 0)
            const uint32_t b8 = sat_u8(bg), word = b8 * 0x01010101u;
            constexpr int kRowWords = kBW * 3 / 4;
            for (int t = tid; t < kBH * kRowWords; t += kCT) {
                const int r = t / kRowWords, c = t - r * kRowWords;
                ((uint32_t*)(bgr + ((size_t)(Y0 + r) * cols0 + X0) * 3))[c] = word;
            }
            return;
        }
    } else {
        constexpr int kPerTile = (kElePixels / kBW) * (kElePixels / kBH);
        const int z = wg / kPerTile, blk = wg - z * kPerTile;
        if (tid < (int)(sizeof(BlendJob) / 4)) ((uint32_t*)&sh.job)[tid] = ((const uint32_t*)(jobs + z))[tid];
        const int b0 = jobs[z].border ? 1 << L : 0;
        ly0 = (blk / (kElePixels / kBW)) * kBH; lx0 = (blk % (kElePixels / kBW)) * kBW;
        Y0 = b0 + ly0; X0 = b0 + lx0; rows0 = cols0 = kElePixels + 2 * b0;
        self = (const PF_GLOBAL char*)jobs[z].src[4];
        out_tile = jobs[z].out;
    }

    // ---- regions of levels 1..L this block depends on (thread i: level i, for the flat loader)
    if (tid >= 1 && tid <= L) {
        const Reg r = level_region(tid, Y0, X0, rows0, cols0);
        RegL q; q.poff = r.poff; q.y0 = r.y0; q.x0 = r.x0; q.w = r.w; q.rcp_w = 1.f / (float)r.w; q.lap_off = (int)lay.lap_off[tid]; q.pad0_ = q.pad1_ = 0;
        sh.reg[tid] = q;
    }
    const Reg r1 = L >= 1 ? level_region(1, Y0, X0, rows0, cols0) : Reg{};
    const Reg rL = L >= 1 ? level_region(L, Y0, X0, rows0, cols0) : Reg{};
    const int total = L >= 1 ? rL.poff + rL.h * rL.w : 0;
    __syncthreads();

    // Roles of phases 1 and 2 are dealt to the waves in an order that turns with the workgroup: the restore of the small top levels is
    // work of the first wave(s) alone, and wave k of every workgroup lands on SIMD k
    const int rt = (tid + 64 * ((blockIdx.x >> 3) & 3)) & (kCT - 1);

    // ---- 1. Laplacian regions of levels 1..L -> LDS; the loads of a thread are issued in batches before their LDS stores
    if (L >= 1 && !(PF_CF_ABLATE & 1)) {
        // level 1 (three in five of the pixels).  Its INTERIOR -- the (kBW / 2) x (kBH / 2) pixels under the block itself -- lies in the block's own
        // tile: a thread takes four pixels of a row with the widest aligned loads (48 / 24 bytes), exactly one such task per thread.  The RING
        // around it (one pixel where the level's image goes on: up to 2 (66 + 16) pixels, possibly another tile's or a halo strip's) goes pixel by
        // pixel.  (Round 6, first form: the whole region pixel by pixel, five fetches a thread -- 225 vector instructions per wave more.)
        static_assert((kBW / 2 / 4) * (kBH / 2) == kCT, "one interior task per thread");
        const int lap1 = (int)lay.lap_off[1];
        const int iy0 = Y0 >> 1, ix0 = X0 >> 1;                                 // the interior's first row / column in the level-1 image
        Row4<F32> in4;
        {
            const int r = rt / (kBW / 8), cq = rt - r * (kBW / 8);
            in4.load((const PF_GLOBAL T*)(self + lap1) + (((ly0 >> 1) + r) * (kElePixels >> 1) + (lx0 >> 1) + 4 * cq) * 3);
        }
        const int ta = iy0 - r1.y0, tb = r1.y0 + r1.h - (iy0 + kBH / 2), la = ix0 - r1.x0, lb = r1.x0 + r1.w - (ix0 + kBW / 2);      // ring: rows above / below, columns left / right (0 or 1 each)
        const int n_top = ta * r1.w, n_bot = tb * r1.w, n_left = la * (kBH / 2), n_ring = n_top + n_bot + n_left + lb * (kBH / 2);
        WT v[3]; int ring_idx = -1;
        v[0] = v[1] = v[2] = (WT)0;
        if (rt < n_ring) {
            int ry, rx;
            if (rt < n_top) { ry = 0; rx = rt; }
            else if (rt < n_top + n_bot) { ry = r1.h - 1; rx = rt - n_top; }
            else if (rt < n_top + n_bot + n_left) { ry = ta + (rt - n_top - n_bot); rx = 0; }
            else { ry = ta + (rt - n_top - n_bot - n_left); rx = r1.w - 1; }
            ring_idx = ry * r1.w + rx;
            if constexpr (MOSAIC) fetch_mosaic<F32>(table, wx, 1, lap1, r1.y0 + ry, r1.x0 + rx, v);
            else fetch_blend<F32>(sh.job, lay.nlev, 1, lap1, r1.y0 + ry, r1.x0 + rx, v);
        }
        const int n1 = r1.h * r1.w;
        // levels 2..L: the flat list dealt over the threads
        constexpr int kIts2 = (region_px_total(2) + kCT - 1) / kCT;           // 3
        WT u[kIts2][3];
#pragma unroll
        for (int it = 0; it < kIts2; it++) {
            const int idx = n1 + rt + it * kCT;
            u[it][0] = u[it][1] = u[it][2] = (WT)0;
            if (idx < total) {
                int lv = 2;
#pragma unroll
                for (int i = 3; i < kMaxLevels; i++) lv += (i <= L && idx >= sh.reg[i].poff) ? 1 : 0;
                const RegL r = sh.reg[lv];
                const int loc = idx - r.poff, ry = div_small(loc, r.rcp_w), rx = loc - ry * r.w;
                if constexpr (MOSAIC) fetch_mosaic<F32>(table, wx, lv, r.lap_off, r.y0 + ry, r.x0 + rx, u[it]);
                else fetch_blend<F32>(sh.job, lay.nlev, lv, r.lap_off, r.y0 + ry, r.x0 + rx, u[it]);
            }
        }
        {
            const int r = rt / (kBW / 8), cq = rt - r * (kBW / 8);
            WT* d = lds + ((ta + r) * r1.w + la + 4 * cq) * 3;
#pragma unroll
            for (int e = 0; e < 12; e++) d[e] = in4.in(e);
        }
        if (ring_idx >= 0) { lds[ring_idx * 3] = v[0]; lds[ring_idx * 3 + 1] = v[1]; lds[ring_idx * 3 + 2] = v[2]; }
#pragma unroll
        for (int it = 0; it < kIts2; it++) {
            const int idx = n1 + rt + it * kCT;
            if (idx < total) { lds[idx * 3] = u[it][0]; lds[idx * 3 + 1] = u[it][1]; lds[idx * 3 + 2] = u[it][2]; }
        }
    }
    __syncthreads();

    // ---- 2. restore levels L-1 .. 1 in place: pyr[i-1] = pyrUp(pyr[i]) + pyr[i-1], one thread per 2 x 2 destination quad
    // (quads aligned to even coordinates; a quad on the rim of the region has pixels outside it, which are not stored)
    for (int i = (PF_CF_ABLATE & 2) ? 1 : L; i >= 2; i--) {
        const Reg rs = level_region(i, Y0, X0, rows0, cols0), rd = level_region(i - 1, Y0, X0, rows0, cols0);
        const WT* src = lds + rs.poff * 3 - (rs.y0 * rs.w + rs.x0) * 3;          // [(y * w + x) * 3] = source pixel (y, x)
        WT* dst = lds + rd.poff * 3 - (rd.y0 * rd.w + rd.x0) * 3;
        const int qy0 = rd.y0 >> 1, qx0 = rd.x0 >> 1, qw = ((rd.x0 + rd.w - 1) >> 1) - qx0 + 1, nq = (((rd.y0 + rd.h - 1) >> 1) - qy0 + 1) * qw;
        const float rcp_qw = 1.f / (float)qw;
        const int ylo = rs.y0, yhi = rs.y0 + rs.h - 1, xlo = rs.x0, xhi = rs.x0 + rs.w - 1;
        for (int qi = rt; qi < nq; qi += kCT) {
            const int qy = div_small(qi, rcp_qw), qx = qi - qy * qw;
            const int sy = qy0 + qy, sx = qx0 + qx;
            // source rows sy - 1, sy, sy + 1 with pyrUp's row rule (-1 -> 1, rows -> rows - 1), then clamped into the region: a row the region
            // lacks is only ever asked for by a destination pixel outside the destination region
            int r0 = sy - 1; if (r0 < 0) r0 = rs.rows > 1 ? 1 : 0;
            int r2 = sy + 1; if (r2 > rs.rows - 1) r2 = rs.rows - 1;
            r0 = r0 < ylo ? ylo : (r0 > yhi ? yhi : r0); r2 = r2 > yhi ? yhi : r2;
            const int r1y = sy > yhi ? yhi : sy;
            int ca = sx - 1; ca = ca < xlo ? xlo : ca;
            int cc = sx + 1; cc = cc > xhi ? xhi : cc;
            const int cb = sx > xhi ? xhi : sx;
            const int rowo[3] = { r0 * rs.w * 3, r1y * rs.w * 3, r2 * rs.w * 3 };
            WT E[3][3], O[3][3];
            if (sx > 0 && sx < rs.cols - 1) {
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const WT a = src[rowo[rr] + ca * 3 + k], b = src[rowo[rr] + cb * 3 + k], c = src[rowo[rr] + cc * 3 + k];
                        E[rr][k] = a + b * 6 + c; O[rr][k] = b + c;
                    }
            } else {
                const bool single = rs.cols == 1, left = sx == 0;
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const WT a = src[rowo[rr] + ca * 3 + k], b = src[rowo[rr] + cb * 3 + k], c = src[rowo[rr] + cc * 3 + k];
                        if (single)    { E[rr][k] = b * 8; O[rr][k] = b * 2; }
                        else if (left) { E[rr][k] = b * 6 + c * 2; O[rr][k] = b + c; }
                        else           { E[rr][k] = a + b * 7; O[rr][k] = b * 2; }      // right edge
                    }
            }
            const int y = 2 * sy, x = 2 * sx;
            const bool vy0 = y >= rd.y0, vy1 = y + 1 < rd.y0 + rd.h, vx0 = x >= rd.x0, vx1 = x + 1 < rd.x0 + rd.w;
            WT* d0 = dst + (y * rd.w + x) * 3;
            WT* d1 = d0 + rd.w * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (vy0 && vx0) d0[k] = add_sat(up_ee(E[0][k], E[1][k], E[2][k]), d0[k]);
                if (vy0 && vx1) d0[3 + k] = add_sat(up_eo(O[0][k], O[1][k], O[2][k]), d0[3 + k]);
                if (vy1 && vx0) d1[k] = add_sat(up_oe(E[1][k], E[2][k]), d1[k]);
                if (vy1 && vx1) d1[3 + k] = add_sat(up_oo(O[1][k], O[2][k]), d1[3 + k]);
            }
        }
        __syncthreads();
    }

    // ---- 3. level 0: 2 rows x 4 columns per thread
    const PF_GLOBAL T* lap0 = (const PF_GLOBAL T*)(self + lay.lap_off[0]);
    const PF_GLOBAL float* w0 = (const PF_GLOBAL float*)(self + lay.w_off[0]);
    const WT* base1 = lds + r1.poff * 3 - (r1.y0 * r1.w + r1.x0) * 3;
#pragma unroll 1
    for (int pass = 0; pass < (kBH / 2) * (kBW / 4) / kCT; pass++) {
        const int t = tid + pass * kCT, q = t % (kBW / 4), rp = t / (kBW / 4);
        const int ty = ly0 + 2 * rp, tx = lx0 + 4 * q;                     // inside the tile
        // the weights first: their eight "== 0" bits are taken while the Laplacian loads are still in flight
        Row4<F32> px[2];
        unsigned zmask;
        {
            const f4 wa = *(const PF_GLOBAL f4*)(w0 + ty * kElePixels + tx), wb = *(const PF_GLOBAL f4*)(w0 + (ty + 1) * kElePixels + tx);
            px[0].load(lap0 + (ty * kElePixels + tx) * 3);
            px[1].load(lap0 + ((ty + 1) * kElePixels + tx) * 3);
            zmask = (wa.x == 0.f ? 1u : 0u) | (wa.y == 0.f ? 2u : 0u) | (wa.z == 0.f ? 4u : 0u) | (wa.w == 0.f ? 8u : 0u) |
                    (wb.x == 0.f ? 16u : 0u) | (wb.y == 0.f ? 32u : 0u) | (wb.z == 0.f ? 64u : 0u) | (wb.w == 0.f ? 128u : 0u);
        }
        if (L >= 1 && !(PF_CF_ABLATE & 4)) {
            const int y = Y0 + 2 * rp, x = X0 + 4 * q;                     // in the padded level-0 image; both even, x % 4 == 0
            const int sy = y >> 1, sx = x >> 1;
            int syp = sy - 1; if (syp < 0) syp = r1.rows > 1 ? 1 : 0;
            int syn = sy + 1; if (syn >= r1.rows) syn = r1.rows - 1;
            const bool left = sx == 0, right = sx + 1 == r1.cols - 1;
            const int ca = left ? sx : sx - 1, cd = right ? sx + 1 : sx + 2;   // clamped column indices (values unused at the edges)
            const int rowo[3] = { syp * r1.w * 3, sy * r1.w * 3, syn * r1.w * 3 };
            // per component: the three source rows' horizontal sums of the two quads, then the eight results
#pragma unroll
            for (int k = 0; k < 3; k++) {
                WT E0[3], O0[3], E1[3], O1[3];
#pragma unroll
                for (int rr = 0; rr < 3; rr++) {
                    const WT* row = base1 + rowo[rr] + k;
                    const WT a = row[ca * 3], b = row[sx * 3], c = row[(sx + 1) * 3], d = row[cd * 3];
                    if (!(left || right)) { E0[rr] = a + b * 6 + c; O0[rr] = b + c; E1[rr] = b + c * 6 + d; O1[rr] = c + d; }
                    else {
                        E0[rr] = left ? b * 6 + c * 2 : a + b * 6 + c; O0[rr] = b + c;
                        E1[rr] = right ? b + c * 7 : b + c * 6 + d;    O1[rr] = right ? c * 2 : c + d;
                    }
                }
                px[0].v[k]     = add_sat(up_ee(E0[0], E0[1], E0[2]), px[0].in(k));
                px[0].v[3 + k] = add_sat(up_eo(O0[0], O0[1], O0[2]), px[0].in(3 + k));
                px[0].v[6 + k] = add_sat(up_ee(E1[0], E1[1], E1[2]), px[0].in(6 + k));
                px[0].v[9 + k] = add_sat(up_eo(O1[0], O1[1], O1[2]), px[0].in(9 + k));
                px[1].v[k]     = add_sat(up_oe(E0[1], E0[2]), px[1].in(k));
                px[1].v[3 + k] = add_sat(up_oo(O0[1], O0[2]), px[1].in(3 + k));
                px[1].v[6 + k] = add_sat(up_oe(E1[1], E1[2]), px[1].in(6 + k));
                px[1].v[9 + k] = add_sat(up_oo(O1[1], O1[2]), px[1].in(9 + k));
            }
        }
        else {
#pragma unroll
            for (int e = 0; e < 12; e++) { px[0].v[e] = px[0].in(e); px[1].v[e] = px[1].in(e); }
        }
        // mask, 8U view, stores
#pragma unroll
        for (int r = 0; r < 2; r++) {
            uint32_t b8[12];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const bool zero = (zmask >> (4 * r + p)) & 1u;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int e = p * 3 + k;
                    if constexpr (MOSAIC) {
                        if constexpr (F32) b8[e] = zero ? sat_u8(bg) : sat_u8(__float2int_rn(px[r].v[e] * 255.f));
                        else b8[e] = zero ? sat_u8(bg) : sat_u8(px[r].v[e]);
                    } else {
                        if (zero) px[r].v[e] = (WT)0;
                        if constexpr (F32) b8[e] = sat_u8(__float2int_rn(px[r].v[e] * 255.f));
                        else b8[e] = sat_u8(px[r].v[e]);
                    }
                }
            }
            size_t o;                                                     // pixel index of the row's first output pixel
            if constexpr (MOSAIC) o = (size_t)(Y0 + 2 * rp + r) * cols0 + X0 + 4 * q;
            else o = (size_t)out_tile * (kElePixels * kElePixels) + (ty + r) * kElePixels + tx;
            if (bgr)
                *(u3*)(bgr + o * 3) = u3{ b8[0] | (b8[1] << 8) | (b8[2] << 16) | (b8[3] << 24), b8[4] | (b8[5] << 8) | (b8[6] << 16) | (b8[7] << 24),
                                          b8[8] | (b8[9] << 8) | (b8[10] << 16) | (b8[11] << 24) };
            if constexpr (!MOSAIC) { if (raw) px[r].store((T*)raw + o * 3); }
        }
    }
}

}  // namespace

// Ele::blend + updateTexture's 8U view for `n` tiles in one launch; jobs in device memory
void launch_blend_fused(hipStream_t s, const TileLayout& lay, const BlendJob* jobs_dev, int n, void* raw_out, uint8_t* bgr_out)
{
    if (n <= 0) return;
    const int nblocks = n * (kElePixels / kBW) * (kElePixels / kBH);
    if (lay.f32) hipLaunchKernelGGL((k_collapse_fused<true, false>), dim3(nblocks), dim3(kCT), 0, s, lay, jobs_dev, (const uint64_t*)nullptr, 0, 0, 0, (char*)raw_out, bgr_out, nblocks);
    else         hipLaunchKernelGGL((k_collapse_fused<false, false>), dim3(nblocks), dim3(kCT), 0, s, lay, jobs_dev, (const uint64_t*)nullptr, 0, 0, 0, (char*)raw_out, bgr_out, nblocks);
}

// save(): the pasted wx x wy mosaic collapsed, 8U, background where no weight (table in device memory, 0 = no tile)
void launch_save_fused(hipStream_t s, const TileLayout& lay, const uint64_t* table_dev, int wx, int wy, int bg, uint8_t* bgr_out)
{
    const int nblocks = wx * (kElePixels / kBW) * wy * (kElePixels / kBH);
    if (nblocks <= 0) return;
    if (lay.f32) hipLaunchKernelGGL((k_collapse_fused<true, true>), dim3(nblocks), dim3(kCT), 0, s, lay, (const BlendJob*)nullptr, table_dev, wx, wy, bg, (char*)nullptr, bgr_out, nblocks);
    else         hipLaunchKernelGGL((k_collapse_fused<false, true>), dim3(nblocks), dim3(kCT), 0, s, lay, (const BlendJob*)nullptr, table_dev, wx, wy, bg, (char*)nullptr, bgr_out, nblocks);
}

}  // namespace pf
