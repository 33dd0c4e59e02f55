// collapse_fused.hip -- the output side of the path as ONE launch per request (gfx950, wave64):
//   Ele::blend        MultiBandMap2DCPU.cpp:77-146   3x3 assembly (:93-117), restoreImageFromLaplacePyr (:119, :142),
//                                                     crop + weights[0]==0 mask (:121-126, :144)
//   updateTexture     :149-188                        the 8U view (convertTo CV_8UC3, :156)
//   save              :806-840                        paste per level, one whole-mosaic collapse, 8U, background
//
// Rounds 1-5 ran this as the reference writes it -- a padded square per level in HBM (k_blend_gather), one k_collapse launch
// per level, a finish launch: 12 launches per chunk and every level through HBM twice.  Here a workgroup owns a 64 x 64
// block of the level-0 result and collapses the part of the pyramid that block depends on inside LDS:
//
//   pyrUp is a 3-tap filter, so level i-1 rows [lo, hi] need level i rows [(lo-1)>>1, (hi>>1)+1]: one pixel of halo per
//   level (34^2, 19^2, 12^2, 8^2, 6^2, 5^2 ... pixels for levels 1, 2, ...: 22 KB of LDS for any band count).
//   1. the Laplacian regions of levels 1..L go from the tile slots / packed halo strips / mosaic tiles straight into LDS;
//   2. level L-1 ... 1 are restored in place in LDS (pyrUp + add, the reference's operation order, C-cast and saturation,
//      and the borders of the padded square / the mosaic -- the edge forms of pyrUp are not index reflections in fp32);
//   3. level 0: a thread takes 2 rows x 4 columns, reads its 3 x 4 neighbourhood of level 1 from LDS once, adds the tile's
//      own Laplacian (16-byte loads), applies the weight mask and the 8U view and writes 12 bytes per row.
// Nothing of a level >= 1 ever returns to HBM, level 0 is read once and the result written once.
//
// Bit-exactness: compiled with -ffp-contract=off; every sum below is written in the association order of
// OpenCV 2.4.9's pyrUp_ and checked against the oracle by the blend / save / dist tests.
#include "kernels.hpp"
#include "warp_index.hpp"

namespace pf {
namespace {

#define PF_GLOBAL __attribute__((address_space(1)))

constexpr int kCB = 64;                               // level-0 block edge of a workgroup
constexpr int kCT = 256;                              // threads
constexpr int region_edge(int level) { int s = kCB; for (int i = 0; i < level; i++) s = ((s + 1) >> 1) + 2; return s; }
constexpr int region_px_total() { int n = 0; for (int i = 1; i < kMaxLevels; i++) n += region_edge(i) * region_edge(i); return n; }
constexpr int kLdsPx = region_px_total();             // 1836 pixels = 22 032 B of 3 x 4-byte components
static_assert(region_edge(1) == 34 && region_edge(2) == 19 && region_edge(3) == 12, "pyrUp dependence regions");

typedef float    f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <bool F32> struct Px;
template <> struct Px<false> { using T = short; using WT = int; };
template <> struct Px<true>  { using T = float; using WT = float; };

__device__ __forceinline__ int   cast_up(int v)   { return (int)(short)((v + 32) >> 6); }      // (short) C cast: wraps
__device__ __forceinline__ float cast_up(float v) { return v * (1.f / 64); }
__device__ __forceinline__ int   add_sat(int up, int lap)     { return sat_short(up + lap); }  // cv::add on 16S saturates
__device__ __forceinline__ float add_sat(float up, float lap) { return up + lap; }
__device__ __forceinline__ uint32_t sat_u8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// region of a level held in LDS: rows [y0, y0 + h) x cols [x0, x0 + w) of the level's image (rows x cols), at lds[off ...]
struct Reg { int y0, x0, h, w, off, rows, cols, lap_off; };

struct Shared {
    Reg reg[kMaxLevels];
    BlendJob job;
};

// horizontal pass of pyrUp_ for destination column x of a source row `row` (indexed by absolute source column, 3 components apart)
template <typename WT>
__device__ __forceinline__ WT up_h(const WT* row, int x, int scols)
{
    const int sx = x >> 1;
    if (scols == 1) return row[0] * 8;
    if (x & 1) {
        if (sx == scols - 1) return row[sx * 3] * 8;
        return (row[sx * 3] + row[(sx + 1) * 3]) * 4;
    }
    if (sx == 0) return row[0] * 6 + row[3] * 2;
    if (sx == scols - 1) return row[(sx - 1) * 3] + row[sx * 3] * 7;
    return row[(sx - 1) * 3] + row[sx * 3] * 6 + row[(sx + 1) * 3];
}

// pyrUp(level i)[y][x][k] from the level's LDS region (reg = level i's)
template <typename WT>
__device__ __forceinline__ WT up_at(const WT* lds, const Reg& r, int y, int x, int k)
{
    const int sy = y >> 1;
    const WT* base = lds + r.off + k - (r.y0 * r.w + r.x0) * 3;      // [sy * w * 3 + sx * 3] is the absolute pixel
    int syn = sy + 1; if (syn >= r.rows) syn = r.rows - 1;
    const WT* r1 = base + sy * r.w * 3;
    const WT* r2 = base + syn * r.w * 3;
    if (y & 1) return cast_up((up_h<WT>(r1, x, r.cols) + up_h<WT>(r2, x, r.cols)) * 4);
    int syp = sy - 1; if (syp < 0) syp = r.rows > 1 ? 1 : 0;
    const WT* r0 = base + syp * r.w * 3;
    return cast_up(up_h<WT>(r0, x, r.cols) + up_h<WT>(r1, x, r.cols) * 6 + up_h<WT>(r2, x, r.cols));
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
    using T = typename Px<F32>::T; using WT = typename Px<F32>::WT;
    const int ts = kElePixels >> level, b = job.border ? 1 << (nlev - 1 - level) : 0;
    int rx = 1, sx = px - b, ry = 1, sy = py - b;
    if (sx < 0) { rx = 0; sx += ts; } else if (sx >= ts) { rx = 2; sx -= ts; }
    if (sy < 0) { ry = 0; sy += ts; } else if (sy >= ts) { ry = 2; sy -= ts; }
    const int j = ry * 3 + rx;
    const PF_GLOBAL T* s;
    if (!((job.strip_mask >> j) & 1)) {
        s = (const PF_GLOBAL T*)((const PF_GLOBAL char*)job.src[j] + lap_off) + (sy * ts + sx) * 3;
    } else {                                              // packed strips: levels concatenated, each h x w row-major from the strip's corner
        const int dx = rx - 1, dy = ry - 1;
        int off = 0;
        for (int i = 0; i < level; i++) { int w, h; strip_dims_d(nlev, i, dx, dy, w, h); off += w * h; }
        int w, h; strip_dims_d(nlev, level, dx, dy, w, h);
        const int lx = rx == 0 ? sx - (ts - b) : sx, ly = ry == 0 ? sy - (ts - b) : sy;
        s = (const PF_GLOBAL T*)job.src[j] + (off + ly * w + lx) * 3;
    }
    out[0] = (WT)s[0]; out[1] = (WT)s[1]; out[2] = (WT)s[2];
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
    out[0] = (WT)s[0]; out[1] = (WT)s[1]; out[2] = (WT)s[2];
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
    __device__ __forceinline__ void store(float* p) const {
        f4* q = (f4*)p;
        q[0] = f4{ v[0], v[1], v[2], v[3] }; q[1] = f4{ v[4], v[5], v[6], v[7] }; q[2] = f4{ v[8], v[9], v[10], v[11] };
    }
};
template <> struct Row4<false> {
    int v[12];
    __device__ __forceinline__ void load(const PF_GLOBAL short* p) {
        const PF_GLOBAL u2* q = (const PF_GLOBAL u2*)p;                // 24 bytes, 8-byte aligned
        const u2 a = q[0], b = q[1], c = q[2];
        const uint32_t u[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
#pragma unroll
        for (int i = 0; i < 6; i++) { v[2 * i] = (int)(short)(u[i] & 0xffffu); v[2 * i + 1] = (int)(short)(u[i] >> 16); }
    }
    __device__ __forceinline__ void store(short* p) const {
        uint32_t u[6];
#pragma unroll
        for (int i = 0; i < 6; i++) u[i] = ((uint32_t)v[2 * i] & 0xffffu) | ((uint32_t)v[2 * i + 1] << 16);
        u2* q = (u2*)p;
        q[0] = u2{ u[0], u[1] }; q[1] = u2{ u[2], u[3] }; q[2] = u2{ u[4], u[5] };
    }
};

// One workgroup = one 64 x 64 block of a level-0 result.
//   MOSAIC = false: block (blockIdx % 16) of tile job[blockIdx / 16] (Ele::blend); results to raw / bgr at tile index job.out
//   MOSAIC = true : block of the wx*4 x wy*4 block grid of the pasted mosaic (save); result to bgr (rows x cols x 3)
// Workgroup ids are dealt so that the blocks of one tile / of neighbouring tiles run on one XCD (they share the upper levels in its L2).
template <bool F32, bool MOSAIC>
__global__ __launch_bounds__(kCT) void k_collapse_fused(TileLayout lay, const BlendJob* __restrict__ jobs, const uint64_t* __restrict__ table,
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
        const int bx = wg % (wx * 4), by = wg / (wx * 4);
        Y0 = by * kCB; X0 = bx * kCB; rows0 = wy * kElePixels; cols0 = wx * kElePixels;
        self = (const PF_GLOBAL char*)table[(Y0 >> 8) * wx + (X0 >> 8)];
        ly0 = Y0 & 255; lx0 = X0 & 255;
        if (!self) {                                   // no tile here: background (.cpp:840, weights never pasted stay 0)
            const uint32_t b8 = sat_u8(bg), word = b8 * 0x01010101u;
            for (int t = tid; t < kCB * (kCB * 3 / 4); t += kCT) {
                const int r = t / (kCB * 3 / 4), c = t - r * (kCB * 3 / 4);
                ((uint32_t*)(bgr + ((size_t)(Y0 + r) * cols0 + X0) * 3))[c] = word;
            }
            return;
        }
    } else {
        const int z = wg >> 4, blk = wg & 15;
        if (tid < (int)(sizeof(BlendJob) / 4)) ((uint32_t*)&sh.job)[tid] = ((const uint32_t*)(jobs + z))[tid];
        const int b0 = jobs[z].border ? 1 << L : 0;
        ly0 = (blk >> 2) * kCB; lx0 = (blk & 3) * kCB;
        Y0 = b0 + ly0; X0 = b0 + lx0; rows0 = cols0 = kElePixels + 2 * b0;
        self = (const PF_GLOBAL char*)jobs[z].src[4];
        out_tile = jobs[z].out;
    }

    // ---- regions of levels 1..L this block depends on
    if (tid == 0) {
        int ylo = Y0, yhi = Y0 + kCB - 1, xlo = X0, xhi = X0 + kCB - 1, off = 0;
        for (int i = 1; i <= L; i++) {
            const int rows = rows0 >> i, cols = cols0 >> i;
            ylo = (ylo - 1) >> 1; if (ylo < 0) ylo = 0;
            xlo = (xlo - 1) >> 1; if (xlo < 0) xlo = 0;
            yhi = (yhi >> 1) + 1; if (yhi > rows - 1) yhi = rows - 1;
            xhi = (xhi >> 1) + 1; if (xhi > cols - 1) xhi = cols - 1;
            Reg r; r.y0 = ylo; r.x0 = xlo; r.h = yhi - ylo + 1; r.w = xhi - xlo + 1; r.off = off; r.rows = rows; r.cols = cols; r.lap_off = (int)lay.lap_off[i];
            sh.reg[i] = r;
            off += r.h * r.w * 3;
        }
    }
    __syncthreads();

    // ---- 1. Laplacian regions of levels 1..L -> LDS
    for (int i = 1; i <= L; i++) {
        const Reg r = sh.reg[i];
        const int n = r.h * r.w;
        for (int idx = tid; idx < n; idx += kCT) {
            const int ry = idx / r.w, rx = idx - ry * r.w;
            WT v[3];
            if constexpr (MOSAIC) fetch_mosaic<F32>(table, wx, i, r.lap_off, r.y0 + ry, r.x0 + rx, v);
            else fetch_blend<F32>(sh.job, lay.nlev, i, r.lap_off, r.y0 + ry, r.x0 + rx, v);
            WT* d = lds + r.off + idx * 3;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2];
        }
    }
    __syncthreads();

    // ---- 2. restore levels L-1 .. 1 in place: pyr[i-1] = pyrUp(pyr[i]) + pyr[i-1]
    for (int i = L; i >= 2; i--) {
        const Reg rs = sh.reg[i], rd = sh.reg[i - 1];
        const int n = rd.h * rd.w;
        for (int idx = tid; idx < n; idx += kCT) {
            const int ry = idx / rd.w, rx = idx - ry * rd.w;
            WT* d = lds + rd.off + idx * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) d[k] = add_sat(up_at<WT>(lds, rs, rd.y0 + ry, rd.x0 + rx, k), d[k]);
        }
        __syncthreads();
    }

    // ---- 3. level 0: 2 rows x 4 columns per thread, two passes over the block
    const PF_GLOBAL T* lap0 = (const PF_GLOBAL T*)(self + lay.lap_off[0]);
    const PF_GLOBAL float* w0 = (const PF_GLOBAL float*)(self + lay.w_off[0]);
    Reg r1{};
    if (L >= 1) r1 = sh.reg[1];
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const int t = tid + pass * kCT, q = t & 15, rp = t >> 4;
        const int ty = ly0 + 2 * rp, tx = lx0 + 4 * q;                     // inside the tile
        Row4<F32> px[2];
        f4 wv[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            px[r].load(lap0 + ((ty + r) * kElePixels + tx) * 3);
            wv[r] = *(const PF_GLOBAL f4*)(w0 + (ty + r) * kElePixels + tx);
        }
        if (L >= 1) {
            const int y = Y0 + 2 * rp, x = X0 + 4 * q;                     // in the padded level-0 image; both even, x % 4 == 0
            const int sy = y >> 1, sx = x >> 1;
            int syp = sy - 1; if (syp < 0) syp = r1.rows > 1 ? 1 : 0;
            int syn = sy + 1; if (syn >= r1.rows) syn = r1.rows - 1;
            const bool left = sx == 0, right = sx + 1 == r1.cols - 1;
            const int ca = left ? sx : sx - 1, cd = right ? sx + 1 : sx + 2;   // clamped column indices (values unused at the edges)
            const WT* base = lds + r1.off - (r1.y0 * r1.w + r1.x0) * 3;
            const int rowi[3] = { syp, sy, syn };
            WT h[3][12];                                                   // horizontal sums of the three source rows, 4 columns x 3 components
#pragma unroll
            for (int rr = 0; rr < 3; rr++) {
                const WT* row = base + rowi[rr] * r1.w * 3;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const WT a = row[ca * 3 + k], b = row[sx * 3 + k], c = row[(sx + 1) * 3 + k], d = row[cd * 3 + k];
                    h[rr][k]     = left ? b * 6 + c * 2 : a + b * 6 + c;
                    h[rr][3 + k] = (b + c) * 4;
                    h[rr][6 + k] = right ? b + c * 7 : b + c * 6 + d;
                    h[rr][9 + k] = right ? c * 8 : (c + d) * 4;
                }
            }
#pragma unroll
            for (int e = 0; e < 12; e++) {
                const WT up0 = cast_up(h[0][e] + h[1][e] * 6 + h[2][e]);
                const WT up1 = cast_up((h[1][e] + h[2][e]) * 4);
                px[0].v[e] = add_sat(up0, px[0].v[e]);
                px[1].v[e] = add_sat(up1, px[1].v[e]);
            }
        }
        // mask, 8U view, stores
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float wr[4] = { wv[r].x, wv[r].y, wv[r].z, wv[r].w };
            uint32_t b8[12];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const bool zero = wr[p] == 0.f;
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
            if (bgr) {
                uint32_t* d = (uint32_t*)(bgr + o * 3);
                d[0] = b8[0] | (b8[1] << 8) | (b8[2] << 16) | (b8[3] << 24);
                d[1] = b8[4] | (b8[5] << 8) | (b8[6] << 16) | (b8[7] << 24);
                d[2] = b8[8] | (b8[9] << 8) | (b8[10] << 16) | (b8[11] << 24);
            }
            if constexpr (!MOSAIC) { if (raw) px[r].store((T*)raw + o * 3); }
        }
    }
}

}  // namespace

// Ele::blend + updateTexture's 8U view for `n` tiles in one launch; jobs in device memory
void launch_blend_fused(hipStream_t s, const TileLayout& lay, const BlendJob* jobs_dev, int n, void* raw_out, uint8_t* bgr_out)
{
    if (n <= 0) return;
    const int nblocks = n * 16;
    if (lay.f32) hipLaunchKernelGGL((k_collapse_fused<true, false>), dim3(nblocks), dim3(kCT), 0, s, lay, jobs_dev, (const uint64_t*)nullptr, 0, 0, 0, (char*)raw_out, bgr_out, nblocks);
    else         hipLaunchKernelGGL((k_collapse_fused<false, false>), dim3(nblocks), dim3(kCT), 0, s, lay, jobs_dev, (const uint64_t*)nullptr, 0, 0, 0, (char*)raw_out, bgr_out, nblocks);
}

// save(): the pasted wx x wy mosaic collapsed, 8U, background where no weight (table in device memory, 0 = no tile)
void launch_save_fused(hipStream_t s, const TileLayout& lay, const uint64_t* table_dev, int wx, int wy, int bg, uint8_t* bgr_out)
{
    const int nblocks = wx * 4 * wy * 4;
    if (nblocks <= 0) return;
    if (lay.f32) hipLaunchKernelGGL((k_collapse_fused<true, true>), dim3(nblocks), dim3(kCT), 0, s, lay, (const BlendJob*)nullptr, table_dev, wx, wy, bg, (char*)nullptr, bgr_out, nblocks);
    else         hipLaunchKernelGGL((k_collapse_fused<false, true>), dim3(nblocks), dim3(kCT), 0, s, lay, (const BlendJob*)nullptr, table_dev, wx, wy, bg, (char*)nullptr, bgr_out, nblocks);
}

}  // namespace pf
