// jpeg_device.hip -- cv::imread's JPEG leg (backup/map2dfusion.cpp:129-132) on the GPU, byte-equal to jpeg_decode.cpp and therefore to
// libjpeg-turbo (tests/test_gpu_jpeg.py); integer arithmetic throughout.
//   Huffman pass   for the streams cameras write (sequential, one scan, with or without restart intervals) the host parses the headers and
//                  strips the byte stuffing and the RSTn markers; the scan's bytes cross PCIe and are decoded one thread per 512-bit subsequence in rounds until a round changes
//                  nothing (jpeg_huff_par.hpp; k_huff_round, k_scan_*, k_huff_write).  Any other stream, and any stream whose write pass
//                  does not end exactly on the frame's last block, is entropy-decoded on the host (jpeg_decode.cpp) and its coefficients
//                  uploaded (2 B per sample: the size of the frame they become).
//   back end       dequantise + ISLOW IDCT per 8x8 block (jidctint.c) into component planes (k_jpeg_idct), then fancy upsampling
//                  (jdsample.c) and YCbCr -> BGR (jdcolor.c) per pixel (k_jpeg_colour8 / k_jpeg_colour), writing the BGR8 keyframe where
//                  the level kernel reads it.  HBM-bound byte work: 2 B of coefficients in and 1 B out per sample; 1.5 B of planes in and
//                  3 B out per pixel (4:2:0) -- 54 MB + 54 MB for a 4000 x 3000 frame.
#include "jpeg_device.hpp"
#include "jpeg_huff_par.hpp"
#include "env.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace pf {

struct JpegDevComp {
    int bw, bh, w, ht, he, ve, stride;         // blocks held, real samples, expansion to full resolution, plane row pitch (bw * 8)
    unsigned coef_off, plane_off;              // int16 / byte offsets of the component
    int pad_[3];                               // q on a 16-byte boundary: a thread loads a row of it at once
    uint16_t q[64];
};
struct JpegDevFrame {
    int rows, cols, ncomp, ycc;
    unsigned block_first[4];                   // first block of each component in the launch's numbering; [ncomp] = all blocks
    JpegDevComp c[3];
};
static_assert(offsetof(JpegDevFrame, c[0].q) % 16 == 0 && sizeof(JpegDevComp) % 16 == 0, "quantiser rows are loaded 16 bytes at a time");
static_assert(sizeof(JpegDevFrame) <= 1024, "the header travels in the first KiB of the coefficient buffer");
constexpr size_t kHeaderBytes = 1024;
constexpr size_t kPlanBytes = (sizeof(HuffParPlan) + 255) & ~(size_t)255;          // the parallel Huffman pass's plan, then the scan's bytes

namespace {

typedef uint32_t u32;                          // sums that wrap, shifted as signed: jpeg_decode.cpp's idct8, statement for statement
__device__ inline int dsc(u32 x, int n) { return (int)(x + (1u << (n - 1))) >> n; }
__device__ inline int lim(int x)              // jdmaster.c prepare_range_limit_table, IDCT part, after "& RANGE_MASK"
{
    x &= 1023;
    return x < 128 ? x + 128 : x < 512 ? 255 : x < 896 ? 0 : x - 896;
}

// jidctint.c: one 8-point pass on values in registers; `sh`: the pass's descale
__device__ inline void idct8(const u32 v[8], int out[8], int sh)
{
    constexpr u32 F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                  F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    u32 z2 = v[2], z3 = v[6];
    u32 z1 = (z2 + z3) * F0_541;
    u32 tmp2 = z1 - z3 * F1_847, tmp3 = z1 + z2 * F0_765;
    u32 tmp0 = (v[0] + v[4]) << 13, tmp1 = (v[0] - v[4]) << 13;
    const u32 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; u32 z4 = tmp1 + tmp3;
    const u32 z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
    z1 *= 0u - F0_899; z2 *= 0u - F2_562; z3 *= 0u - F1_961; z4 *= 0u - F0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = dsc(tmp10 + tmp3, sh); out[7] = dsc(tmp10 - tmp3, sh);
    out[1] = dsc(tmp11 + tmp2, sh); out[6] = dsc(tmp11 - tmp2, sh);
    out[2] = dsc(tmp12 + tmp1, sh); out[5] = dsc(tmp12 - tmp1, sh);
    out[3] = dsc(tmp13 + tmp0, sh); out[4] = dsc(tmp13 - tmp0, sh);
}

// 256 threads = 32 blocks of 8 x 8: a thread loads one row of its block (16 B) and its quantisers, the block is transposed through
// LDS (72-word pitch per block: the eight blocks of a wave sit on disjoint banks for the column accesses), a thread runs the column
// pass in place, then the row pass, and stores 8 bytes of the component plane.
constexpr int kPitch = 72;
__global__ __launch_bounds__(256) void k_jpeg_idct(const JpegDevFrame* __restrict__ F, const int16_t* __restrict__ coef, uint8_t* __restrict__ planes)
{
    __shared__ int ws[32 * kPitch];
    const int t = threadIdx.x, lb = t >> 3, j = t & 7;
    const unsigned blk = blockIdx.x * 32u + (unsigned)lb;
    const int nc = F->ncomp;
    const bool live = blk < F->block_first[nc];
    int c = 0;
    if (nc > 1 && blk >= F->block_first[1]) c = 1;
    if (nc > 2 && blk >= F->block_first[2]) c = 2;
    const JpegDevComp& C = F->c[c];
    const unsigned b = blk - F->block_first[c];
    int* w = ws + lb * kPitch;
    if (live) {
        const uint4 raw = *reinterpret_cast<const uint4*>(coef + C.coef_off + (size_t)b * 64 + 8 * j);
        const uint4 qr = *reinterpret_cast<const uint4*>(C.q + 8 * j);
        const unsigned rv[4] = { raw.x, raw.y, raw.z, raw.w }, qv[4] = { qr.x, qr.y, qr.z, qr.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            w[8 * j + 2 * k] = (int)(short)(rv[k] & 0xffff) * (int)(qv[k] & 0xffff);
            w[8 * j + 2 * k + 1] = (int)(short)(rv[k] >> 16) * (int)(qv[k] >> 16);
        }
    }
    __syncthreads();
    u32 v[8]; int o[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = (u32)w[8 * r + j];                // column j
    idct8(v, o, 11);                                                     // CONST_BITS - PASS1_BITS
#pragma unroll
    for (int r = 0; r < 8; r++) w[8 * r + j] = o[r];                     // in place: the thread owns the column
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (u32)w[8 * j + k];                // row j
    idct8(v, o, 18);                                                     // CONST_BITS + PASS1_BITS + 3
    if (live) {
        const unsigned by = b / (unsigned)C.bw, bx = b - by * (unsigned)C.bw;
        uint2 px;
        px.x = (unsigned)lim(o[0]) | ((unsigned)lim(o[1]) << 8) | ((unsigned)lim(o[2]) << 16) | ((unsigned)lim(o[3]) << 24);
        px.y = (unsigned)lim(o[4]) | ((unsigned)lim(o[5]) << 8) | ((unsigned)lim(o[6]) << 16) | ((unsigned)lim(o[7]) << 24);
        *reinterpret_cast<uint2*>(planes + C.plane_off + (size_t)(by * 8 + j) * C.stride + bx * 8) = px;
    }
}

// jdsample.c: the sample of a component at full-resolution position (x, y).  Rows beyond the real ones repeat the last real row
// (jdmainct.c's context rows); columns beyond are never asked for.
__device__ inline int sample_at(const JpegDevComp& C, const uint8_t* __restrict__ planes, int x, int y)
{
    const uint8_t* p = planes + C.plane_off;
    const int he = C.he, ve = C.ve, w = C.w, last = C.ht - 1, st = C.stride;
    auto row = [&](int r) { r = r < 0 ? 0 : r > last ? last : r; return p + (size_t)r * st; };
    if (he == 1 && ve == 1) return row(y)[x];
    if (he == 2 && ve == 1) {
        const uint8_t* in = row(y);
        const int cx = x >> 1;
        if (w <= 2) return in[cx];
        if (x & 1) return cx == w - 1 ? in[cx] : (in[cx] * 3 + in[cx + 1] + 2) >> 2;
        return cx == 0 ? in[0] : (in[cx] * 3 + in[cx - 1] + 1) >> 2;
    }
    if (he == 1 && ve == 2) {
        const int cy = y >> 1;
        return (row(cy)[x] * 3 + row((y & 1) ? cy + 1 : cy - 1)[x] + ((y & 1) ? 2 : 1)) >> 2;
    }
    if (he == 2 && ve == 2 && w > 2) {
        const int cy = y >> 1, cx = x >> 1;
        const uint8_t* in0 = row(cy); const uint8_t* in1 = row((y & 1) ? cy + 1 : cy - 1);
        const int cur = in0[cx] * 3 + in1[cx];
        if (x & 1) return cx == w - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + in0[cx + 1] * 3 + in1[cx + 1] + 7) >> 4;
        return cx == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + in0[cx - 1] * 3 + in1[cx - 1] + 8) >> 4;
    }
    return row(y / ve)[x / he];
}

__device__ inline unsigned clamp8(int v) { return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); }

// four pixels of a row per thread: upsample, jdcolor.c's YCbCr -> RGB (SCALEBITS 16), BGR out (what cv::imread hands over)
__global__ __launch_bounds__(256) void k_jpeg_colour(const JpegDevFrame* __restrict__ F, const uint8_t* __restrict__ planes, uint8_t* __restrict__ bgr)
{
    const int cols = F->cols, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= cols) return;
    const int nc = F->ncomp, ycc = F->ycc;
    unsigned char o[12];
    const int n = cols - x0 < 4 ? cols - x0 : 4;
    for (int i = 0; i < n; i++) {
        const int x = x0 + i;
        const int a = sample_at(F->c[0], planes, x, y);
        int r, g, b;
        if (nc == 1) r = g = b = a;
        else {
            const int u = sample_at(F->c[1], planes, x, y), v = sample_at(F->c[2], planes, x, y);
            if (ycc) {
                const int cb = u - 128, cr = v - 128;
                r = (int)clamp8(a + ((91881 * cr + 32768) >> 16));
                g = (int)clamp8(a + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
                b = (int)clamp8(a + ((116130 * cb + 32768) >> 16));
            } else { r = a; g = u; b = v; }
        }
        o[3 * i] = (unsigned char)b; o[3 * i + 1] = (unsigned char)g; o[3 * i + 2] = (unsigned char)r;
    }
    uint8_t* dst = bgr + ((size_t)y * cols + x0) * 3;
    if (n == 4 && ((cols * 3) & 3) == 0) {
        unsigned* d4 = reinterpret_cast<unsigned*>(dst);
        d4[0] = o[0] | (o[1] << 8) | (o[2] << 16) | ((unsigned)o[3] << 24);
        d4[1] = o[4] | (o[5] << 8) | (o[6] << 16) | ((unsigned)o[7] << 24);
        d4[2] = o[8] | (o[9] << 8) | (o[10] << 16) | ((unsigned)o[11] << 24);
    } else
        for (int i = 0; i < 3 * n; i++) dst[i] = o[i];
}

// Restart intervals reset the DC prediction (jdhuff.c process_restart): a component's DC value is the running sum of its differences since the
// start of its interval -- the global running sum (k_scan_apply<2>) minus the one at the last block before the interval.
__global__ __launch_bounds__(256) void k_dc_restart(const HuffParPlan* __restrict__ P, const int* __restrict__ sums, int dc_stride, int16_t* __restrict__ coef)
{
    const int comp = (int)blockIdx.y, N = P->cblocks[comp], j = (int)(blockIdx.x * 256u + threadIdx.x);
    if (j >= N) return;
    const int group = (int)(P->rst_blocks / (uint32_t)P->bpm) * P->ch[comp] * P->cv[comp];
    const int g0 = j / group * group;
    const int* sc = sums + comp * dc_stride;
    coef[huff_par_comp_block(*P, comp, (uint32_t)j)] = (int16_t)(sc[j] - (g0 ? sc[g0 - 1] : 0));
}

// The shapes cameras write (three components, luma at full resolution, both chroma planes expanded HE x VE in {1x1, 2x1, 2x2} with
// more than two samples a row, cols a multiple of 8): eight pixels of a row per thread -- the luma bytes in one load, the chroma
// samples of the row (and of its neighbour row for 2x2) as one word plus the two samples next to it, 24 bytes out in six words.
// Same arithmetic as sample_at(), case by case.
template <int HE, int VE>
__global__ __launch_bounds__(256) void k_jpeg_colour8(const JpegDevFrame* __restrict__ F, const uint8_t* __restrict__ planes, uint8_t* __restrict__ bgr)
{
    const int cols = F->cols, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (x0 >= cols) return;
    const JpegDevComp& Y = F->c[0];
    const uint2 yy = *reinterpret_cast<const uint2*>(planes + Y.plane_off + (size_t)y * Y.stride + x0);
    int ch[2][8];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const JpegDevComp& C = F->c[1 + k];
        const uint8_t* p = planes + C.plane_off;
        const int last = C.ht - 1, w = C.w;
        if (HE == 1) {
            const uint2 v = *reinterpret_cast<const uint2*>(p + (size_t)y * C.stride + x0);
#pragma unroll
            for (int i = 0; i < 8; i++) ch[k][i] = (int)(((i < 4 ? v.x : v.y) >> (8 * (i & 3))) & 255);
        } else {
            const int cy = VE == 2 ? y >> 1 : y, cx0 = x0 >> 1;
            const int ny = VE == 2 ? ((y & 1) ? (cy + 1 > last ? last : cy + 1) : (cy - 1 < 0 ? 0 : cy - 1)) : cy;
            const uint8_t* in0 = p + (size_t)cy * C.stride; const uint8_t* in1 = p + (size_t)ny * C.stride;
            const int xl = cx0 > 0 ? cx0 - 1 : 0, xr = cx0 + 4 < w ? cx0 + 4 : w - 1;
            const unsigned m0 = *reinterpret_cast<const unsigned*>(in0 + cx0);
            int cs[6];
            if (VE == 2) {
                const unsigned m1 = *reinterpret_cast<const unsigned*>(in1 + cx0);
                cs[0] = in0[xl] * 3 + in1[xl]; cs[5] = in0[xr] * 3 + in1[xr];
#pragma unroll
                for (int i = 0; i < 4; i++) cs[1 + i] = (int)((m0 >> (8 * i)) & 255) * 3 + (int)((m1 >> (8 * i)) & 255);
            } else {
                cs[0] = in0[xl]; cs[5] = in0[xr];
#pragma unroll
                for (int i = 0; i < 4; i++) cs[1 + i] = (int)((m0 >> (8 * i)) & 255);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int cx = cx0 + (i >> 1), c = cs[1 + (i >> 1)];
                if (VE == 2) ch[k][i] = (i & 1) ? (cx == w - 1 ? (c * 4 + 7) >> 4 : (c * 3 + cs[2 + (i >> 1)] + 7) >> 4)
                                               : (cx == 0 ? (c * 4 + 8) >> 4 : (c * 3 + cs[i >> 1] + 8) >> 4);
                else         ch[k][i] = (i & 1) ? (cx == w - 1 ? c : (c * 3 + cs[2 + (i >> 1)] + 2) >> 2)
                                               : (cx == 0 ? c : (c * 3 + cs[i >> 1] + 1) >> 2);
            }
        }
    }
    const int ycc = F->ycc;
    unsigned o[24];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int a = (int)(((i < 4 ? yy.x : yy.y) >> (8 * (i & 3))) & 255), u = ch[0][i], v = ch[1][i];
        if (ycc) {
            const int cb = u - 128, cr = v - 128;
            o[3 * i + 2] = clamp8(a + ((91881 * cr + 32768) >> 16));
            o[3 * i + 1] = clamp8(a + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
            o[3 * i] = clamp8(a + ((116130 * cb + 32768) >> 16));
        } else { o[3 * i + 2] = (unsigned)a; o[3 * i + 1] = (unsigned)u; o[3 * i] = (unsigned)v; }
    }
    unsigned* d4 = reinterpret_cast<unsigned*>(bgr + ((size_t)y * cols + x0) * 3);
#pragma unroll
    for (int i = 0; i < 6; i++) d4[i] = o[4 * i] | (o[4 * i + 1] << 8) | (o[4 * i + 2] << 16) | (o[4 * i + 3] << 24);
}

// ---- the Huffman pass in parallel (jpeg_huff_par.hpp): one thread per subsequence of kSubBits bits
struct HuffParResult { uint32_t g_end, p, ck, bad; };            // bad: a restart segment did not end on its last block

// A workgroup's 256 subsequences are contiguous in the stream: their words (plus what a symbol starting on the window's last bit may
// still read) and the code tables are staged in LDS once -- a symbol costs two or three dependent look-ups, and from L2 those were the whole
// of a round's 72 us (10 us from LDS).
constexpr int kSubWords = kSubBits / 32;
constexpr int kStageWords = 256 * kSubWords + 4;
struct HuffLds { HuffParTable tab[8]; uint32_t words[kStageWords]; };

__device__ inline void huff_stage(const HuffParPlan* __restrict__ P, const uint32_t* __restrict__ words, HuffLds& L, uint32_t first_sub)
{
    const uint32_t used = P->used;
    for (int t = 0; t < 8; t++) {                                 // the tables the scan uses (four for a camera's frame)
        if (!((used >> t) & 1u)) continue;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&P->tab[t]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&L.tab[t]);
        for (unsigned j = threadIdx.x; j < sizeof(HuffParTable) / 4; j += 256) dst[j] = src[j];
    }
    const uint32_t w0 = first_sub * (uint32_t)kSubWords, navail = ((P->nbits >> 3) + 16 + 3) / 4;      // the scan's bytes and the 16 zero bytes behind them
    for (unsigned j = threadIdx.x; j < (unsigned)kStageWords; j += 256) { const uint32_t w = w0 + j; L.words[j] = w < navail ? words[w] : 0u; }
    __syncthreads();
}

// Launch 0: every subsequence from its own first bit as if a block began there, then sweeps from the end state its predecessor reached in the
// sweep before; launch r: the same sweeps, seeded with launch r - 1's states (ping-pong arrays: a workgroup reads its left neighbour's last
// subsequence as the previous launch left it).  changed[r] is raised when a subsequence's result moves during launch r: a launch that raises
// nothing is the fixed point.
constexpr int kSweeps = 8;                    // measured: 4 / 8 / 16 sweeps per launch = 629 / 600 / 694 us of rounds per 12 MP 4:2:0 frame
template <bool RST>
__global__ __launch_bounds__(256) void k_huff_round(const HuffParPlan* __restrict__ P, const uint32_t* __restrict__ words, const HuffParState* __restrict__ in,
                                                     HuffParState* __restrict__ out, HuffParState* __restrict__ used, uint32_t* __restrict__ nblk,
                                                     uint32_t* __restrict__ changed, int round, const uint32_t* __restrict__ seg_end, const uint32_t* __restrict__ seg_hint)
{
    __shared__ HuffLds L;
    __shared__ HuffParState Es[257];                    // Es[0]: the end state of the subsequence left of the workgroup (as the previous launch left
                                                        // it); Es[1 + t]: lane t's, updated from sweep to sweep
    const uint32_t first = blockIdx.x * 256u;
    const int t = (int)threadIdx.x, i = (int)(first + threadIdx.x);
    const bool mine = i < P->nsub;
    HuffParState cur = { 0u, 0u }, u = { 0xffffffffu, 0xffffffffu };      // this subsequence's end state; the state it was last entered in
    uint32_t n = 0;
    if (round > 0 && mine) { cur = in[i]; u = used[i]; n = nblk[i]; }
    // (launch 0 knows nothing of the subsequence left of the workgroup: lane 0 keeps the guess it started from -- its own first bit, a block
    // beginning there -- through all sweeps; the true start of the stream is that guess exactly)
    if (t == 0) Es[0] = (round > 0 && first > 0) ? in[first - 1] : HuffParState{ first * (uint32_t)kSubBits, 0u };
    Es[1 + t] = cur;
    __syncthreads();
    // A subsequence entered in the state it was entered in the last time it was decoded ends as it ended then: only the subsequences whose
    // predecessor moved have work, and a sweep lasts as long as the slowest lane THAT HAS WORK.  A correction travels one subsequence per sweep;
    // up to kSweeps sweeps run inside one launch (window and tables are staged once), so a chain inside a workgroup's 256 subsequences is
    // followed without a launch per link.  A workgroup none of whose lanes has work does not even stage its window.
    bool staged = false, moved = false;
    for (int sweep = 0; sweep < kSweeps; sweep++) {
        HuffParState s0;
        if (round == 0 && sweep == 0) { s0.p = (uint32_t)i * (uint32_t)kSubBits; s0.ck = 0u; }
        else s0 = Es[t];
        const bool active = mine && (s0.p != u.p || s0.ck != u.ck) && s0.p >= (uint32_t)i * (uint32_t)kSubBits;      // (a start left of the subsequence cannot be: never decode from one)
        if (!__syncthreads_or(active ? 1 : 0)) break;
        if (!staged) { huff_stage(P, words, L, first); staged = true; }
        HuffParState e = cur;
        if (active) {
            uint32_t nn;
            huff_par_sub<RST>(*P, L.tab, L.words, first * (uint32_t)kSubBits, i, s0, e, nn, seg_end, RST ? seg_hint[i] : 0u);
            moved = moved || e.p != cur.p || e.ck != cur.ck || nn != n;
            n = nn; u = s0;
        }
        __syncthreads();                                // every lane has read its left neighbour's state of this sweep
        if (active) { cur = e; Es[1 + t] = e; }
        __syncthreads();
    }
    if (!mine) return;
    out[i] = cur; nblk[i] = n; used[i] = u;
    if (moved && round > 0) changed[round] = 1u;
}

// the write pass: coefficients into the dense array (zeroed before; DC values as differences), the last subsequence's end for the host to check
template <bool RST>
__global__ __launch_bounds__(256) void k_huff_write(const HuffParPlan* __restrict__ P, const uint32_t* __restrict__ words, const HuffParState* __restrict__ st,
                                                     const uint32_t* __restrict__ first_block, int16_t* __restrict__ coef, HuffParResult* __restrict__ res,
                                                     const uint32_t* __restrict__ seg_end, const uint32_t* __restrict__ seg_hint)
{
    __shared__ HuffLds L;
    const uint32_t first = blockIdx.x * 256u;
    huff_stage(P, words, L, first);
    const int i = (int)(first + threadIdx.x);
    if (i >= P->nsub) return;
    HuffParState s0 = { 0u, 0u };
    if (i > 0) s0 = st[i - 1];
    HuffParState e; uint32_t ge;
    huff_par_write<RST>(*P, L.tab, L.words, first * (uint32_t)kSubBits, i, s0, first_block[i], coef, e, ge, seg_end, RST ? seg_hint[i] : 0u, &res->bad);
    if (i == P->nsub - 1) { res->g_end = ge; res->p = e.p; res->ck = e.ck; }
}

// ---- running sums over many workgroups: MODE 0 = blocks completed per subsequence -> first block of each subsequence (exclusive);
// MODE 1 = DC differences of a component in scan order -> DC values (inclusive, jdhuff.c last_dc_val).  Three small launches each:
// per-workgroup totals of 1024 elements, an exclusive scan of the totals, the elements again with their workgroup's offset.
constexpr int kScanPer = 4, kScanTile = 256 * kScanPer;
template <int MODE>
__device__ inline int scan_elem(const HuffParPlan* __restrict__ P, const uint32_t* __restrict__ nblk, const int16_t* __restrict__ coef, int comp, int j)
{
    return MODE == 0 ? (int)nblk[j] : (int)coef[huff_par_comp_block(*P, comp, (uint32_t)j)];          // (MODE 2 reads what MODE 1 reads)
}
template <int MODE>
__global__ __launch_bounds__(256) void k_scan_totals(const HuffParPlan* __restrict__ P, const uint32_t* __restrict__ nblk, const int16_t* __restrict__ coef,
                                                      int* __restrict__ totals, int stride)
{
    __shared__ int red[256];
    const int comp = (int)blockIdx.y, N = MODE == 0 ? P->nsub : P->cblocks[comp], t = (int)threadIdx.x;
    const int j0 = ((int)blockIdx.x * 256 + t) * kScanPer;
    int sum = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; e++) if (j0 + e < N) sum += scan_elem<MODE>(P, nblk, coef, comp, j0 + e);
    red[t] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (t < d) red[t] += red[t + d]; __syncthreads(); }
    if (t == 0) totals[comp * stride + (int)blockIdx.x] = red[0];
}
// exclusive scan of each row of `totals` in place; one workgroup per row
__global__ __launch_bounds__(1024) void k_scan_top(int* __restrict__ totals, int stride, int n0, int n1, int n2)
{
    __shared__ int part[1024];
    const int row = (int)blockIdx.x, n = row == 0 ? n0 : row == 1 ? n1 : n2, t = (int)threadIdx.x, per = (n + 1023) / 1024;
    int* v = totals + row * stride;
    const int a = t * per < n ? t * per : n, b = a + per < n ? a + per : n;
    int sum = 0;
    for (int j = a; j < b; j++) sum += v[j];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int x = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int j = a; j < b; j++) { const int x = v[j]; v[j] = run; run += x; }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_scan_apply(const HuffParPlan* __restrict__ P, const uint32_t* __restrict__ nblk, int16_t* __restrict__ coef,
                                                     uint32_t* __restrict__ first_block, const int* __restrict__ totals, int stride, int dc_stride)
{
    __shared__ int part[256];
    const int comp = (int)blockIdx.y, N = MODE == 0 ? P->nsub : P->cblocks[comp], t = (int)threadIdx.x;
    const int j0 = ((int)blockIdx.x * 256 + t) * kScanPer;
    int v[kScanPer], sum = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; e++) { v[e] = j0 + e < N ? scan_elem<MODE>(P, nblk, coef, comp, j0 + e) : 0; sum += v[e]; }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int x = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int run = totals[comp * stride + (int)blockIdx.x] + (t ? part[t - 1] : 0);
#pragma unroll
    for (int e = 0; e < kScanPer; e++) {
        if (j0 + e >= N) break;
        if (MODE == 0) { first_block[j0 + e] = (uint32_t)run; run += v[e]; }
        else if (MODE == 1) { run += v[e]; coef[huff_par_comp_block(*P, comp, (uint32_t)(j0 + e))] = (int16_t)run; }
        else { run += v[e]; reinterpret_cast<int*>(first_block)[comp * dc_stride + j0 + e] = run; }          // MODE 2: the running sum itself (restart intervals)
    }
}

bool hip_ok(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    set_error(std::string("jpeg device: ") + what + ": " + hipGetErrorString(e));
    return false;
}

}  // namespace

JpegDevice::~JpegDevice()
{
    for (auto& s : slot_) {
        if (s.done) { (void)hipEventSynchronize((hipEvent_t)s.done); (void)hipEventDestroy((hipEvent_t)s.done); }
        if (s.host) (void)hipHostFree(s.host);
    }
    if (dev_) (void)hipFree(dev_);
    if (planes_) (void)hipFree(planes_);
    if (huff_) (void)hipFree(huff_);
    if (res_host_) (void)hipHostFree(res_host_);
}

// Host side of a frame, step 1 (calling thread): the geometry, and a pinned buffer that the previous upload out of it has left
bool JpegDevice::prepare(int i, const uint8_t* data, size_t len, int rows, int cols)
{
    if (i < 0 || i >= kSlots) { set_error("jpeg device: no such staging buffer"); return false; }
    Slot& s = slot_[i];
    s.staged = false;
    if (!data) { set_error("jpeg device: null buffer"); return false; }
    if (!jpeg_frame_info(data, len, s.f)) return false;
    if (rows > 0 && (s.f.rows != rows || s.f.cols != cols)) { set_error("jpeg device: the output buffer does not have the image's size"); return false; }
    if (s.f.coef_count >= (1ull << 31)) { set_error("jpeg device: image too large"); return false; }
    size_t need = kHeaderBytes + s.f.coef_count * sizeof(int16_t);
    need = std::max(need, kHeaderBytes + kPlanBytes + len + 1024 + (len / 2 + len / 64 + 16) * 4);          // the parallel pass's layout: header, plan, the scan's bytes, restart tables
    s.data = data; s.len = len; s.par = false;
    if (s.done && s.used && !hip_ok(hipEventSynchronize((hipEvent_t)s.done), "wait for the staging buffer")) return false;
    s.used = false;
    if (s.cap < need) {
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr; s.cap = 0;
        if (!hip_ok(hipHostMalloc(&s.host, need, hipHostMallocDefault), "pinned staging buffer")) return false;
        s.cap = need;
    }
    if (!s.done) { hipEvent_t e; if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event")) return false; s.done = e; }
    return true;
}

// ... step 2 (any thread; no HIP call): markers + Huffman into the pinned buffer.  An error message stays in the slot.
bool JpegDevice::entropy(int i, const uint8_t* data, size_t len)
{
    Slot& s = slot_[i];
    static const bool host_huffman = exp_env("PF_JPEG_HOST_HUFFMAN") != nullptr;         // experiments library (A/B, tests): the serial pass on the host for every stream
    // a recent stream of this consumer did not settle: its neighbours will not either (several host threads come here in a batch: the counter
    // is taken down by compare-exchange, never below zero)
    bool skip = false;
    for (int v = skip_par_.load(std::memory_order_relaxed); v > 0 && !skip; ) skip = skip_par_.compare_exchange_weak(v, v - 1, std::memory_order_relaxed);
    if (!host_huffman && !skip) {
        // a stream the parallel pass takes: its scan's bytes (stuffing removed) and the plan go to the GPU, nothing else happens here
        HuffParPlan* plan = (HuffParPlan*)((char*)s.host + kHeaderBytes);
        uint8_t* bits = (uint8_t*)s.host + kHeaderBytes + kPlanBytes;
        std::vector<uint32_t> seg;
        if (jpeg_scan_plan(data, len, s.f, *plan, bits, s.cap - kHeaderBytes - kPlanBytes, &s.par_bytes, &seg)) {
            // restart intervals: the segments' ends and, per subsequence, the first segment that ends after its first bit, behind the scan's bytes
            const size_t wbytes = (s.par_bytes + 16 + 255) & ~(size_t)255, S = (size_t)plan->nsub;
            s.aux_words = 0;
            if (plan->rst_blocks) {
                if (kHeaderBytes + kPlanBytes + wbytes + (seg.size() + 1 + S) * 4 <= s.cap) {
                    uint32_t* aux = (uint32_t*)((char*)s.host + kHeaderBytes + kPlanBytes + wbytes);
                    std::memcpy(aux, seg.data(), seg.size() * 4);
                    aux[seg.size()] = plan->nbits;                                          // a guard behind the last end
                    uint32_t* hint = aux + seg.size() + 1;
                    uint32_t sg = 0;
                    for (size_t k = 0; k < S; k++) { while (seg[sg] <= (uint32_t)k * (uint32_t)kSubBits) sg++; hint[k] = sg; }
                    s.aux_words = seg.size() + 1 + S;
                    s.par = true; s.staged = true; return true;
                }
            } else { s.par = true; s.staged = true; return true; }
        }
    }
    s.staged = jpeg_entropy_decode(data, len, s.f, (int16_t*)((char*)s.host + kHeaderBytes), s.f.coef_count);
    if (!s.staged) s.err = last_error();
    return s.staged;
}

// The Huffman pass of slot i on the GPU (jpeg_huff_par.hpp): coefficients into dev_ behind the header.  false: the rounds did not settle or
// the write pass did not end on the frame's last block -- the caller falls back to the serial pass.
bool JpegDevice::huffman_on_device(int i, void* stream, size_t coef_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    Slot& s = slot_[i];
    const HuffParPlan* hp = (const HuffParPlan*)((char*)s.host + kHeaderBytes);
    const size_t S = (size_t)hp->nsub, wbytes = (s.par_bytes + 16 + 255) & ~(size_t)255;
    const bool rst = hp->rst_blocks != 0;
    const size_t aux_bytes = rst ? ((s.aux_words * 4 + 255) & ~(size_t)255) : 0;
    const size_t o_st0 = kPlanBytes + wbytes + aux_bytes, o_st1 = o_st0 + S * 8, o_used = o_st1 + S * 8, o_nblk = o_used + S * 8, o_first = o_nblk + S * 4,
                 o_flags = (o_first + S * 4 + 255) & ~(size_t)255, o_tot = o_flags + kMaxRounds * 4 + 256;
    int cmax = 0;
    for (int c = 0; c < hp->ncomp; c++) cmax = std::max(cmax, hp->cblocks[c]);
    const int stride = (int)std::max((S + kScanTile - 1) / kScanTile, (size_t)(cmax + kScanTile - 1) / kScanTile) + 1;
    size_t dc_total = 0; int dc_stride = 0;
    if (rst) { dc_stride = cmax; dc_total = (size_t)cmax * 3 * 4; }
    const size_t o_dc = (o_tot + (size_t)stride * 3 * 4 + 255) & ~(size_t)255;
    const size_t total = o_dc + dc_total;
    if (huff_cap_ < total) {
        if (huff_) (void)hipFree(huff_);
        huff_ = nullptr; huff_cap_ = 0;
        if (!hip_ok(hipMalloc(&huff_, total + total / 4), "Huffman work buffer")) return false;
        huff_cap_ = total + total / 4;
    }
    if (!res_host_ && !hip_ok(hipHostMalloc(&res_host_, kMaxRounds * 4 + 256, hipHostMallocDefault), "result buffer")) return false;
    char* hb = (char*)huff_;
    const HuffParPlan* P = (const HuffParPlan*)hb;
    const uint32_t* words = (const uint32_t*)(hb + kPlanBytes);
    HuffParState* stt[2] = { (HuffParState*)(hb + o_st0), (HuffParState*)(hb + o_st1) };
    HuffParState* used = (HuffParState*)(hb + o_used);
    uint32_t* nblk = (uint32_t*)(hb + o_nblk); uint32_t* first = (uint32_t*)(hb + o_first);
    uint32_t* changed = (uint32_t*)(hb + o_flags); HuffParResult* res = (HuffParResult*)(hb + o_flags + kMaxRounds * 4);
    int* totals = (int*)(hb + o_tot);
    int* dcsum = (int*)(hb + o_dc);
    const uint32_t* seg_end = rst ? (const uint32_t*)(hb + kPlanBytes + wbytes) : nullptr;
    const uint32_t* seg_hint = rst ? seg_end + hp->nseg + 1 : nullptr;
    if (!hip_ok(hipMemcpyAsync(hb, (char*)s.host + kHeaderBytes, rst ? kPlanBytes + wbytes + s.aux_words * 4 : kPlanBytes + s.par_bytes + 16, hipMemcpyHostToDevice, st), "scan upload")) return false;
    if (!hip_ok(hipMemsetAsync(changed, 0, kMaxRounds * 4 + 256, st), "flags")) return false;
    if (!hip_ok(hipMemsetAsync((char*)dev_ + kHeaderBytes, 0, coef_bytes, st), "coefficient clear")) return false;
    const dim3 grid((unsigned)((S + 255) / 256));
    if (rst) hipLaunchKernelGGL((k_huff_round<true>), grid, dim3(256), 0, st, P, words, (const HuffParState*)stt[1], stt[0], used, nblk, changed, 0, seg_end, seg_hint);
    else hipLaunchKernelGGL((k_huff_round<false>), grid, dim3(256), 0, st, P, words, (const HuffParState*)stt[1], stt[0], used, nblk, changed, 0, seg_end, seg_hint);
    // Launches (of up to kSweeps sweeps each) go out in groups and the flags are read back after each group (a read-back costs a stream
    // synchronisation, a launch past the fixed point costs little: no workgroup has work).  Consecutive keyframes of a camera settle after about
    // the same number of launches: the first group is the previous frame's count plus one, the following groups two launches each.
    int cur = 0, round = 0; bool settled = S == 1;
    int group = std::max(settle_hint_ + 1, 2);
    while (!settled && round < (int)kMaxRounds - 1) {
        group = std::min(group, (int)kMaxRounds - 1 - round);          // never past the flags array, and a late settler cannot switch the loop off for its successors
        for (int r = 0; r < group; r++) {
            round++;
            if (rst) hipLaunchKernelGGL((k_huff_round<true>), grid, dim3(256), 0, st, P, words, (const HuffParState*)stt[cur], stt[cur ^ 1], used, nblk, changed, round, seg_end, seg_hint);
            else hipLaunchKernelGGL((k_huff_round<false>), grid, dim3(256), 0, st, P, words, (const HuffParState*)stt[cur], stt[cur ^ 1], used, nblk, changed, round, seg_end, seg_hint);
            cur ^= 1;
        }
        if (!hip_ok(hipMemcpyAsync(res_host_, changed, kMaxRounds * 4, hipMemcpyDeviceToHost, st), "flags read-back")) return false;
        if (!hip_ok(hipStreamSynchronize(st), "Huffman rounds")) return false;
        const uint32_t* fl = (const uint32_t*)res_host_;
        for (int r = 1; r <= round && !settled; r++) if (fl[r] == 0) { settled = true; settle_hint_ = r; }
        group = 2;
    }
    last_rounds_ = round;
    if (!settled) { skip_par_.store(15, std::memory_order_relaxed); settle_hint_ = 4; return false; }          // (streams of quality 99 noise do this: ~250 rounds are 11 ms thrown away)
    int16_t* dcoef = (int16_t*)((char*)dev_ + kHeaderBytes);
    const int tiles_s = (int)((S + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL((k_scan_totals<0>), dim3((unsigned)tiles_s, 1), dim3(256), 0, st, P, (const uint32_t*)nblk, (const int16_t*)dcoef, totals, stride);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, totals, stride, tiles_s, 0, 0);
    hipLaunchKernelGGL((k_scan_apply<0>), dim3((unsigned)tiles_s, 1), dim3(256), 0, st, P, (const uint32_t*)nblk, dcoef, first, (const int*)totals, stride, 0);
    if (rst) hipLaunchKernelGGL((k_huff_write<true>), grid, dim3(256), 0, st, P, words, (const HuffParState*)stt[cur], (const uint32_t*)first, dcoef, res, seg_end, seg_hint);
    else hipLaunchKernelGGL((k_huff_write<false>), grid, dim3(256), 0, st, P, words, (const HuffParState*)stt[cur], (const uint32_t*)first, dcoef, res, seg_end, seg_hint);
    int tc[3] = { 0, 0, 0 }, tmax = 0;
    for (int c = 0; c < hp->ncomp; c++) { tc[c] = (hp->cblocks[c] + kScanTile - 1) / kScanTile; tmax = std::max(tmax, tc[c]); }
    hipLaunchKernelGGL((k_scan_totals<1>), dim3((unsigned)tmax, (unsigned)hp->ncomp), dim3(256), 0, st, P, (const uint32_t*)nblk, (const int16_t*)dcoef, totals, stride);
    hipLaunchKernelGGL(k_scan_top, dim3((unsigned)hp->ncomp), dim3(1024), 0, st, totals, stride, tc[0], tc[1], tc[2]);
    if (rst) {
        hipLaunchKernelGGL((k_scan_apply<2>), dim3((unsigned)tmax, (unsigned)hp->ncomp), dim3(256), 0, st, P, (const uint32_t*)nblk, dcoef, (uint32_t*)dcsum, (const int*)totals, stride, dc_stride);
        hipLaunchKernelGGL(k_dc_restart, dim3((unsigned)((cmax + 255) / 256), (unsigned)hp->ncomp), dim3(256), 0, st, P, (const int*)dcsum, dc_stride, dcoef);
    } else
        hipLaunchKernelGGL((k_scan_apply<1>), dim3((unsigned)tmax, (unsigned)hp->ncomp), dim3(256), 0, st, P, (const uint32_t*)nblk, dcoef, first, (const int*)totals, stride, 0);
    if (!hip_ok(hipMemcpyAsync(res_host_, res, sizeof(HuffParResult), hipMemcpyDeviceToHost, st), "result read-back")) return false;
    if (!hip_ok(hipStreamSynchronize(st), "Huffman write pass")) return false;
    const HuffParResult* r = (const HuffParResult*)res_host_;
    return r->g_end == (uint32_t)hp->total_blocks && r->ck == 0 && hp->nbits - r->p < 8 && r->bad == 0;
}

// ... step 3 (calling thread): the coefficients get to the GPU -- decoded there, or uploaded -- and the two kernels follow on `stream`
bool JpegDevice::submit(int i, uint8_t* dev_bgr, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    Slot& s = slot_[i];
    if (!s.staged) { set_error(s.err.empty() ? "jpeg device: frame was not staged" : s.err); return false; }
    if (!dev_bgr) { set_error("jpeg device: null buffer"); return false; }
    s.staged = false;
    const JpegFrame& f = s.f;
    const int rows = f.rows, cols = f.cols;
    const size_t coef_bytes = f.coef_count * sizeof(int16_t), need = kHeaderBytes + coef_bytes;
    JpegDevFrame h;
    std::memset(&h, 0, sizeof(h));
    h.rows = f.rows; h.cols = f.cols; h.ncomp = f.ncomp; h.ycc = f.ycc ? 1 : 0;
    size_t plane_bytes = 0; unsigned blocks = 0;
    for (int c = 0; c < f.ncomp; c++) {
        const JpegComponent& k = f.c[c];
        JpegDevComp& o = h.c[c];
        o.bw = k.bw; o.bh = k.bh; o.w = k.w; o.ht = k.ht; o.he = f.hmax / k.h; o.ve = f.vmax / k.v; o.stride = k.bw * 8;
        o.coef_off = (unsigned)k.coef_off; o.plane_off = (unsigned)plane_bytes;
        std::memcpy(o.q, k.q, sizeof(o.q));
        h.block_first[c] = blocks;
        blocks += (unsigned)(k.bw * k.bh);
        plane_bytes += (size_t)k.bw * 8 * k.bh * 8;
    }
    for (int c = f.ncomp; c < 4; c++) h.block_first[c] = blocks;
    if (plane_bytes >= (1ull << 31)) { set_error("jpeg device: image too large"); return false; }

    if (dev_cap_ < need) {
        if (dev_) (void)hipFree(dev_);
        dev_ = nullptr; dev_cap_ = 0;
        if (!hip_ok(hipMalloc(&dev_, need), "coefficient buffer")) return false;
        dev_cap_ = need;
    }
    if (planes_cap_ < plane_bytes) {
        if (planes_) (void)hipFree(planes_);
        planes_ = nullptr; planes_cap_ = 0;
        if (!hip_ok(hipMalloc(&planes_, plane_bytes), "plane buffer")) return false;
        planes_cap_ = plane_bytes;
    }
    bool on_device = false;
    if (s.par) {
        on_device = huffman_on_device(i, stream, coef_bytes);
        par_frames_ += on_device ? 1 : 0;
        if (!on_device) {                                  // the serial pass after all (the stream and its length were kept by prepare)
            fallback_frames_++;
            if (!hip_ok(hipStreamSynchronize(st), "before the fallback")) return false;
            JpegFrame f2;
            if (!jpeg_entropy_decode(s.data, s.len, f2, (int16_t*)((char*)s.host + kHeaderBytes), f.coef_count)) return false;
        }
    }
    std::memcpy(s.host, &h, sizeof(h));
    if (!hip_ok(hipMemcpyAsync(dev_, s.host, on_device ? kHeaderBytes : need, hipMemcpyHostToDevice, st), "coefficient upload")) return false;
    if (!hip_ok(hipEventRecord((hipEvent_t)s.done, st), "event record")) return false;
    s.used = true;
    const JpegDevFrame* dh = (const JpegDevFrame*)dev_;
    const int16_t* dcoef = (const int16_t*)((char*)dev_ + kHeaderBytes);
    hipLaunchKernelGGL(k_jpeg_idct, dim3((blocks + 31) / 32), dim3(256), 0, st, dh, dcoef, (uint8_t*)planes_);
    // the eight-pixel form for the shapes cameras write, the general one otherwise
    const JpegDevComp* hc = h.c;
    const bool camera = f.ncomp == 3 && (cols & 7) == 0 && hc[0].he == 1 && hc[0].ve == 1 && hc[1].he == hc[2].he && hc[1].ve == hc[2].ve &&
                        ((hc[1].he == 1 && hc[1].ve == 1) || (hc[1].he == 2 && hc[1].ve <= 2 && hc[1].w > 2));
    const dim3 g8((unsigned)((cols + 2047) / 2048), (unsigned)rows);
    if (camera && hc[1].he == 1) hipLaunchKernelGGL((k_jpeg_colour8<1, 1>), g8, dim3(256), 0, st, dh, (const uint8_t*)planes_, dev_bgr);
    else if (camera && hc[1].ve == 1) hipLaunchKernelGGL((k_jpeg_colour8<2, 1>), g8, dim3(256), 0, st, dh, (const uint8_t*)planes_, dev_bgr);
    else if (camera) hipLaunchKernelGGL((k_jpeg_colour8<2, 2>), g8, dim3(256), 0, st, dh, (const uint8_t*)planes_, dev_bgr);
    else hipLaunchKernelGGL(k_jpeg_colour, dim3((unsigned)((cols + 1023) / 1024), (unsigned)rows), dim3(256), 0, st, dh, (const uint8_t*)planes_, dev_bgr);
    last_ = { coef_bytes, plane_bytes, (size_t)rows * cols * 3 };
    return hip_ok(hipGetLastError(), "kernel launch");
}

bool JpegDevice::decode_to(const uint8_t* data, size_t len, uint8_t* dev_bgr, int rows, int cols, void* stream)
{
    if (!dev_bgr) { set_error("jpeg device: null buffer"); return false; }
    const int i = next_; next_ ^= 1;              // two buffers: this frame's Huffman pass overlaps the previous frame's upload
    if (!prepare(i, data, len, rows, cols)) return false;
    if (!entropy(i, data, len)) { set_error(slot_[i].err); return false; }
    return submit(i, dev_bgr, stream);
}

int JpegDevice::stage_one(const uint8_t* data, size_t len, unsigned char* ok)
{
    const int i = next_; next_ ^= 1;
    *ok = prepare(i, data, len, 0, 0) ? 1 : 0;
    if (!*ok) slot_[i].err = last_error();
    else *ok = entropy(i, data, len) ? 1 : 0;
    return i;
}

// n frames: step 1 for each, then their Huffman passes side by side on host threads; the caller submits them in the order it wants
bool JpegDevice::stage_batch(int n, const uint8_t* const* data, const size_t* len, int rows, int cols, int threads, unsigned char* ok)
{
    if (n < 1 || n > kSlots) { set_error("jpeg device: a batch holds 1 to 16 frames"); return false; }
    for (int i = 0; i < n; i++) {
        ok[i] = prepare(i, data[i], len[i], rows, cols) ? 1 : 0;
        if (!ok[i]) slot_[i].err = last_error();
    }
    if (threads <= 0 || threads > n) threads = n;
    std::atomic<int> next(0);
    auto work = [&]() { for (int i = next++; i < n; i = next++) if (ok[i]) ok[i] = entropy(i, data[i], len[i]) ? 1 : 0; };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    next_ = 0;
    return true;
}

}  // namespace pf
