// jpeg_decode.cpp -- input side of the file driver: the reference reads every keyframe with cv::imread(imgfile)
// (backup/map2dfusion.cpp:129-132), which for .jpg is libjpeg with its defaults (JDCT_ISLOW, fancy upsampling, YCbCr -> RGB
// with the 16-bit tables) and a swap to BGR.  libjpeg is not part of /root/reference; the arithmetic below restates the
// published algorithms of libjpeg / libjpeg-turbo (jidctint.c, jdsample.c, jdcolor.c, jdhuff.c, jdphuff.c) and is pinned
// against libjpeg-turbo through Pillow (tests/golden/make_jpeg_vectors.py, tests/test_jpeg.py): byte-equal on every fixture.
// Supported: 8-bit baseline / extended sequential / progressive Huffman JPEG, 1 or 3 components, restart intervals, any
// integral sampling ratio.  Not supported (false + pf_last_error): arithmetic coding, lossless, 12-bit, 4 components.
// Host code only; nothing here touches the device.
#include "jpeg_decode.hpp"
#include "jpeg_huff_par.hpp"
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace pf {
namespace {

const uint8_t kZigzag[64 + 16] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };      // a run past the end lands on 63, as in libjpeg

constexpr int kLookBits = 10, kLookSize = 1 << kLookBits;
struct Huff {
    bool     set = false;
    uint8_t  vals[256];
    int      maxcode[18];          // largest code of each length, -1 when none
    int      valoff[17];           // vals index of the first code of a length minus that code
    uint16_t look[kLookSize];      // prefix of kLookBits -> (length << 8) | symbol, 0 when the code is longer (9 / 10 / 11 bits measured: 60.8 / 53.1 / 54.6 ms for a 4.5 MB stream)
    int16_t  fast[kLookSize];      // AC use: a prefix holding a whole (run, size) code AND its value bits -> (value << 8) | (run << 4) | bits used
    bool build(const uint8_t bits[17], const uint8_t* v, int nv)
    {
        std::memcpy(vals, v, (size_t)nv);
        std::memset(look, 0, sizeof(look));
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valoff[l] = k - code;
            for (int i = 0; i < bits[l]; i++, k++, code++) {
                if (code >= (1 << l)) return false;
                if (l <= kLookBits) {
                    const int lo = code << (kLookBits - l);
                    for (int f = 0; f < (1 << (kLookBits - l)); f++) look[lo + f] = (uint16_t)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        // codes short enough that the coefficient's own bits fit into the same prefix: one look-up instead of code, bits, sign extension
        for (int i = 0; i < kLookSize; i++) {
            fast[i] = 0;
            const int len = look[i] >> 8, rs = look[i] & 255, run = rs >> 4, mag = rs & 15;
            if (!len || !mag || len + mag > kLookBits) continue;
            int v = ((i << len) & (kLookSize - 1)) >> (kLookBits - mag);
            if (v < (1 << (mag - 1))) v += 1 - (1 << mag);
            if (v >= -128 && v <= 127) fast[i] = (int16_t)(v * 256 + run * 16 + len + mag);
        }
        set = true;
        return k == nv;
    }
};

// Entropy-coded segment reader: removes the stuffed zero after 0xFF, stops at a marker and hands out zero bits after it
// (libjpeg: "premature end of data segment", the rest of the interval stays zero).
struct Bits {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int n = 0, fake = 0, marker = 0; bool starved = false;
    Bits(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
    __attribute__((always_inline)) void fill()
    {
        if (!marker && end - p >= 8) {                    // eight bytes without a 0xFF among them: take as many as fit at once
            uint64_t v;
            std::memcpy(&v, p, 8);
            v = __builtin_bswap64(v);
            const uint64_t inv = ~v;
            if (!((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull)) {
                const int take = (64 - n) >> 3, rest = (64 - n) & 7;
                if (n < 64) acc |= (v >> n) & ~((1ull << rest) - 1);
                p += take; n += take * 8;
                return;
            }
        }
        while (n <= 56) {
            int b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    while (p < end && *p == 0xFF) p++;
                    if (p >= end) { marker = 0xD9; b = 0; fake += 8; }
                    else if (*p == 0) p++;
                    else { marker = *p++; b = 0; fake += 8; }
                }
            } else {
                if (!marker) marker = 0xD9;
                fake += 8;
            }
            acc |= (uint64_t)b << (56 - n);
            n += 8;
        }
    }
    __attribute__((always_inline)) int peek(int k) { if (n < k) fill(); return (int)(acc >> (64 - k)); }
    __attribute__((always_inline)) void drop(int k) { acc <<= k; n -= k; if (n < fake) { starved = true; fake = n; } }
    __attribute__((always_inline)) int get(int k) { if (!k) return 0; const int v = peek(k); drop(k); return v; }
    __attribute__((always_inline)) int bit() { return get(1); }
    __attribute__((always_inline)) int symbol(const Huff& h)
    {
        const int pre = peek(16);
        const uint16_t e = h.look[pre >> (16 - kLookBits)];
        if (e) { drop(e >> 8); return e & 255; }
        int l = kLookBits + 1;
        while (l <= 16 && (pre >> (16 - l)) > h.maxcode[l]) l++;
        if (l > 16) { drop(16); return 0; }              // libjpeg: "corrupt JPEG data: bad Huffman code", symbol 0
        drop(l);
        return h.vals[((pre >> (16 - l)) + h.valoff[l]) & 255];
    }
    // restart: drop the partial byte, step over RSTn
    void restart()
    {
        if (!marker) {                                    // padding bits only; look for the marker in the stream
            while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) p++;
            if (p + 1 < end) p += 2;
        } else if (marker < 0xD0 || marker > 0xD7) return;    // a different marker: leave it to the caller, bits stay zero
        acc = 0; n = 0; fake = 0; marker = 0; starved = false;
    }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0;
    int bw = 0, bh = 0;                  // blocks held (whole MCUs)
    int w = 0, ht = 0;                   // downsampled_width / _height: the samples that are real
    int td = 0, ta = 0, pred = 0;
    int16_t* coef = nullptr;             // bw*bh blocks of 64, natural order
    std::vector<int16_t> own;            // ... held here unless the caller brought the storage
    std::vector<uint8_t> plane;          // bw*8 x bh*8 after the IDCT
};

// jidctint.c (jpeg_idct_islow): CONST_BITS 13, PASS1_BITS 2.  The zero-AC shortcuts of the original give the same values
// as the full butterfly and are left out.
// All of it in 32-bit arithmetic that wraps (unsigned, shifted as signed): libjpeg's C code keeps these sums in a long and its SIMD code in
// 32-bit lanes; the two agree wherever the stream is one an encoder writes, and a corrupt stream gets the same pixels here and in
// jpeg_device.hip instead of undefined behaviour.
typedef uint32_t u32;
inline int descale(u32 x, int n) { return (int)(x + (1u << (n - 1))) >> n; }
inline uint8_t limit_idct(int x)
{
    x &= 1023;                           // RANGE_MASK; the table of jdmaster.c prepare_range_limit_table past CENTERJSAMPLE
    return (uint8_t)(x < 128 ? x + 128 : x < 512 ? 255 : x < 896 ? 0 : x - 896);
}

// one 8-point pass of jidctint.c on v[0..7]; `sh`: CONST_BITS - PASS1_BITS for the columns, CONST_BITS + PASS1_BITS + 3 for the rows
inline void idct8(const u32 v[8], int out[8], int sh)
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
    out[0] = descale(tmp10 + tmp3, sh); out[7] = descale(tmp10 - tmp3, sh);
    out[1] = descale(tmp11 + tmp2, sh); out[6] = descale(tmp11 - tmp2, sh);
    out[2] = descale(tmp12 + tmp1, sh); out[5] = descale(tmp12 - tmp1, sh);
    out[3] = descale(tmp13 + tmp0, sh); out[4] = descale(tmp13 - tmp0, sh);
}

void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride)
{
    int ws[64];
    for (int c = 0; c < 8; c++) {
        u32 v[8]; int o[8];
        for (int r = 0; r < 8; r++) v[r] = (u32)((int)in[8 * r + c] * (int)q[8 * r + c]);        // DEQUANTIZE: |product| < 2^31
        idct8(v, o, 11);
        for (int r = 0; r < 8; r++) ws[8 * r + c] = o[r];
    }
    for (int r = 0; r < 8; r++) {
        u32 v[8]; int o[8];
        for (int k = 0; k < 8; k++) v[k] = (u32)ws[8 * r + k];
        idct8(v, o, 18);
        uint8_t* dst = out + (size_t)r * stride;
        for (int k = 0; k < 8; k++) dst[k] = limit_idct(o[k]);
    }
}

struct Decoder {
    const uint8_t* d; size_t n; size_t pos = 0;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0, scans = 0;
    bool progressive = false, have_frame = false, jfif = false, adobe = false; int adobe_transform = 0;
    Comp comp[3];
    uint16_t qt[4][64]; bool qt_set[4] = { false, false, false, false };
    Huff dc[4], ac[4];
    std::string err;
    int16_t* store = nullptr; size_t store_cap = 0;      // optional caller storage for the coefficients of all components
    bool plan_only = false; size_t sos_s = 0, sos_e = 0;    // jpeg_scan_plan: stop at the first scan header, before any storage is touched

    Decoder(const uint8_t* data, size_t len) : d(data), n(len) {}
    size_t coef_count() const
    {
        size_t t = 0;
        for (int c = 0; c < ncomp; c++) t += (size_t)comp[c].bw * comp[c].bh * 64;
        return t;
    }
    bool fail(const char* m) { err = std::string("jpeg: ") + m; return false; }
    int u16(size_t at) const { return (d[at] << 8) | d[at + 1]; }

    // next marker code at or after pos (0 when the data ends)
    int next_marker()
    {
        while (pos + 1 < n) {
            if (d[pos] != 0xFF) { pos++; continue; }
            const int m = d[pos + 1];
            if (m == 0 || m == 0xFF) { pos++; continue; }
            pos += 2;
            return m;
        }
        return 0;
    }

    bool read_headers(bool stop_at_frame)
    {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail("not a JPEG stream (no SOI)");
        pos = 2;
        for (;;) {
            const int m = next_marker();
            if (!m) return scans > 0 ? true : fail("no image in the stream");       // no EOI: libjpeg warns and delivers what it has
            if (m == 0xD9) return scans > 0 ? true : fail("EOI before any scan");
            if ((m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
            if (pos + 2 > n) return fail("truncated segment");
            const int len = u16(pos);
            if (len < 2 || pos + len > n) return fail("truncated segment");
            const size_t s = pos + 2, e = pos + len;
            pos = e;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) return fail("second frame header");
                if (e - s < 6) return fail("bad SOF");
                if (d[s] != 8) return fail("only 8-bit samples are supported");
                H = u16(s + 1); W = u16(s + 3); ncomp = d[s + 5];
                if (W <= 0 || H <= 0) return fail("empty image (DNL is not supported)");
                if ((unsigned long long)W * (unsigned long long)H > (1ull << 30)) return fail("image of more than 2^30 pixels");      // OpenCV's CV_IO_MAX_IMAGE_PIXELS
                if (ncomp != 1 && ncomp != 3) return fail("only 1- and 3-component images are supported");
                if (e - s < (size_t)(6 + 3 * ncomp)) return fail("bad SOF");
                progressive = m == 0xC2;
                for (int c = 0; c < ncomp; c++) {
                    Comp& k = comp[c];
                    k.id = d[s + 6 + 3 * c]; k.h = d[s + 7 + 3 * c] >> 4; k.v = d[s + 7 + 3 * c] & 15; k.tq = d[s + 8 + 3 * c] & 3;
                    if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4) return fail("bad sampling factor");
                    hmax = k.h > hmax ? k.h : hmax; vmax = k.v > vmax ? k.v : vmax;
                }
                for (int c = 0; c < ncomp; c++)
                    if (hmax % comp[c].h || vmax % comp[c].v) return fail("fractional sampling ratios are not supported");
                have_frame = true;
                mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
                for (int c = 0; c < ncomp; c++) {
                    Comp& k = comp[c];
                    k.bw = mcux * k.h; k.bh = mcuy * k.v;
                    k.w = (W * k.h + hmax - 1) / hmax; k.ht = (H * k.v + vmax - 1) / vmax;
                }
                if (stop_at_frame) return true;
                if (plan_only) continue;
                if (store && store_cap < coef_count()) return fail("coefficient storage too small");
                size_t off = 0;
                for (int c = 0; c < ncomp; c++) {
                    Comp& k = comp[c];
                    const size_t cnt = (size_t)k.bw * k.bh * 64;
                    if (store) { k.coef = store + off; std::memset(k.coef, 0, cnt * sizeof(int16_t)); }
                    else { k.own.assign(cnt, 0); k.coef = k.own.data(); }
                    off += cnt;
                }
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                return fail("unsupported coding process (lossless, hierarchical or arithmetic)");
            } else if (m == 0xC4) {
                size_t at = s;
                while (at < e) {
                    if (e - at < 17) return fail("bad DHT");
                    const int tc = d[at] >> 4, th = d[at] & 15;
                    uint8_t bits[17] = { 0 }; int nv = 0;
                    for (int l = 1; l <= 16; l++) { bits[l] = d[at + l]; nv += bits[l]; }
                    at += 17;
                    if (tc > 1 || th > 3 || nv > 256 || at + nv > e) return fail("bad DHT");
                    if (!tc)                                  // jdhuff.c jpeg_make_d_derived_tbl: a DC symbol is a bit count of at most 15
                        for (int i = 0; i < nv; i++) if (d[at + i] > 15) return fail("bad Huffman table");
                    if (!(tc ? ac[th] : dc[th]).build(bits, d + at, nv)) return fail("bad Huffman table");
                    at += nv;
                }
            } else if (m == 0xDB) {
                size_t at = s;
                while (at < e) {
                    const int pq = d[at] >> 4, tq = d[at] & 15;
                    at++;
                    if (pq > 1 || tq > 3 || at + (pq ? 128 : 64) > e) return fail("bad DQT");
                    for (int i = 0; i < 64; i++) { qt[tq][kZigzag[i]] = (uint16_t)(pq ? u16(at + 2 * i) : d[at + i]); }
                    at += pq ? 128 : 64;
                    qt_set[tq] = true;
                }
            } else if (m == 0xDD) {
                if (e - s < 2) return fail("bad DRI");
                restart = u16(s);
            } else if (m == 0xE0) {
                if (e - s >= 5 && !std::memcmp(d + s, "JFIF", 5)) jfif = true;
            } else if (m == 0xEE) {
                if (e - s >= 12 && !std::memcmp(d + s, "Adobe", 5)) { adobe = true; adobe_transform = d[s + 11]; }
            } else if (m == 0xDA) {
                if (!have_frame) return fail("scan before the frame header");
                if (plan_only) { sos_s = s; sos_e = e; return true; }
                if (!scan(s, e)) return false;
            }
        }
    }

    bool scan(size_t s, size_t e)
    {
        const int ns = d[s];
        if (ns < 1 || ns > ncomp || e - s < (size_t)(4 + 2 * ns)) return fail("bad SOS");
        Comp* sc[3];
        for (int i = 0; i < ns; i++) {
            const int id = d[s + 1 + 2 * i]; Comp* k = nullptr;
            for (int c = 0; c < ncomp; c++) if (comp[c].id == id) k = &comp[c];
            if (!k) return fail("scan names an unknown component");
            k->td = (d[s + 2 + 2 * i] >> 4) & 3; k->ta = d[s + 2 + 2 * i] & 3; k->pred = 0;
            sc[i] = k;
        }
        if (ns > 1) {                                     // jdinput.c per_scan_setup: at most D_MAX_BLOCKS_IN_MCU (10) blocks per MCU
            int blocks = 0;
            for (int i = 0; i < ns; i++) blocks += sc[i]->h * sc[i]->v;
            if (blocks > 10) return fail("sampling factors too large for an interleaved scan");
        }
        const int Ss = d[s + 1 + 2 * ns], Se = d[s + 2 + 2 * ns], Ah = d[s + 3 + 2 * ns] >> 4, Al = d[s + 3 + 2 * ns] & 15;
        if (progressive) {
            if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) return fail("bad progressive scan parameters");
        }
        for (int i = 0; i < ns; i++) {
            if ((!progressive || (Ss == 0 && Ah == 0)) && !dc[sc[i]->td].set) return fail("scan uses a Huffman table that was not defined");
            if ((!progressive || Ss > 0) && !ac[sc[i]->ta].set) return fail("scan uses a Huffman table that was not defined");
        }
        Bits br(d + e, d + n);
        int eobrun = 0, todo = restart;
        const bool inter = ns > 1;
        const int nx = inter ? mcux : (sc[0]->w + 7) / 8, ny = inter ? mcuy : (sc[0]->ht + 7) / 8;
        for (int my = 0; my < ny; my++)
            for (int mx = 0; mx < nx; mx++) {
                if (restart) {
                    if (todo == 0) {
                        br.restart();
                        for (int i = 0; i < ns; i++) sc[i]->pred = 0;
                        eobrun = 0; todo = restart;
                    }
                    todo--;
                }
                if (br.starved) continue;
                for (int i = 0; i < ns; i++) {
                    Comp& k = *sc[i];
                    const int bh = inter ? k.h : 1, bv = inter ? k.v : 1;
                    for (int v = 0; v < bv; v++)
                        for (int h = 0; h < bh; h++) {
                            int16_t* blk = &k.coef[((size_t)(my * bv + v) * k.bw + (mx * bh + h)) * 64];
                            if (!progressive) block_sequential(br, k, blk);
                            else if (Ss == 0) { if (Ah == 0) block_dc_first(br, k, blk, Al); else if (br.bit()) blk[0] |= (int16_t)(1 << Al); }
                            else if (Ah == 0) block_ac_first(br, k, blk, Ss, Se, Al, eobrun);
                            else block_ac_refine(br, k, blk, Ss, Se, Al, eobrun);
                        }
                }
            }
        // the segment ends at the next marker
        pos = (size_t)(br.p - d);
        if (br.marker && pos >= 2 && d[pos - 1] == br.marker && d[pos - 2] == 0xFF) pos -= 2;      // a marker met in the stream (not the end of the data)
        scans++;
        return true;
    }

    // jdhuff.c decode_mcu_slow
    // (the block decoders work on a copy of the reader: with every method inlined its fields live in registers)
    void block_sequential(Bits& io, Comp& k, int16_t* blk)
    {
        Bits br = io;
        int s = br.symbol(dc[k.td]);
        if (s) { const int r = br.get(s); s = extend(r, s); }
        k.pred += s;
        blk[0] = (int16_t)k.pred;
        const Huff& a = ac[k.ta];
        for (int i = 1; i < 64; i++) {
            const int f = a.fast[br.peek(16) >> (16 - kLookBits)];
            if (f) { i += (f >> 4) & 15; br.drop(f & 15); blk[kZigzag[i]] = (int16_t)(f >> 8); continue; }
            s = br.symbol(a);
            const int r = s >> 4; s &= 15;
            if (s) { i += r; const int x = br.get(s); blk[kZigzag[i]] = (int16_t)extend(x, s); }
            else { if (r != 15) break; i += 15; }
        }
        io = br;
    }
    // jdphuff.c decode_mcu_DC_first
    void block_dc_first(Bits& io, Comp& k, int16_t* blk, int Al)
    {
        Bits br = io;
        int s = br.symbol(dc[k.td]);
        if (s) { const int r = br.get(s); s = extend(r, s); }
        k.pred += s;
        blk[0] = (int16_t)((unsigned)k.pred << Al);
        io = br;
    }
    // jdphuff.c decode_mcu_AC_first
    void block_ac_first(Bits& io, Comp& k, int16_t* blk, int Ss, int Se, int Al, int& eobrun)
    {
        if (eobrun > 0) { eobrun--; return; }
        Bits br = io;
        const Huff& a = ac[k.ta];
        for (int i = Ss; i <= Se; i++) {
            const int f = a.fast[br.peek(16) >> (16 - kLookBits)];
            if (f) { i += (f >> 4) & 15; br.drop(f & 15); blk[kZigzag[i]] = (int16_t)((unsigned)(f >> 8) << Al); continue; }
            int s = br.symbol(a);
            const int r = s >> 4; s &= 15;
            if (s) { i += r; const int x = br.get(s); blk[kZigzag[i]] = (int16_t)((unsigned)extend(x, s) << Al); }
            else if (r == 15) i += 15;
            else { eobrun = 1 << r; if (r) eobrun += br.get(r); eobrun--; break; }
        }
        io = br;
    }
    // jdphuff.c decode_mcu_AC_refine
    void block_ac_refine(Bits& io, Comp& k, int16_t* blk, int Ss, int Se, int Al, int& eobrun)
    {
        Bits br = io;
        const int p1 = 1 << Al, m1 = -(1 << Al);
        const Huff& a = ac[k.ta];
        int i = Ss;
        auto refine = [&](int16_t* c) {
            if (br.bit() && !(*c & p1)) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
        };
        if (eobrun == 0) {
            for (; i <= Se; i++) {
                int s = br.symbol(a);
                int r = s >> 4; s &= 15;
                if (s) s = br.bit() ? p1 : m1;
                else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.get(r); break; }
                do {
                    int16_t* c = blk + kZigzag[i];
                    if (*c) refine(c);
                    else if (--r < 0) break;
                    i++;
                } while (i <= Se);
                if (s) blk[kZigzag[i]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; i <= Se; i++) { int16_t* c = blk + kZigzag[i]; if (*c) refine(c); }
            eobrun--;
        }
        io = br;
    }

    bool reconstruct()
    {
        for (int c = 0; c < ncomp; c++) {
            Comp& k = comp[c];
            if (!qt_set[k.tq]) return fail("frame uses a quantisation table that was not defined");
            const int stride = k.bw * 8;
            k.plane.resize((size_t)stride * k.bh * 8);
            for (int by = 0; by < k.bh; by++)
                for (int bx = 0; bx < k.bw; bx++)
                    idct_islow(&k.coef[((size_t)by * k.bw + bx) * 64], qt[k.tq], &k.plane[(size_t)by * 8 * stride + bx * 8], stride);
            std::vector<int16_t>().swap(k.own); k.coef = nullptr;
        }
        return true;
    }

    // jdsample.c: one output row of a component at full resolution (libjpeg's defaults: fancy upsampling for 2:1 ratios of
    // components wider than two samples, replication otherwise).  Rows beyond the real ones repeat the last real row, as the
    // context rows of jdmainct.c do.
    void upsample_row(const Comp& k, int y, uint8_t* out) const
    {
        const int he = hmax / k.h, ve = vmax / k.v, stride = k.bw * 8, w = k.w;
        auto row = [&](int r) { r = r < 0 ? 0 : r >= k.ht ? k.ht - 1 : r; return &k.plane[(size_t)r * stride]; };
        if (he == 1 && ve == 1) { std::memcpy(out, row(y), (size_t)w); return; }
        if (he == 2 && ve == 1) {
            const uint8_t* in = row(y);
            if (w > 2) {
                out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int x = 1; x < w - 1; x++) {
                    const int v = in[x] * 3;
                    out[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2); out[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
                }
                out[2 * w - 2] = (uint8_t)((in[w - 1] * 3 + in[w - 2] + 1) >> 2); out[2 * w - 1] = in[w - 1];
            } else for (int x = 0; x < w; x++) out[2 * x] = out[2 * x + 1] = in[x];
            return;
        }
        if (he == 1 && ve == 2) {                         // h1v2_fancy_upsample
            const int cy = y >> 1, lower = y & 1;
            const uint8_t* in0 = row(cy); const uint8_t* in1 = row(lower ? cy + 1 : cy - 1);
            const int bias = lower ? 2 : 1;
            for (int x = 0; x < w; x++) out[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            return;
        }
        if (he == 2 && ve == 2 && w > 2) {                // h2v2_fancy_upsample
            const int cy = y >> 1, lower = y & 1;
            const uint8_t* in0 = row(cy); const uint8_t* in1 = row(lower ? cy + 1 : cy - 1);
            int last, cur = in0[0] * 3 + in1[0], next = in0[1] * 3 + in1[1];
            out[0] = (uint8_t)((cur * 4 + 8) >> 4); out[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
            last = cur; cur = next;
            for (int x = 1; x < w - 1; x++) {
                next = in0[x + 1] * 3 + in1[x + 1];
                out[2 * x] = (uint8_t)((cur * 3 + last + 8) >> 4); out[2 * x + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                last = cur; cur = next;
            }
            out[2 * w - 2] = (uint8_t)((cur * 3 + last + 8) >> 4); out[2 * w - 1] = (uint8_t)((cur * 4 + 7) >> 4);
            return;
        }
        const uint8_t* in = row(y / ve);                  // h2v2_upsample / int_upsample: replication
        for (int x = 0; x < w; x++) for (int r = 0; r < he; r++) out[x * he + r] = in[x];
    }

    // jdapimin.c default_decompress_parms: what the three components are
    bool is_ycc() const
    {
        if (ncomp != 3) return false;
        if (jfif) return true;
        if (adobe) return adobe_transform != 0;
        return !(comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
    }

    bool output_bgr(uint8_t* out, size_t stride)
    {
        // jdcolor.c build_ycc_rgb_table, SCALEBITS 16
        struct Tables {
            int crr[256], cbb[256], crg[256], cbg[256];
            Tables() {
                for (int i = 0; i < 256; i++) {
                    const int x = i - 128;
                    crr[i] = (91881 * x + 32768) >> 16; cbb[i] = (116130 * x + 32768) >> 16;
                    crg[i] = -46802 * x; cbg[i] = -22554 * x + 32768;
                }
            }
        };
        static const Tables tb;                                   // built once, thread-safe (callers decode on several threads)
        const int* const crr = tb.crr; const int* const cbb = tb.cbb; const int* const crg = tb.crg; const int* const cbg = tb.cbg;
        const bool ycc = is_ycc();
        const size_t lw = (size_t)mcux * hmax * 8 + 8;
        std::vector<uint8_t> l0(lw), l1(lw), l2(lw);
        auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        for (int y = 0; y < H; y++) {
            uint8_t* o = out + (size_t)y * stride;
            upsample_row(comp[0], y, l0.data());
            if (ncomp == 1) {                              // gray_rgb_convert (cv::imread's default flag asks for colour)
                for (int x = 0; x < W; x++) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = l0[x];
                continue;
            }
            upsample_row(comp[1], y, l1.data()); upsample_row(comp[2], y, l2.data());
            if (ycc) {
                for (int x = 0; x < W; x++) {
                    const int Y = l0[x], cb = l1[x], cr = l2[x];
                    o[3 * x + 2] = clamp(Y + crr[cr]);
                    o[3 * x + 1] = clamp(Y + ((cbg[cb] + crg[cr]) >> 16));
                    o[3 * x] = clamp(Y + cbb[cb]);
                }
            } else for (int x = 0; x < W; x++) { o[3 * x + 2] = l0[x]; o[3 * x + 1] = l1[x]; o[3 * x] = l2[x]; }
        }
        return true;
    }
};

}  // namespace

bool jpeg_info(const uint8_t* data, size_t len, int* rows, int* cols, int* comps)
{
    Decoder dec(data, len);
    if (!data || !dec.read_headers(true)) { set_error(data ? dec.err : "jpeg: null buffer"); return false; }
    if (rows) *rows = dec.H;
    if (cols) *cols = dec.W;
    if (comps) *comps = dec.ncomp;
    return true;
}

bool jpeg_decode_bgr(const uint8_t* data, size_t len, uint8_t* bgr, int rows, int cols, size_t stride)
{
    if (!data || !bgr) { set_error("jpeg: null buffer"); return false; }
    try {
        Decoder dec(data, len);
        if (!dec.read_headers(true)) { set_error(dec.err); return false; }
        if (rows != dec.H || cols != dec.W || stride < (size_t)cols * 3) { set_error("jpeg: the output buffer does not have the image's size"); return false; }
        Decoder full(data, len);
        if (!full.read_headers(false) || !full.reconstruct()) { set_error(full.err); return false; }
        return full.output_bgr(bgr, stride);
    } catch (const std::bad_alloc&) {
        set_error("jpeg: out of memory");
        return false;
    }
}

static void describe(const Decoder& dec, JpegFrame& f)
{
    f.rows = dec.H; f.cols = dec.W; f.ncomp = dec.ncomp; f.hmax = dec.hmax; f.vmax = dec.vmax; f.mcux = dec.mcux; f.mcuy = dec.mcuy;
    f.ycc = dec.is_ycc(); f.coef_count = dec.coef_count();
    size_t off = 0;
    for (int c = 0; c < dec.ncomp; c++) {
        const Comp& k = dec.comp[c];
        JpegComponent& o = f.c[c];
        o.h = k.h; o.v = k.v; o.bw = k.bw; o.bh = k.bh; o.w = k.w; o.ht = k.ht; o.coef_off = off;
        std::memcpy(o.q, dec.qt[k.tq], sizeof(o.q));
        off += (size_t)k.bw * k.bh * 64;
    }
}

bool jpeg_frame_info(const uint8_t* data, size_t len, JpegFrame& f)
{
    Decoder dec(data, len);
    if (!data || !dec.read_headers(true)) { set_error(data ? dec.err : "jpeg: null buffer"); return false; }
    describe(dec, f);
    return true;
}

bool jpeg_entropy_decode(const uint8_t* data, size_t len, JpegFrame& f, int16_t* store, size_t store_cap)
{
    Decoder dec(data, len);
    dec.store = store; dec.store_cap = store_cap;
    if (!data || !store) { set_error("jpeg: null buffer"); return false; }
    if (!dec.read_headers(false)) { set_error(dec.err); return false; }
    for (int c = 0; c < dec.ncomp; c++)
        if (!dec.qt_set[dec.comp[c].tq]) { set_error("jpeg: frame uses a quantisation table that was not defined"); return false; }
    describe(dec, f);
    return true;
}

// What the parallel Huffman pass (jpeg_huff_par.hpp) needs of a stream it is able to take -- sequential, ONE scan holding every component in
// frame order, nothing but entropy-coded bytes (and, with a restart interval, RSTn markers in sequence) up to EOI: the frame, the scan's tables and block layout, and the scan's
// bytes with the stuffing removed (into `bits`, 16 zero bytes after them).  false (quietly: it is a question, not an error) otherwise.
bool jpeg_scan_plan(const uint8_t* data, size_t len, JpegFrame& f, HuffParPlan& P, uint8_t* bits, size_t cap, size_t* nbytes, std::vector<uint32_t>* seg_end)
{
    if (!data || !bits) return false;
    Decoder dec(data, len);
    dec.plan_only = true;
    if (!dec.read_headers(false) || !dec.sos_e || dec.progressive) return false;
    if (dec.restart && !seg_end) return false;
    if (seg_end) seg_end->clear();
    const uint8_t* d = data; const size_t s = dec.sos_s, e = dec.sos_e;
    const int ns = d[s];
    if (ns != dec.ncomp || e - s < (size_t)(4 + 2 * ns)) return false;
    std::memset(&P, 0, sizeof(P));
    int bpm = 0;
    for (int i = 0; i < ns; i++) {
        Comp& k = dec.comp[i];
        if (d[s + 1 + 2 * i] != k.id) return false;                                  // frame order
        const int td = (d[s + 2 + 2 * i] >> 4) & 3, ta = d[s + 2 + 2 * i] & 3;
        if (!dec.dc[td].set || !dec.ac[ta].set || !dec.qt_set[k.tq]) return false;
        const int bh = ns > 1 ? k.h : 1, bv = ns > 1 ? k.v : 1;
        for (int v = 0; v < bv; v++)
            for (int h = 0; h < bh; h++) {
                if (bpm >= 10) return false;
                P.comp_of[bpm] = i; P.hh[bpm] = h; P.vv[bpm] = v; P.dct[bpm] = td; P.act[bpm] = ta;
                P.used |= (1u << td) | (1u << (4 + ta));
                bpm++;
            }
        P.ch[i] = bh; P.cv[i] = bv; P.cbw[i] = k.bw; P.cblocks[i] = k.bw * k.bh;
    }
    for (int t = 0; t < 4; t++)
        for (int w = 0; w < 2; w++) {
            const Huff& h = w ? dec.ac[t] : dec.dc[t];
            if (!h.set) continue;
            HuffParTable& o = P.tab[4 * w + t];
            static_assert(kParLook == kLookBits, "one table shape for both decoders");
            std::memcpy(o.look, h.look, sizeof(o.look)); std::memcpy(o.maxcode, h.maxcode, sizeof(o.maxcode));
            std::memcpy(o.valoff, h.valoff, sizeof(o.valoff)); std::memcpy(o.vals, h.vals, sizeof(o.vals));
            for (int i = 0; i < (1 << kParLook); i++) {
                const int len = h.look[i] >> 8, sym = h.look[i] & 255;
                int bits = 0, adv = 0;
                if (len) {
                    if (!w) { bits = len + (sym & 15); adv = 1; }                     // DC: the difference's bits, then k = 1
                    else if (sym & 15) { bits = len + (sym & 15); adv = (sym >> 4) + 1; }
                    else if ((sym >> 4) == 15) { bits = len; adv = 16; }              // ZRL
                    else { bits = len; adv = 64; }                                    // EOB (jdhuff.c: any run below 15 with size 0 ends the block)
                }
                o.adv[i] = (uint16_t)((bits << 8) | adv);
            }
        }
    describe(dec, f);
    for (int i = 0; i < ns; i++) P.coef_off[i] = (uint32_t)f.c[i].coef_off;
    if (f.coef_count >= (1ull << 31)) return false;
    P.bpm = bpm; P.mcux = dec.mcux; P.mcuy = dec.mcuy; P.ncomp = ns;
    if (ns == 1) {
        // a lone component's scan: one block per "MCU", row-major over the blocks that hold real samples (the padding blocks the storage
        // may have to the right and below -- sampling factors above 1 -- are not in the stream)
        const Comp& k = dec.comp[0];
        P.mcux = (k.w + 7) / 8; P.mcuy = (k.ht + 7) / 8;
        P.ch[0] = P.cv[0] = 1; P.cblocks[0] = P.mcux * P.mcuy;
    }
    P.total_blocks = P.mcux * P.mcuy * bpm;
    // the scan's bytes without the stuffing, up to the marker that ends it -- which has to be EOI.  With a restart interval the RSTn markers
    // are taken out too and the bit position of every one (the end of a segment; the encoder padded to a byte there) is recorded.
    const uint8_t* p = d + e; const uint8_t* end = d + len;
    size_t o = 0;
    for (;;) {
        const uint8_t* ff = (const uint8_t*)std::memchr(p, 0xFF, (size_t)(end - p));
        if (!ff) return false;                                                        // no marker after the scan
        const size_t run = (size_t)(ff - p);
        if (o + run + 1 + 16 > cap) return false;
        std::memcpy(bits + o, p, run); o += run;
        p = ff + 1;
        while (p < end && *p == 0xFF) p++;                                            // fill bytes
        if (p >= end) return false;
        if (*p == 0) { bits[o++] = 0xFF; p++; continue; }
        if (dec.restart && *p >= 0xD0 && *p <= 0xD7) {
            if (*p != 0xD0 + (int)(seg_end->size() & 7) || o == 0 || o * 8 >= (1ull << 31)) return false;      // out of sequence: the serial pass's business
            if (!seg_end->empty() && seg_end->back() == (uint32_t)(o * 8)) return false;                          // an empty segment
            seg_end->push_back((uint32_t)(o * 8));
            p++;
            continue;
        }
        if (*p != 0xD9) return false;                                                 // DNL, another scan's tables, a stray RSTn: the serial pass's business
        break;
    }
    if (o == 0 || o * 8 >= (1ull << 31)) return false;
    std::memset(bits + o, 0, 16);                                                     // a symbol that starts on the last bits may be read to its end
    *nbytes = o;
    P.nbits = (uint32_t)(o * 8);
    P.nsub = (int32_t)((P.nbits + kSubBits - 1) / kSubBits);
    if (dec.restart) {
        const long mcus = ns > 1 ? (long)dec.mcux * dec.mcuy : (long)P.mcux * P.mcuy;
        const long want = (mcus + dec.restart - 1) / dec.restart;
        if (!seg_end->empty() && seg_end->back() == P.nbits) return false;           // a marker right before EOI
        seg_end->push_back(P.nbits);
        if ((long)seg_end->size() != want) return false;                             // not one segment per interval
        P.rst_blocks = (uint32_t)(dec.restart * bpm);
        P.nseg = (uint32_t)seg_end->size();
    }
    return true;
}

bool read_file_bytes(const char* filename, std::vector<uint8_t>& out)
{
    FILE* f = std::fopen(filename, "rb");
    if (!f) { set_error(std::string("cannot open ") + filename); return false; }
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    // (a directory opens, and ftell() then says LONG_MAX: nothing that is not a file of at most 2 GiB is an image cv::imread takes)
    const bool sane = sz > 0 && sz <= (1L << 31);
    out.resize(sane ? (size_t)sz : 0);
    const bool ok = sane && std::fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
    std::fclose(f);
    if (!ok) set_error(std::string("cannot read ") + filename);
    return ok;
}

}  // namespace pf

namespace pf {

static bool ppm_header(const std::vector<uint8_t>& b, int& w, int& h, size_t& at)
{
    if (b.size() < 2 || b[0] != 'P' || b[1] != '6') return false;
    at = 2;
    int v[3];
    for (int i = 0; i < 3; i++) {
        for (;;) {
            while (at < b.size() && (b[at] == ' ' || b[at] == '\t' || b[at] == '\r' || b[at] == '\n')) at++;
            if (at < b.size() && b[at] == '#') { while (at < b.size() && b[at] != '\n') at++; continue; }
            break;
        }
        if (at >= b.size() || b[at] < '0' || b[at] > '9') return false;
        v[i] = 0;
        while (at < b.size() && b[at] >= '0' && b[at] <= '9') {
            if (v[i] > (1 << 26)) return false;          // no header field of a picture cv::imread takes is that large (and the digit loop must not overflow)
            v[i] = v[i] * 10 + (b[at++] - '0');
        }
    }
    if (at >= b.size()) return false;                 // the header ends inside the file: ...
    at++;                                             // ... the single whitespace after maxval, then the pixels (at <= size: the caller subtracts)
    w = v[0]; h = v[1];
    return w > 0 && h > 0 && v[2] == 255 && (long long)w * h <= (1ll << 30);
}

// cv::imread(filename) for the formats the file driver feeds from: JPEG (by its SOI), PNG (by its signature) and binary PPM
static bool read_image_file_checked(const char* filename, std::vector<uint8_t>& bgr, int* rows, int* cols);
bool read_image_file(const char* filename, std::vector<uint8_t>& bgr, int* rows, int* cols)
{
    try { return read_image_file_checked(filename, bgr, rows, cols); }
    catch (const std::bad_alloc&) { set_error(std::string("out of memory reading ") + filename); return false; }
}
static bool read_image_file_checked(const char* filename, std::vector<uint8_t>& bgr, int* rows, int* cols)
{
    std::vector<uint8_t> b;
    if (!read_file_bytes(filename, b)) return false;
    int w = 0, h = 0;
    if (b.size() >= 2 && b[0] == 0xFF && b[1] == 0xD8) {
        if (!jpeg_info(b.data(), b.size(), &h, &w, nullptr)) return false;
        bgr.resize((size_t)w * h * 3);
        if (!jpeg_decode_bgr(b.data(), b.size(), bgr.data(), h, w, (size_t)w * 3)) return false;
    } else if (b.size() >= 8 && b[0] == 0x89 && b[1] == 'P' && b[2] == 'N' && b[3] == 'G') {
        if (!png_info(b.data(), b.size(), &h, &w)) return false;
        bgr.resize((size_t)w * h * 3);
        if (!png_decode_bgr(b.data(), b.size(), bgr.data(), h, w, (size_t)w * 3)) return false;
    } else {
        size_t at = 0;
        if (!ppm_header(b, w, h, at) || b.size() - at < (size_t)w * h * 3) { set_error(std::string("unsupported image file (JPEG, PNG and binary PPM are read): ") + filename); return false; }
        bgr.resize((size_t)w * h * 3);
        const uint8_t* s = b.data() + at;
        for (size_t i = 0; i < (size_t)w * h; i++) { bgr[3 * i] = s[3 * i + 2]; bgr[3 * i + 1] = s[3 * i + 1]; bgr[3 * i + 2] = s[3 * i]; }
    }
    *rows = h; *cols = w;
    return true;
}

}  // namespace pf
