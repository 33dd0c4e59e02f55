// jpeg_huff_par.hpp -- the Huffman pass of a sequential JPEG scan, decoded in parallel (jpeg_device.hip runs it one thread per
// subsequence; tests/cpp/huff_par_check.cpp runs the same functions in loops on the host against jpeg_decode.cpp's serial pass).
//
// A Huffman stream can only be read from its beginning -- but a decoder dropped into the middle of one falls into step with the true
// symbol boundaries after a few dozen symbols (the codes are self-synchronising), and a JPEG decoder's whole state at a bit position is
// small: which block of the MCU it is in and how far into that block (c, k).  So (Weissenberger & Schmidt, "Accelerating JPEG
// decompression on GPUs", 2021; restated from the published description): cut the scan's bits into subsequences of kSubBits;
//   round 0   every subsequence i is decoded from its first bit as if a block began there; its end state (p, c, k) and the number of
//             blocks it completed are recorded;
//   round r   every subsequence is decoded again from the end state its predecessor recorded in round r - 1.  Subsequence 0 is right from
//             the start, so after round r the first r + 1 are right for certain -- and in practice all of them after two or three, because
//             a wrong start state has synchronised long before the subsequence ends.  Rounds repeat until nothing changes;
//   write     an exclusive sum of the block counts gives every subsequence the index of the block it starts in; one more decode from the
//             now exact start states writes the coefficients (DC values still as differences) into the dense array the IDCT kernel reads;
//   DC        a running sum per component in scan order turns the differences into values (jdhuff.c: last_dc_val).
// The stream handed over has its byte stuffing removed (jpeg_scan_plan); the decode of a symbol is jdhuff.c's, statement for
// statement the one of jpeg_decode.cpp's block_sequential.  Restart intervals: the RSTn markers are removed too, the ends of their segments
// recorded, no symbol is read across one, the DC running sums start over per interval.  Streams this does not take -- progressive, several
// scans, a marker other than RSTn inside the scan -- and streams whose write pass does not end exactly on the last block keep the host's serial pass.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PF_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PF_HD inline
#endif

namespace pf {

constexpr int kSubBits = 512;                 // bits per subsequence
constexpr int kParLook = 10;                  // bits of the first-level code table (jpeg_decode.cpp's kLookBits)

struct HuffParTable {
    uint16_t look[1 << kParLook];             // prefix -> (length << 8) | symbol, 0: longer code
    int32_t  maxcode[18];
    int32_t  valoff[17];
    uint8_t  vals[256];
    int32_t  pad_;
    // what the rounds need of a symbol and nothing else: prefix -> (bits of the code AND of its value << 8) | advance of k (a coefficient
    // after a run r: r + 1; ZRL: 16; EOB: 64 = past the block's end; a DC symbol: 1), 0: the code is longer than the prefix
    uint16_t adv[1 << kParLook];
};
struct HuffParPlan {
    uint32_t nbits;                           // entropy-coded bits (stuffing removed)
    int32_t  nsub;                            // subsequences: ceil(nbits / kSubBits)
    int32_t  bpm, total_blocks, mcux, mcuy, ncomp;
    uint32_t used;                            // bit t: table t (0..3 DC, 4..7 AC) is one the scan uses
    // restart intervals (DRI): the stream handed over has its RSTn markers removed; segment s ends at bit seg_end[s] (a byte boundary, the
    // last one = nbits) and holds rst_blocks blocks (the last segment what is left); 0: no restart interval
    uint32_t rst_blocks, nseg;
    int32_t  comp_of[10], hh[10], vv[10], dct[10], act[10];      // per block of an MCU: component, position inside the MCU, tables
    int32_t  ch[3], cv[3], cbw[3], cblocks[3];                   // per component: sampling factors, blocks per row held, blocks in all
    uint32_t coef_off[3];                                        // first coefficient of the component in the dense array (int16 units)
    HuffParTable tab[8];                      // 0..3 DC, 4..7 AC
};

struct HuffParState { uint32_t p; uint32_t ck; };                // bit position; c | k << 8

// what a symbol's decode needs of the plan, in registers: the table of every block of the MCU, four bits each
struct HuffParCtl { uint64_t dc_pack, ac_pack; int bpm; };
PF_HD HuffParCtl huff_par_ctl(const HuffParPlan& P)
{
    HuffParCtl c = { 0, 0, P.bpm };
    for (int i = 0; i < 10; i++) { c.dc_pack |= (uint64_t)(P.dct[i] & 15) << (4 * i); c.ac_pack |= (uint64_t)((4 + P.act[i]) & 15) << (4 * i); }
    return c;
}

PF_HD uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

// 16 bits of the stream at bit position p (words: the stream as aligned little-endian loads of its bytes; 16 zero bytes follow the data)
PF_HD uint32_t peek16(const uint32_t* __restrict__ words, uint32_t p)
{
    // (p is relative to words[0]: the callers subtract the first bit of the window they staged)
    const uint32_t w = p >> 5;
    const uint64_t v = ((uint64_t)bswap32(words[w]) << 32) | bswap32(words[w + 1]);
    return (uint32_t)((v << (p & 31)) >> 48);
}

PF_HD int zigzag_of(int k)
{
    // natural index of the k-th coefficient in zigzag order; a run past the end lands on 63 (libjpeg's padded table)
    const uint8_t z[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
    return k < 64 ? z[k] : 63;
}

// One symbol (jdhuff.c decode_mcu_slow; jpeg_decode.cpp block_sequential).  In: the state.  Out: the state after the symbol, `wpos` = natural
// index the symbol's value belongs at in the current block (-1: none), `done`: the block ended with this symbol.
PF_HD void huff_par_step(const HuffParCtl& P, const HuffParTable* __restrict__ tabs, const uint32_t* __restrict__ words, uint32_t bit0,
                         uint32_t& p, int& c, int& k, int& wpos, int& value, bool& done)
{
    const HuffParTable& t = tabs[(int)(((k == 0 ? P.dc_pack : P.ac_pack) >> (4 * c)) & 15)];
    const uint32_t pre = peek16(words, p - bit0);
    const uint32_t e = t.look[pre >> (16 - kParLook)];
    int len, sym;
    if (e) { len = (int)(e >> 8); sym = (int)(e & 255); }
    else {
        int l = kParLook + 1;
        while (l <= 16 && (int)(pre >> (16 - l)) > t.maxcode[l]) l++;
        if (l > 16) { len = 16; sym = 0; }
        else { len = l; sym = t.vals[((int)(pre >> (16 - l)) + t.valoff[l]) & 255]; }
    }
    p += (uint32_t)len;
    wpos = -1; value = 0; done = false;
    if (k == 0) {                                    // DC difference
        const int s = sym & 15;
        if (s) { const int x = (int)(peek16(words, p - bit0) >> (16 - s)); p += (uint32_t)s; value = x < (1 << (s - 1)) ? x - (1 << s) + 1 : x; }
        wpos = 0; k = 1;
    } else {
        const int r = sym >> 4, s = sym & 15;
        if (s) {
            k += r;
            const int x = (int)(peek16(words, p - bit0) >> (16 - s)); p += (uint32_t)s;
            value = x < (1 << (s - 1)) ? x - (1 << s) + 1 : x;
            wpos = zigzag_of(k);
            k++;
        } else if (r == 15) k += 16;
        else done = true;                            // EOB
    }
    if (k >= 64) done = true;
    if (done) { k = 0; c = c + 1 == P.bpm ? 0 : c + 1; }
}

// Restart intervals: a decoder never reads across the end B of its segment -- the encoder pads the last byte of a segment with one-bits, which
// are the prefix of no code, so the true decode meets at most seven bits there that only make a symbol together with the next segment's; a
// symbol that would end beyond B is therefore not a symbol: the decoder steps to B and starts over (block 0 of an MCU, DC first), as
// jdhuff.c's process_restart does.  `s` = index of the segment p lies in.
PF_HD void huff_par_seg_find(const uint32_t* __restrict__ seg_end, uint32_t hint, uint32_t p, uint32_t& s, uint32_t& B)
{
    s = hint;
    B = seg_end[s];
    while (B <= p) { s++; B = seg_end[s]; }              // ends at nbits > p: terminates
}

// Subsequence i decoded from `start` to its end: the state there and the blocks completed on the way
template <bool RST>
PF_HD void huff_par_sub(const HuffParPlan& P, const HuffParTable* __restrict__ tabs, const uint32_t* __restrict__ words, uint32_t bit0, int i,
                        HuffParState start, HuffParState& end, uint32_t& nblk, const uint32_t* __restrict__ seg_end = nullptr, uint32_t seg_hint = 0)
{
    const uint64_t lim64 = (uint64_t)(i + 1) * kSubBits;
    const uint32_t limit = lim64 < P.nbits ? (uint32_t)lim64 : P.nbits;
    uint32_t p = start.p; int c = (int)(start.ck & 255), k = (int)(start.ck >> 8);
    uint32_t n = 0;
    const HuffParCtl ctl = huff_par_ctl(P);
    uint32_t sg = 0, B = 0xffffffffu;
    if (RST && p < limit) huff_par_seg_find(seg_end, seg_hint, p, sg, B);
    while (p < limit) {
        const uint32_t p0 = p; const int c0 = c, k0 = k; const uint32_t n0 = n;
        // the short way: one look-up says how far the symbol moves p and k (the values themselves are the write pass's business)
        const uint32_t e = tabs[(int)(((k == 0 ? ctl.dc_pack : ctl.ac_pack) >> (4 * c)) & 15)].adv[peek16(words, p - bit0) >> (16 - kParLook)];
        if (e) {
            p += e >> 8; k += (int)(e & 255);
            if (k >= 64) { n++; k = 0; c = c + 1 == ctl.bpm ? 0 : c + 1; }
        } else {
            int wpos, value; bool done;
            huff_par_step(ctl, tabs, words, bit0, p, c, k, wpos, value, done);
            n += done ? 1u : 0u;
        }
        if (RST && p >= B) {
            if (p > B) { n = n0; (void)p0; (void)c0; (void)k0; }          // not a symbol: padding
            p = B; c = 0; k = 0;
            if (p < P.nbits) { sg++; B = seg_end[sg]; }
        }
    }
    end.p = p; end.ck = (uint32_t)c | ((uint32_t)k << 8);
    nblk = n;
}

// first coefficient (int16 index into the dense array) of block c of MCU `mcu`
PF_HD uint32_t huff_par_block_base(const HuffParPlan& P, uint32_t mcu, int c)
{
    const int comp = P.comp_of[c];
    const uint32_t my = mcu / (uint32_t)P.mcux, mx = mcu - my * (uint32_t)P.mcux;
    const uint32_t bx = mx * (uint32_t)P.ch[comp] + (uint32_t)P.hh[c], by = my * (uint32_t)P.cv[comp] + (uint32_t)P.vv[c];
    return P.coef_off[comp] + (by * (uint32_t)P.cbw[comp] + bx) * 64u;
}

// The write pass of subsequence i: from the exact start state, in the block with index `g` (blocks completed before the subsequence).
// `bad` is raised when a segment does not end on its last block (restart intervals): the stream is not one this decoder takes.
template <bool RST>
PF_HD bool huff_par_write(const HuffParPlan& P, const HuffParTable* __restrict__ tabs, const uint32_t* __restrict__ words, uint32_t bit0, int i,
                          HuffParState start, uint32_t g, int16_t* __restrict__ coef, HuffParState& end, uint32_t& g_end,
                          const uint32_t* __restrict__ seg_end = nullptr, uint32_t seg_hint = 0, uint32_t* __restrict__ bad = nullptr)
{
    const uint64_t lim64 = (uint64_t)(i + 1) * kSubBits;
    const uint32_t limit = lim64 < P.nbits ? (uint32_t)lim64 : P.nbits;
    uint32_t p = start.p; int c = (int)(start.ck & 255), k = (int)(start.ck >> 8);
    const uint32_t total = (uint32_t)P.total_blocks;
    uint32_t base = g < total ? huff_par_block_base(P, g / (uint32_t)P.bpm, c) : 0u;
    const HuffParCtl ctl = huff_par_ctl(P);
    uint32_t sg = 0, B = 0xffffffffu;
    if (RST && p < limit) huff_par_seg_find(seg_end, seg_hint, p, sg, B);
    while (p < limit && g < total) {
        int wpos, value; bool done;
        const int c0 = c, k0 = k;
        huff_par_step(ctl, tabs, words, bit0, p, c, k, wpos, value, done);
        if (RST && p > B) {                                   // padding at the end of the segment: nothing to write, the next segment starts over
            p = B; c = 0; k = 0; done = false; wpos = -1;
            // The symbol that crossed the end must have started on a block boundary with every block of the interval done.  A damaged
            // segment that finishes its blocks a byte early would otherwise have its leftover bits decoded as the DC and a few AC symbols
            // of a partial block and WRITTEN into the next interval's first block (the serial pass, like libjpeg, discards them at the
            // restart): such a stream goes to the serial pass.
            if ((g != (sg + 1) * P.rst_blocks || c0 != 0 || k0 != 0) && bad) *bad = 1u;
            if (g < total) base = huff_par_block_base(P, g / (uint32_t)P.bpm, 0);
            if (p < P.nbits) { sg++; B = seg_end[sg]; }
            continue;
        }
        if (wpos >= 0 && value != 0) coef[base + (uint32_t)wpos] = (int16_t)value;
        if (done) {
            g++;
            if (g < total) base = huff_par_block_base(P, g / (uint32_t)P.bpm, c);
        }
        if (RST && p == B) {                                  // the segment ended on a byte boundary without padding
            if ((c != 0 || k != 0 || g != (sg + 1) * P.rst_blocks) && g < total && bad) *bad = 1u;
            c = 0; k = 0;
            if (g < total) base = huff_par_block_base(P, g / (uint32_t)P.bpm, 0);
            if (p < P.nbits) { sg++; B = seg_end[sg]; }
        }
    }
    end.p = p; end.ck = (uint32_t)c | ((uint32_t)k << 8);
    g_end = g;
    return true;
}

// scan-order index t of a component's block -> its first coefficient (for the DC running sum)
PF_HD uint32_t huff_par_comp_block(const HuffParPlan& P, int comp, uint32_t t)
{
    const uint32_t per = (uint32_t)(P.ch[comp] * P.cv[comp]);
    const uint32_t mcu = t / per, r = t - mcu * per;
    const uint32_t v = r / (uint32_t)P.ch[comp], h = r - v * (uint32_t)P.ch[comp];
    const uint32_t my = mcu / (uint32_t)P.mcux, mx = mcu - my * (uint32_t)P.mcux;
    return P.coef_off[comp] + ((my * (uint32_t)P.cv[comp] + v) * (uint32_t)P.cbw[comp] + mx * (uint32_t)P.ch[comp] + h) * 64u;
}

}  // namespace pf
