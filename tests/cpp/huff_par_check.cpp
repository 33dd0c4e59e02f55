// The parallel Huffman pass (csrc/jpeg_huff_par.hpp) run subsequence by subsequence in host loops -- the very functions the GPU runs one thread
// per subsequence -- against the serial pass of csrc/jpeg_decode.cpp: same coefficients, or the stream is one the plan refuses.
//   huff_par_check file.jpg ...   prints per file: "<file> eligible <0|1> rounds <n> equal <0|1>"
#include "jpeg_decode.hpp"
#include "jpeg_huff_par.hpp"
#include <cstdio>
#include <cstring>
#include <vector>

namespace pf { static thread_local std::string g; void set_error(const std::string& m) { g = m; } const char* last_error() { return g.c_str(); } }

int main(int argc, char** argv)
{
    int bad = 0;
    for (int a = 1; a < argc; a++) {
        std::vector<uint8_t> b;
        if (!pf::read_file_bytes(argv[a], b)) { std::printf("%s unreadable\n", argv[a]); bad++; continue; }
        pf::JpegFrame f; static pf::HuffParPlan P;
        std::vector<uint8_t> bits(b.size() + 16);
        std::vector<uint32_t> seg;
        size_t nbytes = 0;
        if (!pf::jpeg_scan_plan(b.data(), b.size(), f, P, bits.data(), bits.size(), &nbytes, &seg)) { std::printf("%s eligible 0\n", argv[a]); continue; }
        std::vector<uint32_t> words((nbytes + 16 + 3) / 4 + 2, 0);
        std::memcpy(words.data(), bits.data(), nbytes);
        const int S = P.nsub;
        const bool rst = P.rst_blocks != 0;
        std::vector<uint32_t> hint(S, 0);                      // first segment that ends after the subsequence's first bit
        if (rst) { uint32_t sg = 0; for (int i = 0; i < S; i++) { while (seg[sg] <= (uint32_t)i * pf::kSubBits) sg++; hint[i] = sg; } }
        auto sub = [&](int i, pf::HuffParState s0, pf::HuffParState& e, uint32_t& n) {
            if (rst) pf::huff_par_sub<true>(P, P.tab, words.data(), 0u, i, s0, e, n, seg.data(), hint[i]);
            else pf::huff_par_sub<false>(P, P.tab, words.data(), 0u, i, s0, e, n);
        };
        std::vector<pf::HuffParState> st[2] = { std::vector<pf::HuffParState>(S), std::vector<pf::HuffParState>(S) };
        std::vector<uint32_t> nblk(S);
        for (int i = 0; i < S; i++) { pf::HuffParState s0 = { (uint32_t)i * pf::kSubBits, 0 }; sub(i, s0, st[0][i], nblk[i]); }
        int rounds = 0, cur = 0;
        for (;;) {
            bool changed = false;
            st[cur ^ 1][0] = st[cur][0];
            for (int i = 1; i < S; i++) {
                pf::HuffParState e; uint32_t n;
                sub(i, st[cur][i - 1], e, n);
                if (e.p != st[cur][i].p || e.ck != st[cur][i].ck || n != nblk[i]) changed = true;
                st[cur ^ 1][i] = e; nblk[i] = n;
            }
            cur ^= 1; rounds++;
            if (!changed) break;
        }
        std::vector<int16_t> coef(f.coef_count, 0), ref(f.coef_count, 0);
        uint32_t g = 0, bad_flag = 0; bool ok = true; pf::HuffParState last = { 0, 0 }; uint32_t g_last = 0;
        for (int i = 0; i < S; i++) {
            pf::HuffParState s0 = i ? st[cur][i - 1] : pf::HuffParState{ 0, 0 };
            pf::HuffParState e; uint32_t ge;
            if (rst) ok = pf::huff_par_write<true>(P, P.tab, words.data(), 0u, i, s0, g, coef.data(), e, ge, seg.data(), hint[i], &bad_flag) && ok;
            else ok = pf::huff_par_write<false>(P, P.tab, words.data(), 0u, i, s0, g, coef.data(), e, ge) && ok;
            g += nblk[i]; last = e; g_last = ge;
        }
        ok = ok && !bad_flag;
        const bool ends = g_last == (uint32_t)P.total_blocks && last.ck == 0 && P.nbits - last.p < 8;
        for (int c = 0; c < P.ncomp; c++) {
            int pred = 0;
            const uint32_t group = rst ? P.rst_blocks / (uint32_t)P.bpm * (uint32_t)(P.ch[c] * P.cv[c]) : 0u;          // a component's blocks per restart interval
            for (uint32_t t = 0; t < (uint32_t)P.cblocks[c]; t++) {
                if (group && t % group == 0) pred = 0;
                const uint32_t at = pf::huff_par_comp_block(P, c, t); pred += coef[at]; coef[at] = (int16_t)pred;
            }
        }
        pf::JpegFrame f2;
        const bool refok = pf::jpeg_entropy_decode(b.data(), b.size(), f2, ref.data(), ref.size());
        const bool equal = refok && coef == ref;
        std::printf("%s eligible 1 subsequences %d rounds %d ends_on_last_block %d equal %d\n", argv[a], S, rounds, (int)ends, (int)equal);
        if (!equal || !ends || !ok) bad++;
    }
    return bad ? 1 : 0;
}
