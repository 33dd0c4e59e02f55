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
        size_t nbytes = 0;
        if (!pf::jpeg_scan_plan(b.data(), b.size(), f, P, bits.data(), bits.size(), &nbytes)) { std::printf("%s eligible 0\n", argv[a]); continue; }
        std::vector<uint32_t> words((nbytes + 16 + 3) / 4 + 2, 0);
        std::memcpy(words.data(), bits.data(), nbytes);
        const int S = P.nsub;
        std::vector<pf::HuffParState> st[2] = { std::vector<pf::HuffParState>(S), std::vector<pf::HuffParState>(S) };
        std::vector<uint32_t> nblk(S);
        // round 0 with every block of the MCU as the assumed one (the GPU: one thread per subsequence and phase), then the phases linked
        // from subsequence to subsequence: c at the end of i - 1 picks the candidate of i
        const int bpm = P.bpm;
        std::vector<pf::HuffParState> cand((size_t)S * bpm); std::vector<uint32_t> cn((size_t)S * bpm);
        for (int i = 0; i < S; i++)
            for (int c0 = 0; c0 < bpm; c0++) {
                pf::HuffParState s0 = { (uint32_t)i * pf::kSubBits, (uint32_t)c0 };
                pf::huff_par_sub(P, P.tab, words.data(), 0u, i, s0, cand[(size_t)i * bpm + c0], cn[(size_t)i * bpm + c0]);
            }
        {
            int c = 0;
            for (int i = 0; i < S; i++) { st[0][i] = cand[(size_t)i * bpm + c]; nblk[i] = cn[(size_t)i * bpm + c]; c = (int)(st[0][i].ck & 255); }
        }
        int rounds = 0, cur = 0;
        for (;;) {
            bool changed = false;
            st[cur ^ 1][0] = st[cur][0];
            for (int i = 1; i < S; i++) {
                pf::HuffParState e; uint32_t n;
                pf::huff_par_sub(P, P.tab, words.data(), 0u, i, st[cur][i - 1], e, n);
                if (e.p != st[cur][i].p || e.ck != st[cur][i].ck || n != nblk[i]) changed = true;
                st[cur ^ 1][i] = e; nblk[i] = n;
            }
            cur ^= 1; rounds++;
            if (!changed) break;
        }
        std::vector<int16_t> coef(f.coef_count, 0), ref(f.coef_count, 0);
        uint32_t g = 0; bool ok = true; pf::HuffParState last = { 0, 0 }; uint32_t g_last = 0;
        for (int i = 0; i < S; i++) {
            pf::HuffParState s0 = i ? st[cur][i - 1] : pf::HuffParState{ 0, 0 };
            pf::HuffParState e; uint32_t ge;
            ok = pf::huff_par_write(P, P.tab, words.data(), 0u, i, s0, g, coef.data(), e, ge) && ok;
            g += nblk[i]; last = e; g_last = ge;
        }
        const bool ends = g_last == (uint32_t)P.total_blocks && last.ck == 0 && P.nbits - last.p < 8;
        for (int c = 0; c < P.ncomp; c++) {
            int pred = 0;
            for (uint32_t t = 0; t < (uint32_t)P.cblocks[c]; t++) { const uint32_t at = pf::huff_par_comp_block(P, c, t); pred += coef[at]; coef[at] = (int16_t)pred; }
        }
        pf::JpegFrame f2;
        const bool refok = pf::jpeg_entropy_decode(b.data(), b.size(), f2, ref.data(), ref.size());
        const bool equal = refok && coef == ref;
        std::printf("%s eligible 1 subsequences %d rounds %d ends_on_last_block %d equal %d\n", argv[a], S, rounds, (int)ends, (int)equal);
        if (!equal || !ends || !ok) bad++;
    }
    return bad ? 1 : 0;
}
