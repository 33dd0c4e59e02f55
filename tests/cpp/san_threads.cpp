// san_threads.cpp -- the host code that runs on several threads, under ThreadSanitizer (no HIP):
//   include/pifusion/DataTrans.h   one producer / one consumer thread, the reference's tracker -> fusion wire (src/DataTrans.h:54-83;
//                                  the reference's own known race is the unguarded deque read at MultiBandMap2DCPU.cpp:606-617)
//   csrc/jpeg_decode.cpp           what pf_feed_jpeg_batch's host threads do per frame: jpeg_scan_plan (header walk, stuffing / RSTn removal),
//                                  the serial entropy pass, the full decode (function-static colour tables), pf::set_error per thread
//   csrc/image_io.cpp              pf_image_info / pf_read_image from dataset loader threads
//   csrc/dist_plan.hpp             plan_blend on every "rank" at once
// Driven by tests/test_sanitizers.py:  san_threads <corpus dir> <threads>.  Exit code 0 = no race reported, results equal across threads.
#include "jpeg_decode.hpp"
#include "jpeg_huff_par.hpp"
#include "dist_plan.hpp"
#include "../../include/pifusion.h"
#include <pifusion/DataTrans.h>
#include <dirent.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace pf {
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
const char* last_error() { return g_err.c_str(); }
}

static uint64_t fnv(const uint8_t* p, size_t n) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; } return h; }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string corpus = argv[1];
    const int T = std::atoi(argv[2]);
    std::vector<std::pair<std::string, std::vector<uint8_t>>> files;
    DIR* d = opendir(corpus.c_str());
    if (!d) return 2;
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        std::vector<uint8_t> b;
        if (pf::read_file_bytes((corpus + "/" + e->d_name).c_str(), b) && b.size() > 3 && ((b[0] == 0xFF && b[1] == 0xD8) || (b[0] == 0x89 && b[1] == 'P'))) files.push_back({ corpus + "/" + e->d_name, b });
    }
    closedir(d);
    int bad = 0;

    // 1. DataTrans: producer / consumer
    {
        DataTrans<int>& q = DataTrans<int>::Instance();
        const int N = 20000;
        std::atomic<long> got{ 0 }; std::atomic<int> order_bad{ 0 };
        std::thread cons([&] { int last = -1, v = 0; for (;;) { q.consumption(v); if (v < 0) break; if (v <= last) order_bad++; last = v; got++; } });
        std::thread prod([&] { for (int i = 0; i < N; i++) { q.product(i); if ((i & 255) == 0) std::this_thread::yield(); } });
        prod.join();
        while (q.size()) std::this_thread::yield();
        q.product(-1);
        cons.join();
        if (order_bad || got + (long)q.dropped() != N) { std::printf("DataTrans: %ld consumed + %zu dropped of %d, %d out of order\n", got.load(), q.dropped(), N, order_bad.load()); bad++; }
    }

    // 2. the decoders on T threads at once: every thread decodes every file, all must agree (first use of the function-static tables races here if they are unguarded)
    std::vector<std::vector<uint64_t>> sums(T, std::vector<uint64_t>(files.size(), 0));
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (size_t k = 0; k < files.size(); k++) {
                const size_t i = (k + (size_t)t) % files.size();
                const std::vector<uint8_t>& b = files[i].second;
                uint64_t h = 0;
                int r = 0, c = 0;
                if (b[0] == 0xFF) {
                    if (pf::jpeg_info(b.data(), b.size(), &r, &c, nullptr) && (long long)r * c <= (1 << 22)) {
                        std::vector<uint8_t> px((size_t)r * c * 3);
                        if (pf::jpeg_decode_bgr(b.data(), b.size(), px.data(), r, c, (size_t)c * 3)) h ^= fnv(px.data(), px.size());
                    }
                    pf::JpegFrame f; static thread_local pf::HuffParPlan P;
                    std::vector<uint8_t> bits(b.size() + 16); std::vector<uint32_t> seg; size_t nb = 0;
                    if (pf::jpeg_scan_plan(b.data(), b.size(), f, P, bits.data(), bits.size(), &nb, &seg)) h ^= fnv(bits.data(), nb) * 3;
                    pf::JpegFrame f2;
                    if (pf::jpeg_frame_info(b.data(), b.size(), f2) && f2.coef_count < ((size_t)1 << 24)) {
                        std::vector<int16_t> st(f2.coef_count);
                        pf::JpegFrame f3;
                        if (pf::jpeg_entropy_decode(b.data(), b.size(), f3, st.data(), st.size())) h ^= fnv((const uint8_t*)st.data(), st.size() * 2) * 5;
                    }
                } else if (pf::png_info(b.data(), b.size(), &r, &c) && (long long)r * c <= (1 << 22)) {
                    std::vector<uint8_t> px((size_t)r * c * 3);
                    if (pf::png_decode_bgr(b.data(), b.size(), px.data(), r, c, (size_t)c * 3)) h ^= fnv(px.data(), px.size());
                }
                if (pf_image_info(files[i].first.c_str(), &r, &c) && (long long)r * c <= (1 << 22)) {
                    std::vector<uint8_t> px((size_t)r * c * 3);
                    if (pf_read_image(files[i].first.c_str(), px.data(), r, c)) h ^= fnv(px.data(), px.size()) * 7;
                }
                pf::set_error("thread " + std::to_string(t));
                sums[t][i] = h;
            }
            if (std::string(pf::last_error()) != "thread " + std::to_string(t)) sums[t][0] = ~0ull;      // the message is per thread
        });
    for (auto& x : th) x.join();
    for (int t = 1; t < T; t++) if (sums[t] != sums[0]) { std::printf("thread %d decoded something else than thread 0\n", t); bad++; }

    // 3. plan_blend for every rank at once
    {
        const int n = 4;
        std::vector<std::vector<pf::TileRec>> all(n);
        for (int y = 0; y < 12; y++) for (int x = 0; x < 12; x++) all[(x * 7 + y * 3) % n].push_back({ x, y, (x + y) % 3 != 0 });
        std::vector<long long> caps(n, 1000);
        size_t hb[9]; for (int j = 0; j < 9; j++) hb[j] = j == 4 ? 0 : 100 + j;
        std::vector<pf::BlendPlan> plan(n);
        std::vector<std::thread> pt;
        for (int me = 0; me < n; me++) pt.emplace_back([&, me] { pf::plan_blend(all, caps, me, true, hb, plan[me]); });
        for (auto& x : pt) x.join();
        for (int a = 0; a < n; a++) for (int b = 0; b < n; b++) if (plan[a].send_bytes[b] != plan[b].recv_bytes[a]) { std::printf("plans disagree\n"); bad++; }
    }
    // 4. the PNG writer compresses bands of 256 rows on threads of its own: two writers at once, each with several bands
    {
        std::vector<uint8_t> px((size_t)900 * 70 * 3);
        for (size_t i = 0; i < px.size(); i++) px[i] = (uint8_t)(i * 2654435761u >> 13);
        const std::string dir = argc > 3 ? argv[3] : "/tmp";
        std::thread a([&] { if (!pf_write_image((dir + "/tsan_a.png").c_str(), px.data(), 900, 70)) bad++; });
        std::thread b([&] { if (!pf_write_image((dir + "/tsan_b.png").c_str(), px.data(), 700, 90)) bad++; });
        a.join(); b.join();
        std::vector<uint8_t> back((size_t)900 * 70 * 3);
        if (!pf_read_image((dir + "/tsan_a.png").c_str(), back.data(), 900, 70) || back != px) { std::printf("PNG written on threads reads back differently\n"); bad++; }
    }
    std::printf("files %zu threads %d violations %d\n", files.size(), T, bad);
    return bad ? 1 : 0;
}
