// san_fuzz.cpp -- the host code that takes untrusted bytes, under AddressSanitizer + UndefinedBehaviorSanitizer (no HIP):
//   csrc/jpeg_decode.cpp   jpeg_info / jpeg_decode_bgr / jpeg_entropy_decode / jpeg_scan_plan (header walk, stuffing and RSTn stripping)
//   csrc/jpeg_huff_par.hpp the GPU's parallel Huffman pass run subsequence by subsequence in host loops, against the serial pass:
//                          a stream the parallel pass ACCEPTS must give the serial pass's coefficients
//   csrc/png_decode.cpp    png_info / png_decode_bgr
//   csrc/image_io.cpp      write_image_file + the pf_image_info / pf_read_image / pf_write_image entry points
//   include/pifusion/TestSystem.h   Config, read_ppm_bgr, DroneMapDataset (config.cfg / trajectory.txt / rgb/<name>.{jpg,png,ppm})
//   include/pifusion/DataTrans.h    cap-30 drop-oldest queue
//   csrc/dist_plan.hpp     plan_blend: what rank a plans to send to b is what b plans to receive from a
// Driven by tests/test_sanitizers.py:  san_fuzz <corpus dir> <work dir> <seed> <mutants per file>
// The corpus holds the golden streams (tests/golden/jpeg_vectors.npz), encoder-made restart-interval streams, PNG / PPM files and dataset
// texts; every file is run as it is and `mutants` times damaged (bit flips, byte runs, truncation, insertion, slices repeated or zeroed,
// entropy-segment flips for JPEG).  Exit code 0 = no sanitizer report and no contract violation; counts on stdout.
// Reference for the contracts: cv::imread's behaviour at backup/map2dfusion.cpp:129-135 (a file it cannot read yields an empty frame, not a crash).
#include "jpeg_decode.hpp"
#include "jpeg_huff_par.hpp"
#include "dist_plan.hpp"
#include <pifusion/TestSystem.h>
#include <pifusion/DataTrans.h>
#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace pf {
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
const char* last_error() { return g_err.c_str(); }
}

static uint64_t g_state = 1;
static uint64_t rnd() { uint64_t z = (g_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static size_t rnd_below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

struct Counts { long jpeg_ok = 0, jpeg_refused = 0, par_eligible = 0, par_accepted = 0, par_to_serial = 0, par_unequal = 0, par_accept_serial_refuse = 0,
                png_ok = 0, png_refused = 0, ppm_ok = 0, ppm_refused = 0, cfg = 0, frames = 0, plans = 0, files = 0, mutants = 0; };
static Counts C;
static int g_fail = 0;
static void fail(const char* what, const std::string& name) { std::printf("VIOLATION %s in %s\n", what, name.c_str()); g_fail++; }

// the parallel pass of jpeg_huff_par.hpp in host loops (tests/cpp/huff_par_check.cpp, with the GPU's limits: at most 64 * 8 sweeps)
static void par_vs_serial(const std::vector<uint8_t>& b, const std::string& name)
{
    pf::JpegFrame f; static pf::HuffParPlan P;
    std::vector<uint8_t> bits(b.size() + 16);
    std::vector<uint32_t> seg;
    size_t nbytes = 0;
    if (!pf::jpeg_scan_plan(b.data(), b.size(), f, P, bits.data(), bits.size(), &nbytes, &seg)) return;
    C.par_eligible++;
    if (f.coef_count > ((size_t)1 << 24)) return;
    std::vector<uint32_t> words((nbytes + 16 + 3) / 4 + 2, 0);
    std::memcpy(words.data(), bits.data(), nbytes);
    const int S = P.nsub;
    const bool rst = P.rst_blocks != 0;
    std::vector<uint32_t> hint(S, 0);
    if (rst) {
        seg.push_back(P.nbits);                                                       // the guard the device path puts behind the last end
        uint32_t sg = 0;
        for (int i = 0; i < S; i++) { while (sg + 1 < seg.size() && seg[sg] <= (uint32_t)i * pf::kSubBits) sg++; hint[i] = sg; }
    }
    auto sub = [&](int i, pf::HuffParState s0, pf::HuffParState& e, uint32_t& n) {
        if (rst) pf::huff_par_sub<true>(P, P.tab, words.data(), 0u, i, s0, e, n, seg.data(), hint[i]);
        else pf::huff_par_sub<false>(P, P.tab, words.data(), 0u, i, s0, e, n);
    };
    std::vector<pf::HuffParState> st[2] = { std::vector<pf::HuffParState>(S), std::vector<pf::HuffParState>(S) };
    std::vector<uint32_t> nblk(S);
    for (int i = 0; i < S; i++) { pf::HuffParState s0 = { (uint32_t)i * pf::kSubBits, 0 }; sub(i, s0, st[0][i], nblk[i]); }
    int rounds = 0, cur = 0; bool settled = false;
    while (rounds < 512) {
        bool changed = false;
        st[cur ^ 1][0] = st[cur][0];
        for (int i = 1; i < S; i++) {
            pf::HuffParState e; uint32_t n;
            sub(i, st[cur][i - 1], e, n);
            if (e.p != st[cur][i].p || e.ck != st[cur][i].ck || n != nblk[i]) changed = true;
            st[cur ^ 1][i] = e; nblk[i] = n;
        }
        cur ^= 1; rounds++;
        if (!changed) { settled = true; break; }
    }
    if (!settled) { C.par_to_serial++; return; }
    std::vector<int16_t> coef(f.coef_count, 0), ref(f.coef_count, 0);
    uint32_t g = 0, bad_flag = 0; bool ok = true; pf::HuffParState last = { 0, 0 }; uint32_t g_last = 0;
    for (int i = 0; i < S; i++) {
        pf::HuffParState s0 = i ? st[cur][i - 1] : pf::HuffParState{ 0, 0 };
        pf::HuffParState e; uint32_t ge;
        if (rst) ok = pf::huff_par_write<true>(P, P.tab, words.data(), 0u, i, s0, g, coef.data(), e, ge, seg.data(), hint[i], &bad_flag) && ok;
        else ok = pf::huff_par_write<false>(P, P.tab, words.data(), 0u, i, s0, g, coef.data(), e, ge) && ok;
        g += nblk[i]; last = e; g_last = ge;
    }
    const bool ends = g_last == (uint32_t)P.total_blocks && last.ck == 0 && P.nbits - last.p < 8;
    if (!(ok && !bad_flag && ends)) { C.par_to_serial++; return; }                     // the device path hands such a stream to the serial pass
    C.par_accepted++;
    for (int c = 0; c < P.ncomp; c++) {
        int pred = 0;
        const uint32_t group = rst ? P.rst_blocks / (uint32_t)P.bpm * (uint32_t)(P.ch[c] * P.cv[c]) : 0u;
        for (uint32_t t = 0; t < (uint32_t)P.cblocks[c]; t++) {
            if (group && t % group == 0) pred = 0;
            const uint32_t at = pf::huff_par_comp_block(P, c, t); pred += coef[at]; coef[at] = (int16_t)pred;
        }
    }
    pf::JpegFrame f2;
    if (!pf::jpeg_entropy_decode(b.data(), b.size(), f2, ref.data(), ref.size())) { C.par_accept_serial_refuse++; return; }
    if (coef != ref) { C.par_unequal++; fail("parallel Huffman pass accepted a stream and differs from the serial pass", name); }
}

static void check_jpeg(const std::vector<uint8_t>& b, const std::string& name)
{
    int r = 0, c = 0, n = 0;
    if (pf::jpeg_info(b.data(), b.size(), &r, &c, &n) && (long long)r * c <= (1 << 22)) {
        const size_t stride = (size_t)c * 3 + 5;                                       // an odd row step: rows must not run into each other
        std::vector<uint8_t> out((size_t)r * stride);
        if (pf::jpeg_decode_bgr(b.data(), b.size(), out.data(), r, c, stride)) C.jpeg_ok++; else C.jpeg_refused++;
    } else C.jpeg_refused++;
    pf::JpegFrame f;
    if (pf::jpeg_frame_info(b.data(), b.size(), f) && f.coef_count <= ((size_t)1 << 24)) {
        std::vector<int16_t> store(f.coef_count);
        pf::JpegFrame f2;
        (void)pf::jpeg_entropy_decode(b.data(), b.size(), f2, store.data(), store.size());
    }
    par_vs_serial(b, name);
}

static void check_png(const std::vector<uint8_t>& b)
{
    int r = 0, c = 0;
    if (pf::png_info(b.data(), b.size(), &r, &c) && (long long)r * c <= (1 << 22)) {
        const size_t stride = (size_t)c * 3 + 1;
        std::vector<uint8_t> out((size_t)r * stride);
        if (pf::png_decode_bgr(b.data(), b.size(), out.data(), r, c, stride)) C.png_ok++; else C.png_refused++;
    } else C.png_refused++;
}

static bool write_file(const std::string& p, const uint8_t* d, size_t n)
{
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) return false;
    const bool ok = n == 0 || std::fwrite(d, 1, n, f) == n;
    std::fclose(f);
    return ok;
}

// through the file entry points the dataset reader uses (cv::imread's stand-ins) and the reader itself
static void check_files(const std::vector<uint8_t>& b, const std::string& work, const char* ext)
{
    const std::string p = work + "/probe" + ext;
    if (!write_file(p, b.data(), b.size())) return;
    int r = 0, c = 0;
    if (pf_image_info(p.c_str(), &r, &c) && (long long)r * c <= (1 << 22)) {
        std::vector<uint8_t> px((size_t)r * c * 3);
        (void)pf_read_image(p.c_str(), px.data(), r, c);
        (void)pf_read_image(p.c_str(), px.data(), r + 1, c);                           // a buffer of another size is refused, not overrun
    }
    if (!std::strcmp(ext, ".ppm")) {
        pifusion::OwnedImage im;
        if (pifusion::read_ppm_bgr(p, im)) C.ppm_ok++; else C.ppm_refused++;
    }
}

static void check_dataset(const std::vector<uint8_t>& cfg, const std::vector<uint8_t>& traj, const std::string& work)
{
    const std::string d = work + "/ds";
    mkdir(d.c_str(), 0755); mkdir((d + "/rgb").c_str(), 0755);
    write_file(d + "/config.cfg", cfg.data(), cfg.size());
    write_file(d + "/trajectory.txt", traj.data(), traj.size());
    pifusion::DroneMapDataset ds;
    if (!ds.open(d)) return;
    C.cfg++;
    (void)ds.cfg.GetVec("Plane"); (void)ds.cfg.GetVec("Camera.Paraments"); (void)ds.cfg.GetString("GPS.Origin", ""); (void)ds.cfg.GetInt("PrepareFrameNum", 10);
    (void)ds.cfg.GetDouble("Map2D.Scale", 1.0);
    std::pair<pifusion::OwnedImage, pi::SE3d> fr;
    std::vector<unsigned char> enc;
    for (int k = 0; k < 64 && ds.obtainFrame(fr, (k & 1) ? &enc : nullptr); k++) C.frames++;
}

static void mutate(std::vector<uint8_t>& s, bool jpeg)
{
    if (s.empty()) { s.push_back((uint8_t)rnd()); return; }
    size_t lo = 0;
    if (jpeg && (rnd() & 3)) {                                                         // three in four: inside the entropy-coded segment
        for (size_t i = 0; i + 1 < s.size(); i++) if (s[i] == 0xFF && s[i + 1] == 0xDA) { lo = std::min(s.size() - 1, i + 14); break; }
    }
    switch (rnd() % 7) {
    case 0: for (int k = 0, n = 1 + (int)rnd_below(4); k < n; k++) s[lo + rnd_below(s.size() - lo)] ^= (uint8_t)(1u << rnd_below(8)); break;
    case 1: for (int k = 0, n = 1 + (int)rnd_below(8); k < n; k++) s[lo + rnd_below(s.size() - lo)] = (uint8_t)rnd(); break;
    case 2: s.resize(1 + rnd_below(s.size())); break;
    case 3: { const size_t at = rnd_below(s.size()); std::vector<uint8_t> ins(1 + rnd_below(20)); for (auto& v : ins) v = (uint8_t)rnd(); s.insert(s.begin() + at, ins.begin(), ins.end()); } break;
    case 4: { const size_t a = rnd_below(s.size()), n = 1 + rnd_below(std::min<size_t>(64, s.size() - a)); std::vector<uint8_t> cp(s.begin() + a, s.begin() + a + n); s.insert(s.begin() + rnd_below(s.size()), cp.begin(), cp.end()); } break;
    case 5: { const size_t a = rnd_below(s.size()), n = 1 + rnd_below(std::min<size_t>(32, s.size() - a)); std::memset(s.data() + a, (rnd() & 1) ? 0 : 0xFF, n); } break;
    default: { const size_t a = rnd_below(s.size()); s[a] = 0xFF; if (a + 1 < s.size()) s[a + 1] = (uint8_t)(0xC0 + rnd_below(0x40)); } break;   // a marker where none belongs
    }
}

// a damaged PNG whose chunk checksums are right again: the damage reaches the chunk parsers and the inflate / unfilter stage
static void fix_png_crcs(std::vector<uint8_t>& s)
{
    size_t at = 8;
    while (at + 12 <= s.size()) {
        const size_t len = ((size_t)s[at] << 24) | ((size_t)s[at + 1] << 16) | ((size_t)s[at + 2] << 8) | s[at + 3];
        if (len > s.size() || at + 12 + len > s.size()) break;
        const uint32_t c = (uint32_t)crc32(0, s.data() + at + 4, (uInt)(len + 4));
        s[at + 8 + len] = (uint8_t)(c >> 24); s[at + 9 + len] = (uint8_t)(c >> 16); s[at + 10 + len] = (uint8_t)(c >> 8); s[at + 11 + len] = (uint8_t)c;
        at += 12 + len;
    }
}

static void run_one(const std::vector<uint8_t>& b, const std::string& name, const std::string& work, const std::vector<uint8_t>& cfg, const std::vector<uint8_t>& traj)
{
    const bool jpg = b.size() >= 2 && b[0] == 0xFF && b[1] == 0xD8, png = b.size() >= 4 && b[0] == 0x89 && b[1] == 'P';
    const bool ppm = b.size() >= 2 && b[0] == 'P' && b[1] == '6';
    if (jpg) { check_jpeg(b, name); check_files(b, work, ".jpg"); }
    else if (png) { check_png(b); check_files(b, work, ".png"); }
    else if (ppm) check_files(b, work, ".ppm");
    else {                                                                             // a text: as config.cfg and as trajectory.txt
        check_dataset(b, traj, work);
        check_dataset(cfg, b, work);
        pifusion::Config c; std::string line;
        for (uint8_t ch : b) { if (ch == '\n') { c.ParseLine(line); line.clear(); } else line.push_back((char)ch); }
        c.ParseLine(line);
    }
}

static void check_plans()
{
    // random tile lists on a small grid, 2..5 ranks: every rank's plan must agree with its peers'
    for (int it = 0; it < 300; it++) {
        const int n = 2 + (int)rnd_below(4), hq = (int)(rnd() & 1);
        std::vector<std::vector<pf::TileRec>> all(n);
        for (int y = 0; y < 7; y++) for (int x = 0; x < 9; x++) if (rnd() % 5) all[rnd_below(n)].push_back({ x - 3, y - 2, (int)(rnd() % 3 != 0) });
        std::vector<long long> caps(n);
        for (auto& c : caps) c = 1 + (long long)rnd_below(40);
        size_t hb[9]; for (int j = 0; j < 9; j++) hb[j] = j == 4 ? 0 : 64 + 16 * rnd_below(8);
        std::vector<pf::BlendPlan> plan(n);
        for (int me = 0; me < n; me++) pf::plan_blend(all, caps, me, hq != 0, hb, plan[me]);
        for (int a = 0; a < n; a++)
            for (int b = 0; b < n; b++) {
                if (plan[a].send_bytes[b] != plan[b].recv_bytes[a]) fail("plan_blend: send and receive sizes disagree", "plans");
                size_t sum = 0;
                for (auto& q : plan[a].send_req[b]) { if (q.out_off != sum) fail("plan_blend: strip offsets are not dense", "plans"); sum += hb[3 * (q.dy + 1) + q.dx + 1]; }
                if (sum != plan[a].send_bytes[b]) fail("plan_blend: requests do not add up to the bytes sent", "plans");
            }
        C.plans++;
    }
}

static void check_datatrans()
{
    DataTrans<int>& q = DataTrans<int>::Instance();
    for (int i = 0; i < 100; i++) q.product(i);
    if (q.size() != 30 || q.dropped() != 70) fail("DataTrans: capacity 30, drop-oldest", "datatrans");
    for (int i = 70; i < 100; i++) { int v = -1; q.consumption(v); if (v != i) fail("DataTrans: order", "datatrans"); }
}

static void check_writer(const std::string& work)
{
    for (int it = 0; it < 40; it++) {
        const int r = it < 36 ? 1 + (int)rnd_below(70) : 250 + (int)rnd_below(600), c = 1 + (int)rnd_below(90);      // the last few: several 256-row bands (the writer's threads)
        std::vector<uint8_t> px((size_t)r * c * 3);
        for (auto& v : px) v = (uint8_t)rnd();
        for (const char* ext : { ".png", ".ppm" }) {
            const std::string p = work + "/w" + ext;
            if (!pf_write_image(p.c_str(), px.data(), r, c)) { fail("pf_write_image failed", p); continue; }
            int rr = 0, cc = 0;
            std::vector<uint8_t> back((size_t)r * c * 3);
            if (!pf_image_info(p.c_str(), &rr, &cc) || rr != r || cc != c || !pf_read_image(p.c_str(), back.data(), r, c) || back != px) fail("image written and read back differs", p);
        }
    }
}

int main(int argc, char** argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: san_fuzz <corpus dir> <work dir> <seed> <mutants per file>\n"); return 2; }
    const std::string corpus = argv[1], work = argv[2];
    g_state = std::strtoull(argv[3], nullptr, 10);
    const int mutants = std::atoi(argv[4]);
    std::vector<std::pair<std::string, std::vector<uint8_t>>> files;
    DIR* d = opendir(corpus.c_str());
    if (!d) return 2;
    std::vector<std::string> names;
    while (dirent* e = readdir(d)) if (e->d_name[0] != '.') names.push_back(e->d_name);
    closedir(d);
    std::sort(names.begin(), names.end());
    std::vector<uint8_t> cfg, traj;
    for (auto& n : names) {
        std::vector<uint8_t> b;
        if (!pf::read_file_bytes((corpus + "/" + n).c_str(), b)) continue;
        if (n == "config.cfg") cfg = b;
        if (n == "trajectory.txt") traj = b;
        files.push_back({ n, b });
    }
    // the dataset's images: what the trajectory names
    mkdir((work + "/ds").c_str(), 0755); mkdir((work + "/ds/rgb").c_str(), 0755);
    for (auto& f : files) if (f.first.rfind("frame", 0) == 0) write_file(work + "/ds/rgb/" + f.first, f.second.data(), f.second.size());
    mkdir((work + "/ds/rgb/dirframe.jpg").c_str(), 0755);       // regression: a DIRECTORY where the trajectory names a frame (it opens; ftell() says LONG_MAX)
    for (auto& f : files) {
        C.files++;
        run_one(f.second, f.first, work, cfg, traj);
        const bool jpg = f.second.size() >= 2 && f.second[0] == 0xFF && f.second[1] == 0xD8;
        for (int m = 0; m < mutants; m++) {
            std::vector<uint8_t> s = f.second;
            for (int k = 0, n = 1 + (int)rnd_below(3); k < n; k++) mutate(s, jpg);
            if (s.size() > 8 && s[0] == 0x89 && s[1] == 'P' && (rnd() & 3)) fix_png_crcs(s);
            C.mutants++;
            run_one(s, f.first + "#" + std::to_string(m), work, cfg, traj);
            // a damaged image inside the dataset as well
            if (f.first.rfind("frame", 0) == 0 && (m & 15) == 0) { write_file(work + "/ds/rgb/" + f.first, s.data(), s.size()); check_dataset(cfg, traj, work); write_file(work + "/ds/rgb/" + f.first, f.second.data(), f.second.size()); }
        }
    }
    check_plans();
    check_datatrans();
    check_writer(work);
    std::printf("files %ld mutants %ld | jpeg decoded %ld refused %ld | parallel pass: eligible %ld accepted %ld to-serial %ld unequal %ld accepted-but-serial-refuses %ld | "
                "png decoded %ld refused %ld | ppm read %ld refused %ld | datasets opened %ld frames %ld | plans %ld | violations %d\n",
                C.files, C.mutants, C.jpeg_ok, C.jpeg_refused, C.par_eligible, C.par_accepted, C.par_to_serial, C.par_unequal, C.par_accept_serial_refuse,
                C.png_ok, C.png_refused, C.ppm_ok, C.ppm_refused, C.cfg, C.frames, C.plans, g_fail);
    return g_fail ? 1 : 0;
}
