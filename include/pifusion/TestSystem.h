// TestSystem.h -- the file-driven caller of Map2D, in C++ like the reference's own
// (backup/map2dfusion.cpp:122-135 obtainFrame, :137-230 testMap2D; live variant
// Map2DFusion/Map2DFusion.cpp:250-329): read `<datapath>/config.cfg`, take keyframes from
// `<datapath>/trajectory.txt` (`name x y z qx qy qz qw` per line) and `<datapath>/rgb/<name>.*`,
// size the grid with the first PrepareFrameNum frames, then feed while `queueSize() < 2`, paced at
// Video.fps, and save() to Map.File2Save at the end.
//
// Header-only over include/pifusion/Map2D.h.  The reference decodes frames with cv::imread; here
// <name>.jpg goes through the library's own JPEG decoder (pf_read_image: libjpeg's default decode byte
// for byte, csrc/jpeg_decode.cpp), <name>.png or <name>.ppm (binary P6) is read when there is no .jpg, and a decoder
// hook (`DroneMapDataset::decoder`) takes over the .jpg when set -- cv::imread plugs in there.
#ifndef PIFUSION_TESTSYSTEM_H
#define PIFUSION_TESTSYSTEM_H
#include "Map2D.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <thread>

namespace pifusion {

// owning BGR8 image with cv::Mat's field names (rows, cols, type(), data, step)
struct OwnedImage {
    int rows = 0, cols = 0, flags = PF_8UC3;
    unsigned char* data = nullptr;
    size_t step = 0;
    std::shared_ptr<std::vector<unsigned char>> store;
    int  type() const { return flags; }
    bool empty() const { return !data; }
    void create(int r, int c, int t = PF_8UC3)
    {
        rows = r; cols = c; flags = t; step = (size_t)c * (t == PF_8UC4 ? 4 : 3);
        store = std::make_shared<std::vector<unsigned char>>((size_t)r * step);
        data = store->data();
    }
};

// the subset of the svar grammar the dataset files and the command line use:
// `key = value`, `key ?= value` (assign if unset), `//` and `#` comments (GSLAM/core/Svar.h:1126-1184)
class Config {
public:
    bool ParseLine(std::string line)
    {
        size_t c = line.find("//"); if (c != std::string::npos) line.erase(c);
        c = line.find('#'); if (c != std::string::npos) line.erase(c);
        const size_t eq = line.find('=');
        if (eq == std::string::npos || eq == 0) return false;
        const bool weak = line[eq - 1] == '?';
        std::string key = trim(line.substr(0, weak ? eq - 1 : eq)), val = trim(line.substr(eq + 1));
        if (key.empty()) return false;
        if (weak && kv_.count(key)) return true;
        kv_[key] = val;
        return true;
    }
    bool ParseFile(const std::string& path)
    {
        std::ifstream f(path.c_str());
        if (!f.is_open()) return false;
        std::string line;
        while (std::getline(f, line)) ParseLine(line);
        return true;
    }
    bool exist(const std::string& k) const { return kv_.count(k) != 0; }
    std::string GetString(const std::string& k, const std::string& def) const { auto it = kv_.find(k); return it == kv_.end() ? def : it->second; }
    double GetDouble(const std::string& k, double def) const { auto it = kv_.find(k); return it == kv_.end() ? def : std::atof(it->second.c_str()); }
    int GetInt(const std::string& k, int def) const { return (int)GetDouble(k, def); }
    // numbers of a value like `[4000 3000 3000 3000 2000 1500]` or `0 0 0 0 0 0 1`
    std::vector<double> GetVec(const std::string& k) const
    {
        std::vector<double> v;
        std::string s = GetString(k, "");
        for (auto& ch : s) if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
        std::stringstream ss(s);
        double d;
        while (ss >> d) v.push_back(d);
        return v;
    }
    const std::map<std::string, std::string>& all() const { return kv_; }
private:
    static std::string trim(const std::string& s)
    {
        const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }
    std::map<std::string, std::string> kv_;
};

// binary PPM (P6, maxval 255) -> BGR8, the channel order cv::imread returns
inline bool read_ppm_bgr(const std::string& path, OwnedImage& out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = { 0, 0, 0 };
    int w = 0, h = 0, maxv = 0;
    auto next_int = [&](int& v) {
        int ch = std::fgetc(f);
        for (;;) {
            while (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') ch = std::fgetc(f);
            if (ch == '#') { while (ch != '\n' && ch != EOF) ch = std::fgetc(f); continue; }
            break;
        }
        if (ch < '0' || ch > '9') return false;
        v = 0;
        while (ch >= '0' && ch <= '9') {
            if (v > (1 << 26)) return false;          // not a picture size (and the digit loop must not overflow)
            v = v * 10 + (ch - '0'); ch = std::fgetc(f);
        }
        return true;          // the single whitespace after the last header field has been consumed
    };
    bool ok = std::fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '6' && next_int(w) && next_int(h) && next_int(maxv) &&
              w > 0 && h > 0 && maxv == 255 && (long long)w * h <= (1ll << 28);
    if (ok) {
        // the pixels must be in the file before memory is asked for them (a damaged header must not allocate gigabytes)
        const long here = std::ftell(f);
        std::fseek(f, 0, SEEK_END);
        const long end = std::ftell(f);
        std::fseek(f, here, SEEK_SET);
        ok = here >= 0 && end - here >= (long)((size_t)w * h * 3);
    }
    if (ok) {
        out.create(h, w, PF_8UC3);
        const size_t want = (size_t)w * h * 3, got_n = std::fread(out.data, 1, want, f);
        ok = got_n == want;
        for (size_t i = 0; ok && i < (size_t)w * h; i++) std::swap(out.data[3 * i], out.data[3 * i + 2]);
    }
    std::fclose(f);
    return ok;
}

class DroneMapDataset {
public:
    Config      cfg;
    std::string datapath;
    // when set, decodes <name>.jpg instead of pf_read_image (the reference: cv::imread(imgfile)); returns false when it cannot
    std::function<bool(const std::string& file, OwnedImage& out)> decoder;

    bool open(const std::string& path)
    {
        datapath = path;
        cfg.ParseFile(datapath + "/config.cfg");
        in_.reset(new std::ifstream((datapath + "/trajectory.txt").c_str()));
        if (!in_->is_open()) { std::cerr << "Can't open file " << (datapath + "/trajectory.txt") << std::endl; return false; }
        return true;
    }
    // backup/map2dfusion.cpp:122-135.  With `encoded` given (and no decoder hook), a frame that exists as <name>.jpg is handed over
    // as the file's bytes (frame.first stays empty) for Map2D::feedJpeg to decode on the GPU.
    bool obtainFrame(std::pair<OwnedImage, pi::SE3d>& frame, std::vector<unsigned char>* encoded = nullptr)
    {
        std::string line;
        if (!in_ || !std::getline(*in_, line)) return false;
        std::stringstream ifs(line);
        std::string name;
        ifs >> name;
        double p[7];
        for (int i = 0; i < 7; i++) if (!(ifs >> p[i])) return false;
        const std::string base = datapath + "/rgb/" + name;
        frame.first = OwnedImage();
        if (encoded) encoded->clear();
        if (encoded && !decoder && read_bytes(base + ".jpg", *encoded)) { /* decoded by the map */ }
        else if (!(decoder ? decoder(base + ".jpg", frame.first) : read_native(base + ".jpg", frame.first)) &&
                 !read_native(base + ".png", frame.first) && !read_ppm_bgr(base + ".ppm", frame.first)) return false;
        frame.second = pi::SE3d(p[0], p[1], p[2], p[3], p[4], p[5], p[6]);      // SE3 stream order x y z qx qy qz qw (SE3.h:112-117)
        return true;
    }
    static bool read_bytes(const std::string& file, std::vector<unsigned char>& out)
    {
        FILE* f = std::fopen(file.c_str(), "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        const bool sane = n > 0 && n <= (1L << 31);                      // (a directory opens, and ftell() then says LONG_MAX)
        out.resize(sane ? (size_t)n : 0);
        const bool ok = sane && std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
        std::fclose(f);
        if (!ok) out.clear();
        return ok;
    }
    // cv::imread(file) through the library: JPEG (or PPM) -> BGR8
    static bool read_native(const std::string& file, OwnedImage& out)
    {
        int rows = 0, cols = 0;
        FILE* f = std::fopen(file.c_str(), "rb");                      // a dataset without this .jpg is not an error yet: the .ppm is tried next
        if (!f) return false;
        std::fclose(f);
        if (!pf_image_info(file.c_str(), &rows, &cols)) return false;
        out.create(rows, cols, PF_8UC3);
        return pf_read_image(file.c_str(), out.data, rows, cols) != 0;
    }
private:
    std::shared_ptr<std::ifstream> in_;
};

// testMap2D (backup/map2dfusion.cpp:137-230) without the window: returns 0 on success, the reference's
// negative codes otherwise.  `args` are svar-style `key=value` overrides parsed after config.cfg.
inline int testMap2D(const std::string& datapath, const std::vector<std::string>& args, std::shared_ptr<Map2D>* out_map = nullptr)
{
    std::cout << "Act=TestMap2D\n";
    if (datapath.empty()) { std::cerr << "Map2D.DataPath is not seted!\n"; return -1; }
    DroneMapDataset ds;
    if (!ds.open(datapath)) return -3;
    for (const auto& a : args) ds.cfg.ParseLine(a);
    const Config& svar = ds.cfg;

    std::deque<std::pair<OwnedImage, pi::SE3d>> frames;
    for (int i = 0, iend = svar.GetInt("PrepareFrameNum", 10); i < iend; i++) {
        std::pair<OwnedImage, pi::SE3d> frame;
        if (!ds.obtainFrame(frame)) break;
        frames.push_back(frame);
    }
    std::cout << "Loaded " << frames.size() << " frames.\n";
    if (frames.empty()) return -4;

    pf_options opt;
    pf_default_options(&opt);
    for (const auto& kv : svar.all()) pf_options_set(&opt, kv.first.c_str(), kv.second.c_str());
    std::shared_ptr<Map2D> map = Map2D::create(svar.GetInt("Map2D.Type", Map2D::TypeMultiBandCPU), svar.GetInt("Map2D.Thread", 1) != 0, &opt);
    if (!map) { std::cerr << "No map2d created!\n"; return -5; }
    const std::vector<double> vecP = svar.GetVec("Camera.Paraments");
    if (vecP.size() != 6) { std::cerr << "Invalid camera parameters!\n"; return -5; }
    const std::vector<double> pl = svar.GetVec("Plane");
    const pi::SE3d plane = pl.size() == 7 ? pi::SE3d(pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], pl[6]) : pi::SE3d();
    if (!map->prepare(plane, PinHoleParameters((int)vecP[0], (int)vecP[1], vecP[2], vecP[3], vecP[4], vecP[5]), frames)) return -6;

    long fed = 0;
    if (svar.GetInt("AutoFeedFrames", 1)) {
        const int fps = svar.GetInt("Video.fps", 100);
        const auto period = std::chrono::microseconds(fps > 0 ? 1000000 / fps : 0);
        // .jpg frames go to the map as the file's bytes (decoded on the GPU) unless Map2D.DecodeOnHost=1 asks for the reference's order
        const bool on_device = svar.GetInt("Map2D.DecodeOnHost", 0) == 0;
        std::vector<unsigned char> encoded;
        for (;;) {                                                     // Map2DFusion.cpp:311-327
            const auto t0 = std::chrono::steady_clock::now();
            if (map->queueSize() < 2) {
                std::pair<OwnedImage, pi::SE3d> frame;
                if (!ds.obtainFrame(frame, on_device ? &encoded : nullptr)) break;
                if (!encoded.empty()) map->feedJpeg(encoded.data(), encoded.size(), frame.second);
                else map->feed(frame.first, frame.second);
                fed++;
            }
            if (fps > 0) std::this_thread::sleep_until(t0 + period);
        }
    }
    map->sync();
    std::cout << "Fed " << fed << " frames.\n";
    const std::string file = svar.GetString("Map.File2Save", "");
    if (!file.empty() && !map->save(file)) return -7;                  // TestSystem's destructor: map->save(Map.File2Save)
    if (out_map) *out_map = map;
    return 0;
}

}  // namespace pifusion
#endif  // PIFUSION_TESTSYSTEM_H
