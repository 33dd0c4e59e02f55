// Map2D.h -- C++ face of libpifusion.so with the reference's own interface
// (Map2DFusion/Map2D.h:37-98): PinHoleParameters, Map2D::create / prepare / feed /
// draw / save / queueSize, plus the MultiBandMap2DCPU::Ele tile surface
// (Map2DFusion/MultiBandMap2DCPU.h:32-51).  Header-only over the C ABI (pifusion.h),
// so there is exactly one binary boundary.
//
// Drop-in notes
//   * poses are taken from ANY type with get_translation() / get_rotation() whose
//     members are x,y,z(,w) -- the reference's own pi::SE3d works unchanged; a minimal
//     pi::SE3d is provided when GSLAM is not on the include path.
//   * images are taken from ANY type with rows, cols, type(), data (cv::Mat) or
//     rows, cols, type(), data (GSLAM::GImage); pixels are copied before feed() returns,
//     so the caller's refcounted buffer may go away (the reference keeps a cv::Mat
//     reference instead, MultiBandMap2DCPU.cpp:297-301).
//   * draw() has no GL here: it refreshes the blended tiles (the texture upload's
//     pixel work, .cpp:705-742) and hands them to an optional callback.
#ifndef PIFUSION_MAP2D_H
#define PIFUSION_MAP2D_H
#include "../pifusion.h"
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#ifndef ELE_PIXELS
#define ELE_PIXELS 256
#endif

#ifndef GSLAM_SE3_H          // the reference's header defines pi::SE3d itself
namespace pi {
struct Point3d { double x = 0, y = 0, z = 0; Point3d() {} Point3d(double X, double Y, double Z) : x(X), y(Y), z(Z) {} };
struct SO3d { double x = 0, y = 0, z = 0, w = 1; SO3d() {} SO3d(double X, double Y, double Z, double W) : x(X), y(Y), z(Z), w(W) {} };
class SE3d {
public:
    SE3d() {}
    SE3d(double x, double y, double z, double wx, double wy, double wz, double w) : r_(wx, wy, wz, w), t_(x, y, z) {}
    SE3d(const SO3d& r, const Point3d& t) : r_(r), t_(t) {}
    const SO3d& get_rotation() const { return r_; }
    const Point3d& get_translation() const { return t_; }
private:
    SO3d r_; Point3d t_;
};
}  // namespace pi
#endif

struct PinHoleParameters {        // Map2DFusion/Map2D.h:37-43
    PinHoleParameters() {}
    PinHoleParameters(int _w, int _h, double _fx, double _fy, double _cx, double _cy)
        : w(_w), h(_h), fx(_fx), fy(_fy), cx(_cx), cy(_cy) {}
    double w = 0, h = 0, fx = 0, fy = 0, cx = 0, cy = 0;
};

namespace pifusion {

// non-owning image view with cv::Mat / GImage field names
struct ImageView {
    int rows = 0, cols = 0, flags = 0;
    unsigned char* data = nullptr;
    size_t step = 0;
    ImageView() {}
    ImageView(int r, int c, int t, void* d, size_t s = 0) : rows(r), cols(c), flags(t), data((unsigned char*)d), step(s) {}
    int type() const { return flags; }
};

template <class SE3> inline void pose7(const SE3& p, double o[7])
{
    const auto& t = p.get_translation(); const auto& r = p.get_rotation();
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = r.x; o[4] = r.y; o[5] = r.z; o[6] = r.w;
}
// row pitch of the caller's image: cv::Mat::step (MatStep converts to size_t) and ImageView::step are carried over,
// so a ROI of a wider buffer or a row-padded image is read with its own pitch; types without a step member
// (GSLAM::GImage is always packed, GImage.h:387-389) give 0 = packed
template <class Img> inline auto step_of(const Img& m, int) -> decltype(size_t(m.step)) { return size_t(m.step); }
template <class Img> inline size_t step_of(const Img&, long) { return 0; }
template <class Img> inline pf_image view(const Img& m)
{
    pf_image v; v.rows = m.rows; v.cols = m.cols; v.type = m.type() & 0xfff; v.data = m.data; v.step = step_of(m, 0);
    return v;
}

}  // namespace pifusion

class Map2D {
public:
    enum Map2DType { NoType = 0, TypeCPU = 1, TypeGPU = 2, TypeMultiBandCPU = 3, TypeRender = 4 };   // Map2D.h:83

    // MultiBandMap2DCPU::MultiBandMap2DCPUEle, read side
    struct Ele {
        int ix = 0, iy = 0;
        std::vector<std::vector<unsigned char>> pyr_laplace;   // level i: (256>>i)^2 x 3 x {int16|float}
        std::vector<std::vector<float>> weights;               // level i: (256>>i)^2
    };

    static std::shared_ptr<Map2D> create(int type = TypeCPU, bool thread = true, const pf_options* opt = nullptr)
    {
        pf_map* h = pf_create(type, thread ? 1 : 0, opt);
        return h ? std::shared_ptr<Map2D>(new Map2D(h)) : std::shared_ptr<Map2D>();
    }
    ~Map2D() { pf_host_free(tex_); pf_destroy(h_); }
    Map2D(const Map2D&) = delete;
    Map2D& operator=(const Map2D&) = delete;

    template <class Img, class SE3>
    bool prepare(const SE3& plane, const PinHoleParameters& camera, const std::deque<std::pair<Img, SE3>>& frames)
    {
        double pl[7]; pifusion::pose7(plane, pl);
        const double cam[6] = { camera.w, camera.h, camera.fx, camera.fy, camera.cx, camera.cy };
        std::vector<double> poses(frames.size() * 7);
        std::vector<pf_image> imgs(frames.size());
        size_t i = 0;
        for (const auto& f : frames) { imgs[i] = pifusion::view(f.first); pifusion::pose7(f.second, &poses[7 * i]); i++; }
        return pf_prepare(h_, pl, cam, (int)frames.size(), imgs.data(), poses.data()) != 0;
    }
    template <class Img, class SE3> bool feed(const Img& img, const SE3& pose)      // Map2D.h:91
    {
        double p[7]; pifusion::pose7(pose, p);
        const pf_image v = pifusion::view(img);
        return pf_feed(h_, &v, p) != 0;
    }
    // feed(cv::imread(file), pose) with the file's bytes instead of its pixels (backup/map2dfusion.cpp:129-135): the JPEG stream is
    // decoded straight into HBM (Huffman on this thread, IDCT / upsampling / colour on the GPU) and rendered from there
    template <class SE3> bool feedJpeg(const unsigned char* data, size_t len, const SE3& pose)
    {
        double p[7]; pifusion::pose7(pose, p);
        return pf_feed_jpeg(h_, data, len, p) != 0;
    }
    // draw(): refresh every tile whose Ischanged flag is set and pass (ix, iy, BGR8 256x256) on.
    // With fuseGoogle() set, `announce` receives what the reference hands to scommand.Call("MapWidget", ...) for every refreshed
    // tile that is not on the rim of the grid (MultiBandMap2DCPU.cpp:744-757): "Map2DUpdate LastTexMat <gpsTL> <gpsBR>".
    void draw(const std::function<void(int, int, const unsigned char*)>& sink = nullptr,
              const std::function<void(const std::string&)>& announce = nullptr)
    {
        const int cap = pf_tile_count(h_);
        if (cap <= 0) return;
        std::vector<int> xy(2 * (size_t)cap);
        // the tiles come back in ONE launch and one PCIe transfer into a page-locked buffer that this object keeps between draws
        // (pf_host_alloc: filled straight from HBM; the textures updateTexture hands to GL, .cpp:159-176)
        const size_t need = (size_t)cap * ELE_PIXELS * ELE_PIXELS * 3;
        if (tex_cap_ < need) {
            pf_host_free(tex_); tex_cap_ = 0;
            tex_ = (unsigned char*)pf_host_alloc(need + need / 4);
            if (!tex_) return;
            tex_cap_ = need + need / 4;
        }
        unsigned char* px = tex_;
        const int n = pf_blend_changed(h_, xy.data(), px, cap);
        for (int i = 0; i < n; i++) {
            if (sink) sink(xy[2 * i], xy[2 * i + 1], px + (size_t)i * ELE_PIXELS * ELE_PIXELS * 3);
            char cmd[256];
            if (fuse2google_ && announce && pf_map_update_command(h_, xy[2 * i], xy[2 * i + 1], gps_origin_, cmd, (int)sizeof cmd) > 0) announce(cmd);
        }
    }
    // svar "Fuse2Google" and "GPS.Origin" (longitude latitude altitude) of the reference (.cpp:196, :744)
    void fuseGoogle(bool on, double lng = 0, double lat = 0, double alt = 0) { fuse2google_ = on; gps_origin_[0] = lng; gps_origin_[1] = lat; gps_origin_[2] = alt; }
    bool save(const std::string& filename) { return pf_save(h_, filename.c_str()) != 0; }
    unsigned queueSize() { return pf_queue_size(h_); }
    bool sync() { return pf_sync(h_) != 0; }

    // Ele access (D2H copy of one tile's pyramids)
    bool ele(int ix, int iy, Ele& e)
    {
        const int nl = pf_num_levels(h_), es = pf_pyramid_type(h_) == PF_32FC3 ? 4 : 2;
        e.ix = ix; e.iy = iy; e.pyr_laplace.resize(nl); e.weights.resize(nl);
        for (int i = 0; i < nl; i++) {
            const size_t n = (size_t)(ELE_PIXELS >> i) * (ELE_PIXELS >> i);
            e.pyr_laplace[i].resize(n * 3 * es); e.weights[i].resize(n);
            if (!pf_get_tile_level(h_, ix, iy, i, e.pyr_laplace[i].data(), e.weights[i].data())) return false;
        }
        return true;
    }
    bool blend(int ix, int iy, unsigned char* bgr256) { return pf_blend_tile(h_, ix, iy, bgr256) != 0; }
    pf_map* handle() { return h_; }

private:
    explicit Map2D(pf_map* h) : h_(h) {}
    pf_map* h_;
    unsigned char* tex_ = nullptr; size_t tex_cap_ = 0;      // draw()'s page-locked tile buffer
    bool fuse2google_ = false;
    double gps_origin_[3] = { 0, 0, 0 };
};

#endif  // PIFUSION_MAP2D_H
