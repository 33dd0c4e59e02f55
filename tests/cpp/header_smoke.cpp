// Compiles the C++ face (include/pifusion/Map2D.h) the way the reference's driver uses
// Map2D (Map2DFusion/Map2DFusion.cpp:273-327): create, prepare(plane, camera, frames),
// feed(img, pose), queueSize, save.  With -DUSE_REFERENCE_SE3 the pose type is the
// reference's own pi::SE3d (GSLAM/core/SE3.h) -- the drop-in case.
#ifdef USE_REFERENCE_SE3
#include <GSLAM/core/SE3.h>
#endif
#include <pifusion/Map2D.h>
#include <pifusion/DataTrans.h>
#include <cstdio>
#include <thread>
#include <vector>

// the tracker -> fusion queue (src/DataTrans.h:40-83): capacity 30, the oldest element is dropped, consumption blocks
static int datatrans_check()
{
    typedef DataTrans<std::pair<int, pi::SE3d> > Wire;
    Wire& w = Wire::Instance();
    if (&w != &Wire::Instance()) return 20;                      // singleton per payload type
    for (int k = 0; k < 35; k++) w.product(std::make_pair(k, pi::SE3d(k, 0, -100, 0, 0, 0, 1)));
    if (w.size() != 30 || w.dropped() != 5) return 21;
    std::pair<int, pi::SE3d> v;
    w.consumption(v);
    if (v.first != 5) return 22;                                 // 0..4 were dropped
    while (w.size()) w.consumption(v);
    if (v.first != 34) return 23;
    int got = -1;
    std::thread consumer([&] { std::pair<int, pi::SE3d> q; w.consumption(q); got = q.first; });   // blocks until the product below
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    w.product(std::make_pair(77, pi::SE3d()));
    consumer.join();
    return got == 77 ? 0 : 24;
}

int main()
{
    if (int rc = datatrans_check()) { std::printf("DataTrans check failed: %d\n", rc); return rc; }
    std::shared_ptr<Map2D> none = Map2D::create(Map2D::NoType, false);
    if (none) return 2;                                  // Map2D.cpp:53
    std::shared_ptr<Map2D> map = Map2D::create(Map2D::TypeMultiBandCPU, false);
    if (!map) { std::printf("no device: create() returned null\n"); return 0; }   // CPU-only box: fails loudly, no fallback
    std::vector<unsigned char> px(480 * 640 * 3, 90);
    pifusion::ImageView img(480, 640, PF_8UC3, px.data());
    std::deque<std::pair<pifusion::ImageView, pi::SE3d>> frames;
    for (int k = 0; k < 3; k++) frames.push_back(std::make_pair(img, pi::SE3d(10. * k, 0, -100, 0, 0, 0, 1)));
    if (!map->prepare(pi::SE3d(), PinHoleParameters(640, 480, 500, 500, 320, 240), frames)) return 3;
    for (auto& f : frames) if (!map->feed(f.first, f.second)) return 4;
    if (!map->sync() || map->queueSize() != 0) return 5;
    int n = 0;
    map->draw([&](int, int, const unsigned char* bgr) { n += bgr[0] >= 0; });
    std::printf("tiles refreshed: %d\n", n);
    if (n <= 0) return 6;
    // the overlay message of draw() (MultiBandMap2DCPU.cpp:744-757): every refreshed tile off the rim of the grid is announced
    {
        for (auto& f : frames) if (!map->feed(f.first, f.second)) return 12;
        int told = 0, bad = 0;
        map->fuseGoogle(true, 108.888931, 34.257287, 400.0);
        map->draw(nullptr, [&](const std::string& s) { told++; bad += s.compare(0, 23, "Map2DUpdate LastTexMat ") != 0; std::printf("%s\n", s.c_str()); });
        std::printf("tiles announced: %d\n", told);
        if (bad) return 13;
    }

    // A ROI of a wider buffer (cv::Mat::step > cols * channels) through the C++ face: the row pitch must travel.
    // Same pixels packed and padded -> the same tiles.
    {
        std::vector<unsigned char> packed(480 * 640 * 3), wide(480 * 2100, 7);
        for (size_t i = 0; i < packed.size(); i++) packed[i] = (unsigned char)((i * 2654435761u) >> 13);
        for (int y = 0; y < 480; y++) std::copy(packed.begin() + y * 1920, packed.begin() + (y + 1) * 1920, wide.begin() + y * 2100 + 90);
        std::shared_ptr<Map2D> a = Map2D::create(Map2D::TypeMultiBandCPU, false), b = Map2D::create(Map2D::TypeMultiBandCPU, false);
        pifusion::ImageView pk(480, 640, PF_8UC3, packed.data()), roi(480, 640, PF_8UC3, wide.data() + 90, 2100);
        std::deque<std::pair<pifusion::ImageView, pi::SE3d>> fa(1, std::make_pair(pk, pi::SE3d(0, 0, -100, 0, 0, 0.0436194, 0.9990482)));
        if (!a->prepare(pi::SE3d(), PinHoleParameters(640, 480, 500, 500, 320, 240), fa) ||
            !b->prepare(pi::SE3d(), PinHoleParameters(640, 480, 500, 500, 320, 240), fa)) return 7;
        if (!a->feed(pk, fa[0].second) || !b->feed(roi, fa[0].second) || !a->sync() || !b->sync()) return 8;
        Map2D::Ele ea, eb;
        int tiles = 0;
        for (int iy = -4; iy < 12; iy++)
            for (int ix = -4; ix < 12; ix++) {
                const bool ha = a->ele(ix, iy, ea), hb = b->ele(ix, iy, eb);
                if (ha != hb) return 9;
                if (!ha) continue;
                tiles++;
                if (ea.pyr_laplace != eb.pyr_laplace || ea.weights != eb.weights) { std::printf("padded view differs at tile %d,%d\n", ix, iy); return 10; }
            }
        std::printf("padded view == packed view on %d tiles\n", tiles);
        if (!tiles) return 11;
    }
    return 0;
}
