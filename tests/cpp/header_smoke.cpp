// Compiles the C++ face (include/pifusion/Map2D.h) the way the reference's driver uses
// Map2D (Map2DFusion/Map2DFusion.cpp:273-327): create, prepare(plane, camera, frames),
// feed(img, pose), queueSize, save.  With -DUSE_REFERENCE_SE3 the pose type is the
// reference's own pi::SE3d (GSLAM/core/SE3.h) -- the drop-in case.
#ifdef USE_REFERENCE_SE3
#include <GSLAM/core/SE3.h>
#endif
#include <pifusion/Map2D.h>
#include <cstdio>
#include <vector>

int main()
{
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
    Map2D::Ele e;
    std::printf("tiles refreshed: %d\n", n);
    return n > 0 ? 0 : 6;
}
