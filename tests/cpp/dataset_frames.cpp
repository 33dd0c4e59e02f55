// DroneMapDataset::obtainFrame (include/pifusion/TestSystem.h; backup/map2dfusion.cpp:122-135) on a dataset given on the
// command line: one line per frame -- name-order index, rows, cols, FNV-1a of the BGR pixels, the pose's z.
#include <pifusion/TestSystem.h>
#include <cstdio>

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    pifusion::DroneMapDataset ds;
    if (!ds.open(argv[1])) return 1;
    std::pair<pifusion::OwnedImage, pi::SE3d> frame;
    int k = 0;
    while (ds.obtainFrame(frame)) {
        unsigned long long h = 1469598103934665603ull;
        const size_t n = (size_t)frame.first.rows * frame.first.cols * 3;
        for (size_t i = 0; i < n; i++) { h ^= frame.first.data[i]; h *= 1099511628211ull; }
        std::printf("%d %d %d %llu\n", k++, frame.first.rows, frame.first.cols, h);
    }
    std::printf("frames %d\n", k);
    return 0;
}
