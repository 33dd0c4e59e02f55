// pifusion_replay -- the reference's file driver (backup/map2dfusion.cpp testMap2D) over libpifusion.so:
//
//     pifusion_replay <datapath> [key=value ...]
//
// keys are the reference's svar names: Map2D.Type (3), Map2D.Thread (1), PrepareFrameNum (10), Video.fps (100, 0 = unpaced),
// Map.File2Save, MultiBandMap2DCPU.ForceFloat, MultiBandMap2DCPU.BandNumber, Map2D.Scale, Result.BackGroundColor ...
// Build: g++ -std=c++11 -Iinclude tools/cpp/pifusion_replay.cpp -Lpi-slam-fusion_amd -l:libpifusion.so -lpthread
#include <pifusion/TestSystem.h>

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <datapath> [key=value ...]\n", argv[0]); return 2; }
    std::vector<std::string> args;
    for (int i = 2; i < argc; i++) args.push_back(argv[i]);
    std::shared_ptr<Map2D> map;
    const int rc = pifusion::testMap2D(argv[1], args, &map);
    if (rc) std::fprintf(stderr, "testMap2D failed: %d (%s)\n", rc, pf_last_error());
    return rc ? 1 : 0;
}
