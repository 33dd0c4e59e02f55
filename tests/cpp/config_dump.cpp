// Prints what include/pifusion/TestSystem.h's Config reads out of a config.cfg, in the shape of oracle/ref_svar.cpp's
// output (tests/test_config_format.py compares both with the values of the reference's own Svar parser).
#include <pifusion/TestSystem.h>
#include <cstdio>

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    pifusion::Config cfg;
    if (!cfg.ParseFile(argv[1])) return 3;
    std::vector<double> plane = cfg.GetVec("Plane"), cam = cfg.GetVec("Camera.Paraments");
    if (plane.size() != 7) { const double id[7] = { 0, 0, 0, 0, 0, 0, 1 }; plane.assign(id, id + 7); }      // pi::SE3d()
    std::printf("{\"has_plane\":%d,\"plane\":[", cfg.exist("Plane") ? 1 : 0);
    for (size_t k = 0; k < plane.size(); k++) std::printf("%s%.17g", k ? "," : "", plane[k]);
    std::printf("],\"camera\":[");
    for (size_t k = 0; k < cam.size(); k++) std::printf("%s%.17g", k ? "," : "", cam[k]);
    std::printf("],\"has_gps\":%d,\"gps\":\"%s\",\"prepare\":%d,\"scale\":%.17g}\n", cfg.exist("GPS.Origin") ? 1 : 0,
                cfg.GetString("GPS.Origin", "").c_str(), cfg.GetInt("PrepareFrameNum", 10), cfg.GetDouble("Map2D.Scale", 1.0));
    return 0;
}
