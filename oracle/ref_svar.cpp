// ref_svar.cpp -- golden-vector generator for the dataset's config.cfg wire format (row f3).
//
// TEST INFRASTRUCTURE ONLY.  Compiled (oracle/Makefile target `ref`) against the reference's own header-only
// GSLAM/core/Svar.h, VecParament.h and SE3.h where they lie under /root/reference; the binary goes to oracle/_ref/ and
// never into git.  tests/golden/make_svar_vectors.py writes the config texts below to files, runs this program on
// them and commits texts + the values it prints as tests/golden/svar_vectors.json.
//
// What is pinned: what the reference's file driver gets out of a config.cfg -- svar.ParseFile, then
// svar.get_var<pi::SE3d>("Plane", pi::SE3d()), VecParament<double> of "Camera.Paraments", "GPS.Origin",
// svar.GetInt("PrepareFrameNum", 10) (backup/map2dfusion.cpp:153-192, Map2DFusion.cpp:163-206).
#include <GSLAM/core/Svar.h>
#include <GSLAM/core/VecParament.h>
#include <GSLAM/core/SE3.h>
#include <cstdio>

using GSLAM::Svar;

int main(int argc, char** argv)
{
    printf("[");
    for (int i = 1; i < argc; i++) {
        Svar var;                                   // a fresh variable table per file
        var.ParseFile(argv[i]);
        const pi::SE3d plane = var.get_var<pi::SE3d>("Plane", pi::SE3d());
        VecParament<double> cam = var.get_var("Camera.Paraments", VecParament<double>());
        const pi::Point3d t = plane.get_translation(); const pi::SO3d r = plane.get_rotation();
        printf("%s{\"has_plane\":%d,\"plane\":[%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g],\"camera\":[", i > 1 ? ",\n" : "",
               var.exist("Plane") ? 1 : 0, t.x, t.y, t.z, r.x, r.y, r.z, r.w);
        for (size_t k = 0; k < cam.size(); k++) printf("%s%.17g", k ? "," : "", cam[k]);
        printf("],\"has_gps\":%d,\"gps\":\"%s\",\"prepare\":%d,\"scale\":%.17g}", var.exist("GPS.Origin") ? 1 : 0,
               var.GetString("GPS.Origin", "").c_str(), var.GetInt("PrepareFrameNum", 10), var.GetDouble("Map2D.Scale", 1.0));
    }
    printf("]\n");
    return 0;
}
