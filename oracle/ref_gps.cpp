// ref_gps.cpp -- golden-vector generator for the output side of row f4: the "Map2DUpdate LastTexMat" message the
// reference emits for a refreshed tile (Map2DFusion/MultiBandMap2DCPU.cpp:744-757).
//
// TEST INFRASTRUCTURE ONLY.  Compiled (oracle/Makefile target `ref`) TOGETHER WITH the reference's own
// PIL/src/hardware/Gps/utils_GPS.cpp, against its GSLAM/core/SE3.h / Point.h where they lie under /root/reference; the binary goes to
// oracle/_ref/ and never into git.  tests/golden/make_gps_vectors.py runs it and commits what it prints as
// tests/golden/gps_vectors.json.
//
// What the reference's own code computes here (this file only lays the operands out the way the call site does and prints):
//   * pi::calcLngLatFromDistance (utils_GPS.cpp:133-160), truncated DEG2RAD and all;
//   * pi::SE3d * pi::Point3d (GSLAM/core/SE3.h) for plane * (x, y, 0);
//   * operator<<(ostream&, pi::Point3d) (GSLAM/core/Point.h:166-170) = std::to_string per field: SIX decimals -- the
//     `setiosflags(ios::fixed) << setprecision(9)` of the call site act on nothing, the stream only ever sees strings.
// The operands follow MultiBandMap2DCPU.cpp:709-712: the tile's corners are rounded to FLOAT before they meet the plane.
#include <GSLAM/core/SE3.h>
#include <hardware/Gps/utils_GPS.h>
#include <cstdint>
#include <cstdio>
#include <iomanip>
#include <sstream>
#include <string>

static uint64_t s_state = 20261004;
static uint64_t splitmix64()
{
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double urand(double a, double b) { return a + (b - a) * ((splitmix64() >> 11) * (1.0 / 9007199254740992.0)); }

int main()
{
    printf("{\"lnglat\":[\n");
    const int nl = 24;
    for (int i = 0; i < nl; i++) {
        // origins over both hemispheres and near the poles / the equator; offsets of a survey flight
        const double lng1 = i == 0 ? 108.888931 : urand(-180, 180), lat1 = i == 0 ? 34.257287 : (i == 1 ? 0.0 : (i == 2 ? 89.5 : urand(-85, 85)));
        const double dx = i < 3 ? (i == 0 ? 0.0 : 92.0) : urand(-5000, 5000), dy = i < 3 ? (i == 0 ? 0.0 : 110.9) : urand(-5000, 5000);
        double lng2 = 0, lat2 = 0;
        pi::calcLngLatFromDistance(lng1, lat1, dx, dy, lng2, lat2);
        printf("  {\"lng1\":%.17g,\"lat1\":%.17g,\"dx\":%.17g,\"dy\":%.17g,\"lng2\":%.17g,\"lat2\":%.17g}%s\n", lng1, lat1, dx, dy, lng2, lat2, i + 1 < nl ? "," : "");
    }
    printf("],\n\"messages\":[\n");
    const int nm = 24;
    for (int i = 0; i < nm; i++) {
        // a plane as the RANSAC producer publishes it (small tilt, any heading), the grid of a prepared map, one tile of it
        pi::SE3d plane;
        if (i > 0) {
            double q[4] = { urand(-0.05, 0.05), urand(-0.05, 0.05), urand(-1, 1), urand(-1, 1) };
            if (i % 5 == 4) { q[0] = urand(-1, 1); q[1] = urand(-1, 1); }          // and a few arbitrary ones
            const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            plane = pi::SE3d(pi::SO3d(q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq), pi::Point3d(urand(-300, 300), urand(-300, 300), urand(-20, 20)));
        }
        const pi::Point3d origin(i == 0 ? 108.888931 : urand(-180, 180), i == 0 ? 34.257287 : urand(-80, 80), urand(0, 500));
        const double min_x = urand(-2000, 2000), min_y = urand(-2000, 2000), ele = 256 * urand(0.01, 0.3);
        const int x = (int)(splitmix64() % 40), y = (int)(splitmix64() % 40);
        // MultiBandMap2DCPU.cpp:709-712
        float x0 = min_x + x * ele;
        float y0 = min_y + y * ele;
        float x1 = x0 + ele;
        float y1 = y0 + ele;
        // MultiBandMap2DCPU.cpp:747-755
        std::stringstream cmd;
        pi::Point3d worldTl = plane * pi::Point3d(x0, y0, 0);
        pi::Point3d worldBr = plane * pi::Point3d(x1, y1, 0);
        pi::Point3d gpsTl, gpsBr;
        pi::calcLngLatFromDistance(origin.x, origin.y, worldTl.x, worldTl.y, gpsTl.x, gpsTl.y);
        pi::calcLngLatFromDistance(origin.x, origin.y, worldBr.x, worldBr.y, gpsBr.x, gpsBr.y);
        cmd << "Map2DUpdate LastTexMat " << std::setiosflags(std::ios::fixed) << std::setprecision(9) << gpsTl << " " << gpsBr;
        const pi::Point3d t = plane.get_translation();
        const pi::SO3d r = plane.get_rotation();
        printf("  {\"plane\":[%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g],\"origin\":[%.17g,%.17g,%.17g],\"min\":[%.17g,%.17g],\"ele\":%.17g,"
               "\"x\":%d,\"y\":%d,\"gps\":[%.17g,%.17g,%.17g,%.17g],\"cmd\":\"%s\"}%s\n",
               t.x, t.y, t.z, r.x, r.y, r.z, r.w, origin.x, origin.y, origin.z, min_x, min_y, ele, x, y,
               gpsTl.x, gpsTl.y, gpsBr.x, gpsBr.y, cmd.str().c_str(), i + 1 < nm ? "," : "");
    }
    printf("]}\n");
    return 0;
}
