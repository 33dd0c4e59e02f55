// ref_se3.cpp -- golden-vector generator for the SE3 / footprint geometry.
//
// TEST INFRASTRUCTURE ONLY.  Compiled (oracle/Makefile target `ref`) against the
// reference's own header-only geometry where it lies under /root/reference
// (GSLAM/GSLAM/core/SE3.h, SO3.h, Point.h); the binary goes to oracle/_ref/ and
// never into git.  tests/golden/make_se3_vectors.py runs it and commits the
// JSON it prints as tests/golden/se3_vectors.json.
//
// What is pinned: the image type codes of the boundary (GImage.h:97-101); plane.inverse()*pose (Map2D.cpp:45, MultiBandMap2DCPU.cpp:297),
// SO3*Point3d (the quaternion sandwich) and the four-corner ground footprint
// with the 0.4 obliqueness gate (MultiBandMap2DCPU.cpp:324-347).
#include <GSLAM/core/SE3.h>
#include <GSLAM/core/GImage.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

static uint64_t s_state;
static uint64_t splitmix64() {
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double urand(double a, double b) { return a + (b - a) * ((splitmix64() >> 11) * (1.0 / 9007199254740992.0)); }

static pi::SE3d randomPose(double tilt_deg, double h_sign) {
    // small tilt about a random horizontal axis, random yaw, height 50..150
    double yaw = urand(-M_PI, M_PI), tilt = urand(0, tilt_deg * M_PI / 180.), dir = urand(-M_PI, M_PI);
    pi::SO3d ryaw  = pi::SO3d::FromAxis(pi::Point3d(0, 0, 1), yaw);
    pi::SO3d rtilt = pi::SO3d::FromAxis(pi::Point3d(cos(dir), sin(dir), 0), tilt);
    pi::SO3d r = ryaw * rtilt;
    if (h_sign > 0) r = pi::SO3d(1, 0, 0, 0) * r;          // look along -z from above
    return pi::SE3d(r, pi::Point3d(urand(-200, 200), urand(-200, 200), h_sign * urand(50, 150)));
}

static void printSE3(const char* key, const pi::SE3d& p, bool comma = true) {
    const pi::Point3d& t = p.get_translation(); const pi::SO3d& r = p.get_rotation();
    printf("\"%s\":[%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g]%s", key, t.x, t.y, t.z, r.x, r.y, r.z, r.w, comma ? "," : "");
}

int main() {
    s_state = 20240607;
    const double cam[6] = { 4000, 3000, 3000, 3000, 2000, 1500 };
    const double fxinv = 1. / cam[2], fyinv = 1. / cam[3];
    printf("{\"cam\":[%.17g,%.17g,%.17g,%.17g,%.17g,%.17g],\n\"cases\":[\n", cam[0], cam[1], cam[2], cam[3], cam[4], cam[5]);
    const int N = 64;
    for (int c = 0; c < N; c++) {
        double hs = (c & 1) ? 1. : -1.;
        pi::SE3d plane = (c % 4 == 0) ? pi::SE3d()
                       : pi::SE3d(pi::SO3d::FromAxis(pi::Point3d(urand(-1, 1), urand(-1, 1), urand(-1, 1)), urand(-0.2, 0.2)),
                                  pi::Point3d(urand(-5, 5), urand(-5, 5), urand(-5, 5)));
        pi::SE3d pose = randomPose((c % 8 == 7) ? 75. : 12., hs);   // every 8th: oblique, may fail the gate
        pi::SE3d world = plane * pose;                             // camera-to-world handed to feed()
        pi::SE3d local = plane.inverse() * world;                  // MultiBandMap2DCPU.cpp:297
        pi::Point3d probe(urand(-1, 1), urand(-1, 1), 1.);
        pi::Point3d rot = local.get_rotation() * probe;
        // MultiBandMap2DCPU.cpp:324-347
        double img[8] = { 0, 0, cam[0], 0, 0, cam[1], cam[0], cam[1] };
        pi::Point3d downLook(0, 0, -1);
        if (local.get_translation().z < 0) downLook = pi::Point3d(0, 0, 1);
        bool ok = true; double pts[8] = { 0 };
        for (int i = 0; i < 4 && ok; i++) {
            pi::Point3d axis = local.get_rotation() * pi::Point3d((img[2 * i] - cam[4]) * fxinv, (img[2 * i + 1] - cam[5]) * fyinv, 1.);
            if (axis.dot(downLook) < 0.4) { ok = false; break; }
            axis = local.get_translation() - axis * (local.get_translation().z / axis.z);
            pts[2 * i] = axis.x; pts[2 * i + 1] = axis.y;
        }
        printf("{");
        printSE3("plane", plane); printSE3("world", world); printSE3("local", local);
        printSE3("plane_inv", plane.inverse());
        printf("\"probe\":[%.17g,%.17g,%.17g],\"rot\":[%.17g,%.17g,%.17g],", probe.x, probe.y, probe.z, rot.x, rot.y, rot.z);
        printf("\"ok\":%d,\"pts\":[%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g]}%s\n", ok ? 1 : 0,
               pts[0], pts[1], pts[2], pts[3], pts[4], pts[5], pts[6], pts[7], c + 1 < N ? "," : "");
    }
    // image type codes of the boundary (pf_image.type == cv::Mat::type() == GImage::type(), GImage.h:97-101): the
    // reference's own GImageType<element, channels>::Type for the types Map2D::feed and the Ele pyramids use
    printf("],\n\"gimage_types\":{\"8UC1\":%d,\"8UC3\":%d,\"8UC4\":%d,\"16SC3\":%d,\"32FC1\":%d,\"32FC3\":%d}}\n",
           (int)GSLAM::GImageType<uint8_t, 1>::Type, (int)GSLAM::GImageType<uint8_t, 3>::Type, (int)GSLAM::GImageType<uint8_t, 4>::Type,
           (int)GSLAM::GImageType<int16_t, 3>::Type, (int)GSLAM::GImageType<float, 1>::Type, (int)GSLAM::GImageType<float, 3>::Type);
    return 0;
}
