// ref_ransac.cpp -- golden-vector generator for the tracker-side pieces of rows f1 / f3.
//
// TEST INFRASTRUCTURE ONLY.  Compiled (oracle/Makefile target `ref`) TOGETHER WITH the reference's own
// src/RANSAC.cpp, against its src/RANSAC.h, src/DataTrans.h and GSLAM/core/SE3.h where they lie under
// /root/reference; the binary goes to oracle/_ref/ and never into git.  tests/golden/make_ransac_vectors.py runs it
// and commits what it prints as tests/golden/ransac_vectors.json.
//
// What is pinned (the reference's code computes, this file only calls and prints):
//   * RANSAC::solve_plane (RANSAC.cpp:22-51): plane point, unit normal, the published (non-unit) quaternion;
//   * RANSAC::solve_distance (RANSAC.cpp:12-20);
//   * DataTrans<T> (src/DataTrans.h:40-83): capacity 30, product() drops the OLDEST element, FIFO consumption.
// RANSAC::ransac_core itself reseeds srand(time(nullptr)) inside its loop (RANSAC.cpp:71): its draws are not
// reproducible and are not pinned.  Both translation units are built with -Dprivate=public so that the two private
// helpers can be called; no reference source is modified or copied.
#include "RANSAC.h"
#include "DataTrans.h"
#include <cstdint>
#include <cstdio>

static uint64_t s_state = 20260311;
static uint64_t splitmix64()
{
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double urand(double a, double b) { return a + (b - a) * ((splitmix64() >> 11) * (1.0 / 9007199254740992.0)); }
static pi::Point3d rpoint(double r) { return pi::Point3d(urand(-r, r), urand(-r, r), urand(-r, r)); }
static void p3(const char* k, const pi::Point3d& p, const char* end) { printf("\"%s\":[%.17g,%.17g,%.17g]%s", k, p.x, p.y, p.z, end); }

int main()
{
    RANSAC& r = RANSAC::Instance();
    printf("{\"planes\":[\n");
    const int n = 48;
    for (int i = 0; i < n; i++) {
        // every third case nearly horizontal (the ground plane of a survey flight), the others arbitrary
        pi::Point3d A = rpoint(50), B = rpoint(50), C = rpoint(50);
        if (i % 3 == 0) { A.z = 2 + urand(-0.2, 0.2); B.z = 2 + urand(-0.2, 0.2); C.z = 2 + urand(-0.2, 0.2); }
        const pi::Point3d M = rpoint(80);
        r.solve_plane(A, B, C);
        printf("{"); p3("a", A, ","); p3("b", B, ","); p3("c", C, ","); p3("m", M, ",");
        p3("P", r.plane_P, ","); p3("N", r.plane_N, ",");
        printf("\"Q\":[%.17g,%.17g,%.17g,%.17g],", r.plane_Q.x, r.plane_Q.y, r.plane_Q.z, r.plane_Q.w);
        printf("\"dist\":%.17g}%s\n", RANSAC::solve_distance(M, r.plane_P, r.plane_N), i + 1 < n ? "," : "");
    }
    printf("],\n\"datatrans\":{");
    DataTrans<int>& q = DataTrans<int>::Instance();
    for (int k = 0; k < 35; k++) q.product(k);
    printf("\"produced\":35,\"consumed\":[");
    for (int k = 0; k < 30; k++) { int v = -1; q.consumption(v); printf("%d%s", v, k < 29 ? "," : ""); }
    printf("],\"max\":%d}}\n", q.m_maxSize);
    return 0;
}
