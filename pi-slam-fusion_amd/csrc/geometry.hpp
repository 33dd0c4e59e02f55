// geometry.hpp -- host-side fp64 geometry of the fusion path.
//
// The operation ORDER of every expression follows the reference so that tile
// ranges, canvases and homographies are bit-identical to what the reference
// computes on the same inputs (compiled with -ffp-contract=off):
//   SE3 / SO3      GSLAM/GSLAM/core/SE3.h:70-101, SO3.h:435-450,481-484
//   footprint      Map2DFusion/MultiBandMap2DCPU.cpp:324-347
//   homography     cv::getPerspectiveTransform set-up (call site .cpp:441);
//                  solved by partial-pivot Gaussian elimination
//   3x3 inverse    cv::invert closed form used by cv::warpPerspective
#pragma once
#include <cmath>
#include <cstring>

namespace pf {

struct Pose {                 // x y z qx qy qz qw  (SE3.h:112-117 stream order)
    double t[3];
    double q[4];
};

inline Pose pose_from7(const double p[7]) { Pose r; std::memcpy(r.t, p, 24); std::memcpy(r.q, p + 3, 32); return r; }

inline void quat_mul(const double a[4], const double b[4], double o[4])
{
    const double x = a[0], y = a[1], z = a[2], w = a[3];
    const double r0 = w * b[0] + x * b[3] + y * b[2] - z * b[1];
    const double r1 = w * b[1] + y * b[3] + z * b[0] - x * b[2];
    const double r2 = w * b[2] + z * b[3] + x * b[1] - y * b[0];
    const double r3 = w * b[3] - x * b[0] - y * b[1] - z * b[2];
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}

// SO3 * Point3: the quaternion sandwich (q * (p,0)) * q^-1
inline void rotate(const double q[4], const double p[3], double out[3])
{
    const double pq[4] = { p[0], p[1], p[2], 0 }, qi[4] = { -q[0], -q[1], -q[2], q[3] };
    double t[4], r[4];
    quat_mul(q, pq, t);
    quat_mul(t, qi, r);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

inline Pose inverse(const Pose& a)
{
    Pose r;
    r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
    double t[3];
    rotate(r.q, a.t, t);
    r.t[0] = -t[0]; r.t[1] = -t[1]; r.t[2] = -t[2];
    return r;
}

inline Pose mul(const Pose& a, const Pose& b)
{
    Pose r;
    quat_mul(a.q, b.q, r.q);
    double t[3];
    rotate(a.q, b.t, t);
    r.t[0] = a.t[0] + t[0]; r.t[1] = a.t[1] + t[1]; r.t[2] = a.t[2] + t[2];
    return r;
}

struct Camera { double w, h, fx, fy, cx, cy, fxinv, fyinv; };

// four image corners -> ground (z=0 plane) points; false if any ray fails the
// 0.4 down-look gate
inline bool footprint(const Camera& c, const Pose& pose, double pts[8])
{
    const double img[8] = { 0, 0, c.w, 0, 0, c.h, c.w, c.h };
    double down[3] = { 0, 0, -1 };
    if (pose.t[2] < 0) down[2] = 1;
    for (int i = 0; i < 4; i++) {
        const double p[3] = { (img[2 * i] - c.cx) * c.fxinv, (img[2 * i + 1] - c.cy) * c.fyinv, 1. };
        double axis[3];
        rotate(pose.q, p, axis);
        if (axis[0] * down[0] + axis[1] * down[1] + axis[2] * down[2] < 0.4) return false;
        const double s = pose.t[2] / axis[2];
        pts[2 * i]     = pose.t[0] - s * axis[0];
        pts[2 * i + 1] = pose.t[1] - s * axis[1];
    }
    return true;
}

inline void perspective_transform(const float src[8], const float dst[8], double M[9])
{
    double a[8][9];
    for (int i = 0; i < 4; i++) {
        const float sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        double* r0 = a[i]; double* r1 = a[i + 4];
        r0[0] = r1[3] = sx; r0[1] = r1[4] = sy; r0[2] = r1[5] = 1;
        r0[3] = r0[4] = r0[5] = r1[0] = r1[1] = r1[2] = 0;
        r0[6] = (double)(-sx * dx); r0[7] = (double)(-sy * dx);       // float products, as Point2f
        r1[6] = (double)(-sx * dy); r1[7] = (double)(-sy * dy);
        r0[8] = dx; r1[8] = dy;
    }
    for (int c = 0; c < 8; c++) {
        int piv = c; double best = std::fabs(a[c][c]);
        for (int r = c + 1; r < 8; r++) { const double v = std::fabs(a[r][c]); if (v > best) { best = v; piv = r; } }
        if (piv != c) for (int k = 0; k < 9; k++) { const double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
        if (a[c][c] == 0) continue;
        for (int r = c + 1; r < 8; r++) {
            const double f = a[r][c] / a[c][c];
            if (f == 0) continue;
            for (int k = c; k < 9; k++) a[r][k] = a[r][k] - f * a[c][k];
        }
    }
    double x[8];
    for (int r = 7; r >= 0; r--) {
        double s = a[r][8];
        for (int k = r + 1; k < 8; k++) s = s - a[r][k] * x[k];
        x[r] = a[r][r] != 0 ? s / a[r][r] : 0;
    }
    for (int i = 0; i < 8; i++) M[i] = x[i];
    M[8] = 1.;
}

inline bool invert3x3(const double S[9], double D[9])
{
    auto m = [&](int r, int c) { return S[r * 3 + c]; };
    double d = m(0,0) * (m(1,1) * m(2,2) - m(1,2) * m(2,1))
             - m(0,1) * (m(1,0) * m(2,2) - m(1,2) * m(2,0))
             + m(0,2) * (m(1,0) * m(2,1) - m(1,1) * m(2,0));
    if (d == 0.) return false;
    d = 1. / d;
    double t[9];
    t[0] = (m(1,1) * m(2,2) - m(1,2) * m(2,1)) * d;
    t[1] = (m(0,2) * m(2,1) - m(0,1) * m(2,2)) * d;
    t[2] = (m(0,1) * m(1,2) - m(0,2) * m(1,1)) * d;
    t[3] = (m(1,2) * m(2,0) - m(1,0) * m(2,2)) * d;
    t[4] = (m(0,0) * m(2,2) - m(0,2) * m(2,0)) * d;
    t[5] = (m(0,2) * m(1,0) - m(0,0) * m(1,2)) * d;
    t[6] = (m(1,0) * m(2,1) - m(1,1) * m(2,0)) * d;
    t[7] = (m(0,1) * m(2,0) - m(0,0) * m(2,1)) * d;
    t[8] = (m(0,0) * m(1,1) - m(0,1) * m(1,0)) * d;
    std::memcpy(D, t, sizeof(t));
    return true;
}

}  // namespace pf
