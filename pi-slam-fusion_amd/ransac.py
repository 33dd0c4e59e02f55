"""Ground-plane producer of the tracker side: the reference's RANSAC singleton
(src/RANSAC.h:12-58, src/RANSAC.cpp) collects SLAM map points, fits a plane once 2000 points
have arrived and publishes it as an SE3 (rotation that turns the plane normal onto +-z, any
point of the plane as translation) on the `Trans_Plane` queue, from where `Map2D::prepare`
takes its `plane` argument (RANSAC.cpp:115, Map2DFusion.cpp:189-205).

Restated here so a GSLAM-style producer can be wired to `Map2D.prepare` unchanged.  Deliberate
differences from the reference, both about determinism of a test fixture, not about the fit:
  * the sample indices come from a caller-supplied `random.Random` (the reference reseeds
    `srand(time(nullptr))` inside the loop, RANSAC.cpp:71, so all draws within one second repeat);
  * the adaptive iteration count uses the inlier *ratio* as a real number and only ever shrinks the budget
    (RANSAC.cpp:96 divides two integers, which is 0 until every point is an inlier and makes the
    logarithm's argument 1).
"""
import math
import random

from .datatrans import DataTrans

SIGMA = 0.15          # inlier distance, RANSAC.cpp:62
CONFIDENCE = 0.999    # RANSAC.cpp:66
MAX_ITERS = 10000     # RANSAC.cpp:60
MIN_POINTS = 2000     # RANSAC.cpp:112
REF_PI = 3.1415926535  # the constant RANSAC.cpp:25 folds the angle with (not M_PI): kept, so that the published
                       # quaternion equals the reference's to the last bit (tests/golden/ransac_vectors.json)


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def point_plane_distance(m, p, n):
    """|n . (m - p)| / |n|  (RANSAC.cpp:12-20 writes it as Ax+By+Cz+D with D = -n.p)."""
    d = -_dot(n, p)
    return abs(_dot(n, m) + d) / math.sqrt(_dot(n, n))


def plane_from_points(a, b, c):
    """Plane through three points as (point, unit normal, quaternion xyzw) -- RANSAC.cpp:22-51.
    normal = (b - c) x (b - a) normalised; the quaternion is the axis-angle rotation about
    normal x z by acos(normal . z), folded so that the angle never exceeds pi/2 (a plane has
    two normals); the plane's point is b."""
    n = _cross(_sub(b, c), _sub(b, a))
    ln = math.sqrt(_dot(n, n))
    if ln == 0.0:
        return None
    inv = 1.0 / ln                                        # Point3_::normalize multiplies by 1./norm() (Point.h:158-164)
    n = (n[0] * inv, n[1] * inv, n[2] * inv)
    z = (0.0, 0.0, 1.0)
    axis = _cross(n, z)
    angle = math.acos(max(-1.0, min(1.0, _dot(n, z))))    # (a unit normal's z never leaves [-1, 1] by more than rounding)
    if angle > REF_PI / 2.0:
        angle = REF_PI - angle
        axis = (axis[0] * -1, axis[1] * -1, axis[2] * -1)
    s = math.sin(angle / 2.0)
    # the reference scales the UNnormalised axis (|axis| = sin(angle)) by sin(angle/2), RANSAC.cpp:46-48;
    # kept as is -- the result is a non-unit quaternion whose direction is the rotation's axis
    q = (axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle / 2.0))
    return tuple(b), n, q


def fit(points, rng=None, sigma=SIGMA, confidence=CONFIDENCE, max_iters=MAX_ITERS):
    """RANSAC.cpp:53-106.  Returns (point, normal, quaternion, inliers) of the last model drawn when the
    loop ends: like the reference, the loop stops at the first model that more than half of the points
    agree with, otherwise after the adaptive number of draws."""
    rng = rng or random.Random(0)
    size = len(points)
    if size < 3:
        return None
    iters, best, model, i = max_iters, 0, None, 0
    while i < iters:
        ia, ib, ic = rng.randrange(size), rng.randrange(size), rng.randrange(size)
        if ia == ib or ib == ic or ic == ia:              # RANSAC.cpp:76-80: draw again, not counted
            continue
        i += 1
        m = plane_from_points(points[ia], points[ib], points[ic])
        if m is None:
            continue
        inliers = sum(1 for p in points if point_plane_distance(p, m[0], m[1]) < sigma)
        model = m + (inliers,)
        if inliers > best:
            best = inliers
            w = inliers / float(size)
            if 0.0 < w < 1.0:
                iters = min(iters, int(math.log(1.0 - confidence) / math.log(1.0 - w * w)) + 1)
        if inliers > size // 2:                           # RANSAC.cpp:101-102
            break
    return model


class Ransac:
    """`ransac.solve(point)` collector (RANSAC.cpp:108-121): publishes [x y z qx qy qz qw] -- the SE3 stream
    order of SE3.h:112-117 -- on `trans_plane` once `min_points` points are in."""

    def __init__(self, trans_plane=None, min_points=MIN_POINTS, rng=None):
        self.points = []
        self.finished = False
        self.trans_plane = trans_plane if trans_plane is not None else DataTrans()
        self.min_points = min_points
        self.rng = rng or random.Random(0)
        self.model = None

    def solve(self, point):
        self.points.append(tuple(point))
        if len(self.points) < self.min_points:
            return
        self.model = fit(self.points, self.rng)
        if self.model is None:
            return
        self.finished = True
        p, _, q, _ = self.model
        self.trans_plane.product([p[0], p[1], p[2], q[0], q[1], q[2], q[3]])

    def is_finished(self):
        return self.finished
