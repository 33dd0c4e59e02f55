"""Ground-plane producer: restatement of src/RANSAC.cpp, host logic only.  The plane fit of three points,
the point-plane distance and the queue semantics are pinned by vectors from the reference's own RANSAC.cpp / DataTrans.h
compiled here (oracle/ref_ransac.cpp -> oracle/_ref/ransac_ref -> tests/golden/ransac_vectors.json)."""
import importlib
import json
import math
import os
import random

import numpy as np

from conftest import load_package


def mod():
    load_package()
    return importlib.import_module("pi_slam_fusion_amd.ransac")


def rotate(q, v):
    """unit quaternion xyzw applied to v"""
    x, y, z, w = q
    u = np.array([x, y, z]); v = np.asarray(v, float)
    return v + 2 * np.cross(u, np.cross(u, v) + w * v)


def test_plane_from_three_points():
    r = mod()
    a, b, c = (0, 0, 1.0), (1, 0, 1.5), (0, 1, 0.5)
    p, n, q = r.plane_from_points(a, b, c)
    assert p == b                                           # RANSAC.cpp:43
    assert abs(np.linalg.norm(n) - 1) < 1e-12
    for pt in (a, b, c):
        assert r.point_plane_distance(pt, p, n) < 1e-12
    # (b - c) x (b - a), RANSAC.cpp:29
    ref = np.cross(np.subtract(b, c), np.subtract(b, a)); ref /= np.linalg.norm(ref)
    assert np.allclose(n, ref)
    # the published quaternion is not normalised (axis of length sin(angle) times sin(angle/2)); its direction is
    # the rotation axis normal x z and its scalar part cos(angle/2) of the folded angle
    angle = math.acos(abs(ref[2]))
    assert abs(q[3] - math.cos(angle / 2)) < 1e-12
    axis = np.cross(ref, (0, 0, 1.0)) * (1 if ref[2] >= 0 else -1)
    assert np.allclose(q[:3], axis * math.sin(angle / 2))
    # with the axis normalised it is the rotation that turns the normal onto +-z
    qa = np.array(q[:3]); qa = qa / np.linalg.norm(qa) * math.sin(angle / 2)
    z = rotate((qa[0], qa[1], qa[2], q[3]), ref)
    assert abs(abs(z[2]) - 1) < 1e-9
    assert r.plane_from_points(a, a, c) is None             # degenerate triple


def test_fit_recovers_plane_with_outliers():
    r = mod()
    rs = np.random.RandomState(3)
    n_true = np.array([0.1, -0.2, 1.0]); n_true /= np.linalg.norm(n_true)
    pts = []
    for _ in range(1500):
        x, y = rs.uniform(-50, 50, 2)
        z = (5.0 - n_true[0] * x - n_true[1] * y) / n_true[2] + rs.normal(0, 0.03)
        pts.append((x, y, z))
    for _ in range(500):
        pts.append(tuple(rs.uniform(-50, 50, 3)))
    p, n, q, inl = r.fit(pts, random.Random(7))
    assert inl > len(pts) // 2                              # stops at the first majority model, RANSAC.cpp:101
    assert abs(abs(np.dot(n, n_true)) - 1) < 1e-3
    assert r.fit(pts[:2]) is None


def test_collector_publishes_on_trans_plane():
    r = mod()
    dt = importlib.import_module("pi_slam_fusion_amd.datatrans")
    q = dt.DataTrans()
    col = r.Ransac(q, min_points=60, rng=random.Random(1))
    rs = np.random.RandomState(0)
    for k in range(60):
        assert not col.is_finished() and q.size() == 0
        x, y = rs.uniform(-10, 10, 2)
        col.solve((x, y, 2.0 + 0.01 * rs.normal()))
    assert col.is_finished() and q.size() == 1
    se3 = q.consumption(timeout=1)
    assert len(se3) == 7                                    # x y z qx qy qz qw, SE3.h:112-117
    assert abs(se3[2] - 2.0) < 0.1 and abs(se3[6] - 1.0) < 1e-3     # near-horizontal plane: almost no rotation
    col.solve((0.0, 0.0, 2.0))                              # every further point refits and republishes (RANSAC.cpp:112-120)
    assert q.size() == 1


def test_plane_fit_and_queue_equal_the_reference_compiled_here():
    """48 triples through the reference's RANSAC::solve_plane / solve_distance and 35 products through its DataTrans<int>:
    plane point, unit normal, the published quaternion (folded with the reference's truncated pi constant) and the distance
    of a probe point are reproduced bit for bit; the queue keeps the newest 30 in order."""
    r = mod()
    dt = importlib.import_module("pi_slam_fusion_amd.datatrans")
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ransac_vectors.json")))
    assert len(vec["planes"]) == 48
    for c in vec["planes"]:
        p, n, q = r.plane_from_points(tuple(c["a"]), tuple(c["b"]), tuple(c["c"]))
        assert list(p) == c["P"] and list(n) == c["N"]
        assert [x + 0.0 for x in q] == [x + 0.0 for x in c["Q"]]          # (-0.0 == 0.0)
        assert r.point_plane_distance(tuple(c["m"]), p, n) == c["dist"]
    d = vec["datatrans"]
    q = dt.DataTrans()
    for k in range(d["produced"]):
        q.product(k)
    assert q.size() == d["max"] == 30
    assert [q.consumption(timeout=1) for _ in range(30)] == d["consumed"]
