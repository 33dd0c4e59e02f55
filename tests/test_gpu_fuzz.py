"""Seeded fuzz of the whole feed path against the oracle: random camera intrinsics, heights, tilts up to the
0.4 obliqueness gate, yaw anywhere, frame sizes that are not multiples of anything, both pyramid types and the
single-band path.  Bit-exact like every other parity test; the point is to walk the wave-uniform fast paths of the
warp (tame coordinates, inside-the-frame taps, one-reflection borders) across their switch-over conditions."""
import math

import numpy as np
import pytest

from helpers import compare_maps, workloads

pytestmark = pytest.mark.gpu
FED = []


def random_case(rs, wl):
    w = int(rs.choice([320, 333, 480, 641, 800]))
    h = int(rs.choice([240, 257, 360, 479, 600]))
    f = float(rs.uniform(0.6, 1.6) * w)
    cam = [w, h, f, f * float(rs.uniform(0.9, 1.1)), w / 2 + float(rs.uniform(-20, 20)), h / 2 + float(rs.uniform(-20, 20))]
    height = float(rs.uniform(40, 160))
    n = int(rs.randint(2, 5))
    poses = []
    for k in range(n):
        yaw = rs.uniform(-math.pi, math.pi)
        tilt = math.radians(rs.uniform(0, 55))               # the gate rejects views more oblique than ~66 degrees
        axis = rs.uniform(-1, 1, 2); axis = axis / (np.linalg.norm(axis) + 1e-9)
        q = wl.quat_mul(wl.quat_axis((0, 0, 1), yaw), wl.quat_axis((axis[0], axis[1], 0), tilt))
        poses.append([float(k * rs.uniform(5, 40)), float(k * rs.uniform(-30, 30)), -height + float(rs.uniform(-5, 5))] + list(q))
    frames = [wl.noise_frame(h, w, int(rs.randint(1 << 20))) for _ in range(n)]
    return cam, poses, frames


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_multiband(pf, orc, seed):
    wl = workloads()
    rs = np.random.RandomState(1000 + seed)
    cam, poses, frames = random_case(rs, wl)
    ff = seed & 1
    bands = int(rs.choice([3, 5, 6]))
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff, band_number=bands)
    o = orc.OracleMap(band_num=bands, force_float=ff)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses[:2]) == o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    fed = 0
    for img, p in zip(frames, poses):
        a, b = g.feed(img, p), o.feed(img, p)
        assert a == b
        fed += bool(a)
    assert g.sync()
    FED.append(fed)
    if fed:
        assert g.grid() == o.grid()
        assert compare_maps(g, o) == []


def test_fuzz_rendered_something():
    """the cases above are not all rejections"""
    assert len(FED) >= 40 and sum(1 for f in FED if f >= 2) >= 20, FED


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_single_band(pf, orc, seed):
    wl = workloads()
    rs = np.random.RandomState(2000 + seed)
    cam, poses, frames = random_case(rs, wl)
    g = pf.Map2D.create(pf.TypeCPU, False)
    o = orc.OracleMap(single_band=True)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses[:2]) == o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    for img, p in zip(frames, poses):
        assert g.feed(img, p) == o.feed(img, p)
    assert g.sync()
    for t in o.tiles():
        assert np.array_equal(g.tile_bgra(*t), o.tile_bgra(*t)), t
