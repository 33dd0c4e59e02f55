"""Helpers shared by the parity tests."""
import hashlib
import importlib
import math

import numpy as np

from conftest import load_package


def workloads():
    load_package()
    return importlib.import_module("pi_slam_fusion_amd.workloads")


def jitter_poses(n, seed, step=(18.0, 7.0), height=100.0, yaw_deg=25.0, tilt_deg=6.0, below=True):
    """Small perspective workload: every frame has its own yaw and tilt, so the warp
    exercises bilinear taps at all 32x32 sub-pixel phases and the REFLECT border."""
    wl = workloads()
    rng = np.random.RandomState(seed)
    poses = []
    for k in range(n):
        yaw = math.radians(rng.uniform(-yaw_deg, yaw_deg))
        roll = math.radians(rng.uniform(-tilt_deg, tilt_deg))
        pitch = math.radians(rng.uniform(-tilt_deg, tilt_deg))
        q = wl.quat_mul(wl.quat_axis((0, 0, 1), yaw), wl.quat_mul(wl.quat_axis((0, 1, 0), pitch), wl.quat_axis((1, 0, 0), roll)))
        z = -height + rng.uniform(-3, 3)
        if not below:      # camera above the plane, looking along -z
            q = wl.quat_mul([1, 0, 0, 0], q); z = -z
        poses.append([k * step[0] + rng.uniform(-2, 2), k * step[1] + rng.uniform(-2, 2), z] + q)
    return poses


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def compare_maps(gpu, orc, exact=True, max_ulp=1):
    """Every tile, every level: Laplacian and weight.  Returns list of mismatch strings."""
    bad = []
    gt, ot = gpu.tiles(), orc.tiles()
    if gt != ot:
        return ["tile sets differ: gpu %d oracle %d" % (len(gt), len(ot))]
    for (ix, iy) in ot:
        for lv in range(orc.num_levels):
            gl, gw = gpu.tile_level(ix, iy, lv)
            ol, ow = orc.tile_level(ix, iy, lv)
            if not np.array_equal(gw, ow):
                bad.append("weight tile (%d,%d) level %d: %d px differ" % (ix, iy, lv, int((gw != ow).sum())))
            if exact or gl.dtype == np.int16:
                if not np.array_equal(gl, ol):
                    bad.append("lap tile (%d,%d) level %d: %d values differ, max |d| %g" %
                               (ix, iy, lv, int((gl != ol).sum()), float(np.abs(gl.astype(np.float64) - ol).max())))
            else:
                u = ulp_diff(gl, ol)
                if u.max() > max_ulp:
                    bad.append("lap tile (%d,%d) level %d: max %d ulp" % (ix, iy, lv, int(u.max())))
    return bad


def ulp_diff(a, b):
    """ULP distance between two float32 arrays (sign-magnitude ordering)."""
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def map_digest(m):
    """Order-independent digest of a whole map: {(ix,iy,level): (sha(lap), sha(w))}."""
    out = {}
    for (ix, iy) in m.tiles():
        for lv in range(m.num_levels):
            l, w = m.tile_level(ix, iy, lv)
            out["%d,%d,%d" % (ix, iy, lv)] = [sha(l), sha(w)]
    return out
