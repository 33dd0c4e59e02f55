"""The cull of FusionMap::render_frame (64x64 cells in which a keyframe cannot win the max-weight select at any level are not rendered)
against the oracle, which renders every tile of every canvas (MultiBandMap2DCPU.cpp:476-555): random sorties at 640x480 with
Map2D.Scale 1.5-4 (canvases of 6-12 tiles a side), heavy overlap, yaw up to 180 degrees, tilt up to 12, both weight types
(WeightType 0 / 1, .cpp:396-418), 3-7 bands, both pyramid types, and a revisit of the first keyframes at the end.
tools/cull_soak.py runs the same generator over many more seeds."""
import numpy as np
import pytest

from helpers import compare_maps, workloads

pytestmark = pytest.mark.gpu
CAM = [640, 480, 500, 500, 320, 240]


def run_case(pf, orc, seed):
    """-> (mismatches, frames rendered, culled tiles, culled cells, description)"""
    wl = workloads()
    rs = np.random.RandomState(9000 + seed)
    ff = seed & 1; wt = (seed >> 1) & 1
    bands = int(rs.choice([3, 4, 5, 5, 5, 6, 7])); scale = float(rs.choice([1.5, 2.0, 2.5, 3.0, 4.0]))
    yaw = float(rs.choice([3.0, 12.0, 45.0, 180.0])); tilt = float(rs.choice([0.5, 4.0, 12.0]))
    per_row = int(rs.randint(3, 8)); nfr = int(rs.randint(10, 22))
    poses = wl.serpentine(CAM, float(rs.uniform(60, 140)), nfr, per_row=per_row, fwd_overlap=float(rs.uniform(0.5, 0.9)),
                          side_overlap=float(rs.uniform(0.3, 0.8)), seed=seed, yaw_jitter_deg=yaw, tilt_jitter_deg=tilt, max_rows=3)
    poses = poses + [list(p) for p in poses[:3]]                  # fly over the start again
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff, fused=1, scale=scale, band_number=bands, weight_type=wt)
    o = orc.OracleMap(force_float=ff, scale=scale, band_num=bands, weight_type=wt)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    frames = 0
    for k, p in enumerate(poses):
        img = wl.noise_frame(480, 640, 100 * seed + k) if k % 3 else wl.smooth_frame(480, 640, k)
        a, b = g.feed(img, p), o.feed(img, p)
        assert a == b, (seed, k, a, b)
        frames += bool(a)
    assert g.sync()
    miss = compare_maps(g, o)
    res = (miss, frames, g.culled_tiles(), g.culled_cells(),
           "ff=%d wt=%d bands=%d scale=%.1f yaw=%g tilt=%g frames=%d" % (ff, wt, bands, scale, yaw, tilt, len(poses)))
    g.close()
    return res


@pytest.mark.parametrize("seed", [1, 2, 3, 10, 36, 57])
def test_random_sortie_with_cull_equals_oracle(pf, orc, seed):
    miss, frames, tiles, cells, what = run_case(pf, orc, seed)
    assert miss == [], (what, miss[:4])
    assert frames >= 10 and tiles + cells > 0, (what, frames, tiles, cells)
