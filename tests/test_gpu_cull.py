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


# ------------------------------------------------------------------------------------------------------------------------------
# Round 5: the corners of the cull's argument the random generator above does not reach.  Every case is compared
# with the oracle, which renders every tile of every canvas.
def tilted_poses(wl, n, seed, height, tilt_lo_deg, tilt_hi_deg, step, yaw_deg=180.0):
    """keyframes tilted tilt_lo..tilt_hi degrees off nadir about a random horizontal axis, random yaw, a short walk over the ground"""
    import math
    rs = np.random.RandomState(4200 + seed)
    poses = []
    for k in range(n):
        th = math.radians(rs.uniform(tilt_lo_deg, tilt_hi_deg)); az = rs.uniform(0, 2 * math.pi)
        q = wl.quat_mul(wl.quat_axis((0, 0, 1), math.radians(rs.uniform(-yaw_deg, yaw_deg))), wl.quat_axis((math.cos(az), math.sin(az), 0.0), th))
        poses.append([k * step * math.cos(0.7 * k) + rs.uniform(-3, 3), k * step * math.sin(0.7 * k) + rs.uniform(-3, 3), -height] + q)
    return poses


def feed_both(pf, orc, poses, prep, seed, **opt):
    wl = workloads()
    o_opt = {"force_float": opt.get("force_float", 0), "scale": opt.get("scale", 1.0), "band_num": opt.get("band_number", 5), "weight_type": opt.get("weight_type", 0)}
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, fused=1, **opt)
    o = orc.OracleMap(**o_opt)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, prep) == o.prepare(wl.IDENTITY_PLANE, CAM, prep)
    frames = 0
    for k, p in enumerate(poses):
        img = wl.noise_frame(480, 640, 100 * seed + k) if k % 3 else wl.smooth_frame(480, 640, k)
        a, b = g.feed(img, p), o.feed(img, p)
        assert a == b, (seed, k, a, b)
        frames += bool(a)
    assert g.sync()
    return g, o, frames


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_tilt_up_to_the_obliqueness_gate(pf, orc, seed):
    """Corner rays up to the 0.4 gate (MultiBandMap2DCPU.cpp:336-343: 66 degrees off nadir; the camera's half diagonal is 38.7 degrees, so
    tilts of 18-27 degrees put a corner ray at 57-66 degrees and some keyframes beyond the gate, which both sides reject): M[6], M[7] are
    large, the footprint is a long trapezoid, and the weight's level sets are far from circles on the canvas."""
    wl = workloads()
    poses = tilted_poses(wl, 14, seed, 80.0, 18.0, 27.5, 9.0)
    g, o, frames = feed_both(pf, orc, poses, poses[:6], seed, force_float=seed & 1, weight_type=(seed >> 1) & 1, scale=1.5)
    miss = compare_maps(g, o)
    assert miss == [], miss[:4]
    assert frames >= 6 and g.culled_tiles() + g.culled_cells() > 0, (frames, g.culled_tiles(), g.culled_cells())
    g.close()


@pytest.mark.parametrize("force_float", [0, 1])
def test_weight_type_1_with_seven_bands(pf, orc, force_float):
    """WeightType 1 (squared radial weight, .cpp:407-414) with seven bands: 256-px dilation, single-pixel cells at level 6."""
    wl = workloads()
    poses = wl.serpentine(CAM, 90.0, 16, per_row=4, fwd_overlap=0.85, side_overlap=0.7, seed=77, yaw_jitter_deg=20.0, tilt_jitter_deg=6.0, max_rows=4)
    g, o, frames = feed_both(pf, orc, poses + poses[:3], poses[:8], 77, force_float=force_float, weight_type=1, band_number=7, scale=4.0)
    miss = compare_maps(g, o)
    assert miss == [], miss[:4]
    assert g.culled_tiles() + g.culled_cells() > 0
    g.close()


@pytest.mark.parametrize("seed", [5, 6])
def test_footprint_edges_across_the_dilated_cells_of_earlier_keyframes(pf, orc, seed):
    """Keyframes a few pixels to a few cells apart, at every yaw: each one's footprint edge cuts through the dilated cells (64 px + the
    pyramid's 64-px reach at five bands) of the ones before it, at every level's grid phase -- where 'wholly inside the earlier footprint'
    (the stored weights' lower bound) and 'nearest point of the new quadrilateral' (the new weights' upper bound) are closest to failing."""
    wl = workloads()
    rs = np.random.RandomState(700 + seed)
    scale = 2.0; px = 100.0 / 500.0 / scale                        # metres per canvas pixel (lengthPixel = H / f / Scale)
    poses, x, y = [], 0.0, 0.0
    for k in range(20):
        x += float(rs.choice([3, 17, 40, 64, 97, 130])) * px * float(rs.choice([-1, 1])); y += float(rs.choice([5, 23, 64, 111])) * px
        q = wl.quat_mul(wl.quat_axis((0, 0, 1), float(rs.uniform(-3.2, 3.2))), wl.quat_mul(wl.quat_axis((0, 1, 0), float(rs.uniform(-0.1, 0.1))), wl.quat_axis((1, 0, 0), float(rs.uniform(-0.1, 0.1)))))
        poses.append([x, y, -100.0] + q)
    g, o, frames = feed_both(pf, orc, poses, poses[:8], seed, force_float=seed & 1, scale=scale)
    miss = compare_maps(g, o)
    assert miss == [], miss[:4]
    assert frames == 20 and g.culled_cells() > 0
    g.close()


def test_revisit_after_spreadmap(pf, orc):
    """The grid grows (spreadMap, .cpp:561-604: new origin, tile offsets move) between two visits of the same ground: the tiles' lower
    bounds live with the tiles (stable coordinates), not with the dense index."""
    wl = workloads()
    poses = wl.serpentine(CAM, 100.0, 12, per_row=4, fwd_overlap=0.8, side_overlap=0.6, seed=5, yaw_jitter_deg=8.0, tilt_jitter_deg=3.0, max_rows=3)
    far = [[-420.0, -380.0, -100.0, 0, 0, 0, 1], [560.0, 610.0, -100.0, 0, 0, 0, 1]]        # well outside the prepared box, both directions
    seq = poses + far + poses[:6] + [far[0]] + poses[3:9]
    g, o, frames = feed_both(pf, orc, seq, poses[:8], 5, force_float=1, scale=2.0)
    assert g.grid()[0] == o.grid()[0] and g.grid()[0][2:] != [0, 0]            # the origin moved
    miss = compare_maps(g, o)
    assert miss == [], miss[:4]
    assert frames == len(seq) and g.culled_tiles() + g.culled_cells() > 0
    g.close()


def test_prepare_again_mid_sortie(pf, orc):
    """prepare() in the middle of a sortie (.cpp:266-286: a new data object; in-flight state is dropped): tiles and their lower bounds start
    over, the keyframes after it are culled against what THEY stored only."""
    wl = workloads()
    poses = wl.serpentine(CAM, 100.0, 18, per_row=6, fwd_overlap=0.85, side_overlap=0.7, seed=9, yaw_jitter_deg=10.0, tilt_jitter_deg=3.0, max_rows=3)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, fused=1, force_float=0, scale=2.0)
    o = orc.OracleMap(force_float=0, scale=2.0)
    for lo, hi in ((0, 10), (6, 18)):
        assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[lo:lo + 6]) and o.prepare(wl.IDENTITY_PLANE, CAM, poses[lo:lo + 6])
        for k in range(lo, hi):
            img = wl.noise_frame(480, 640, 900 + k)
            assert g.feed(img, poses[k]) == o.feed(img, poses[k])
    assert g.sync()
    miss = compare_maps(g, o)
    assert miss == [], miss[:4]
    assert g.culled_tiles() + g.culled_cells() > 0
    g.close()


def test_threaded_map_whose_queue_drops(pf, orc):
    """thread = 1 with a two-deep queue fed as fast as the host can (.cpp:298-304 drops the oldest): the lower bounds may rise for the
    keyframes that were RENDERED only.  pf_debug_render_log says which ones were; the oracle is fed exactly those, in that order."""
    wl = workloads()
    poses = wl.serpentine(CAM, 100.0, 60, per_row=6, fwd_overlap=0.9, side_overlap=0.7, seed=21, yaw_jitter_deg=10.0, tilt_jitter_deg=3.0, max_rows=4)
    frames = [wl.noise_frame(480, 640, 300 + k) for k in range(8)]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, True, fused=1, force_float=1, scale=2.0, max_queue=2)
    o = orc.OracleMap(force_float=1, scale=2.0)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) and o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    for k, p in enumerate(poses):
        assert g.feed(frames[k % 8], p)
    assert g.sync()
    st, log = g.stats(), g.render_log()
    assert st["rendered"] + st["dropped"] == len(poses) and len(log) == st["rendered"] and log == sorted(log)
    for k in log:
        assert o.feed(frames[k % 8], poses[k])
    miss = compare_maps(g, o)
    assert miss == [], (st, miss[:4])
    assert g.culled_tiles() + g.culled_cells() > 0
    g.close()
