"""The LOOKAHEAD of the cull (pf_options.lookahead, round 6): a keyframe waits, fed but not rendered, until `lookahead` more keyframes are in, and
is then left out of the cells in which one of THOSE is bound to overwrite it (FusionMap::render_frame / render_front).  The reference renders
every keyframe inside its own feed call (MultiBandMap2DCPU.cpp:288-309, :311-558); the max-weight select (.cpp:521, :542: `>=`) ends with the
largest weight, the newest keyframe among equals, whatever the order -- so the map a caller can observe after feed k must be the reference's
after keyframes 1..k, for every lookahead.  Checked here against the oracle, which renders every tile of every keyframe in feed order."""
import numpy as np
import pytest

from helpers import compare_maps, map_digest, workloads

pytestmark = pytest.mark.gpu
CAM = [640, 480, 500, 500, 320, 240]


def sortie(wl, seed, n=18, scale=2.0, **kw):
    rs = np.random.RandomState(5100 + seed)
    return wl.serpentine(CAM, float(rs.uniform(70, 130)), n, per_row=int(rs.randint(3, 7)), fwd_overlap=float(rs.uniform(0.6, 0.9)),
                         side_overlap=float(rs.uniform(0.4, 0.8)), seed=seed, yaw_jitter_deg=kw.get("yaw", 10.0), tilt_jitter_deg=kw.get("tilt", 3.0), max_rows=3)


def frame(wl, seed, k):
    return wl.noise_frame(480, 640, 100 * seed + k) if k % 3 else wl.smooth_frame(480, 640, k)


@pytest.mark.parametrize("lookahead", [0, 1, 2, 4, 9, 40])
@pytest.mark.parametrize("seed", [11, 12])
def test_every_lookahead_equals_the_oracle(pf, orc, seed, lookahead):
    wl = workloads()
    ff = seed & 1
    poses = sortie(wl, seed)
    poses = poses + [list(p) for p in poses[:4]]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff, scale=2.0, lookahead=lookahead)
    o = orc.OracleMap(force_float=ff, scale=2.0)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    for k, p in enumerate(poses):
        img = frame(wl, seed, k)
        assert g.feed(img, p) == o.feed(img, p)
    assert g.sync()
    assert compare_maps(g, o) == []
    assert g.stats()["rendered"] == len(poses)
    g.close()


def test_lookahead_culls_more_and_leaves_the_same_map(pf):
    """the same sortie with lookahead 0 and 6: identical tiles, identical Ischanged set, more cells left out"""
    wl = workloads()
    poses = sortie(wl, 3, n=24, yaw=5.0, tilt=2.0)
    out = []
    for la in (0, 6):
        g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, scale=2.0, lookahead=la)
        assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
        for k, p in enumerate(poses):
            assert g.feed(frame(wl, 3, k), p)
        dig, cul, st, tl = map_digest(g), g.culled_tiles() * 16 + g.culled_cells(), g.stats(), sorted(g.tiles())
        out.append((dig, cul, st, tl, sorted(g.blend_changed()[0])))          # Ischanged of every tile a canvas held (.cpp:553)
        g.close()
    assert out[0][0] == out[1][0]
    assert out[0][2] == out[1][2] and out[0][3] == out[1][3] and out[0][4] == out[1][4]
    assert out[1][1] > out[0][1], (out[0][1], out[1][1])


@pytest.mark.parametrize("force_float", [0, 1])
def test_looking_at_the_map_between_feeds(pf, orc, force_float):
    """a reader between two feeds (tile access, tile list, statistics, blend) sees the map after the keyframes fed so far"""
    wl = workloads()
    poses = sortie(wl, 21, n=16)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, scale=2.0, lookahead=5)
    o = orc.OracleMap(force_float=force_float, scale=2.0)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    for k, p in enumerate(poses):
        img = frame(wl, 21, k)
        assert g.feed(img, p) == o.feed(img, p)
        if k % 4 == 1:
            assert g.stats()["rendered"] == k + 1                  # no sync: the reader itself renders what waits
            assert sorted(g.tiles()) == sorted(o.tiles())
        if k % 5 == 2:
            assert compare_maps(g, o) == [], k
        if k == 9:
            ix, iy = sorted(o.tiles())[len(o.tiles()) // 2]
            assert np.array_equal(g.blend_tile(ix, iy), o.blend_tile(ix, iy))
    assert g.sync() and compare_maps(g, o) == []
    g.close()


def test_rejected_and_geometry_only_feeds_among_waiting_keyframes(pf, orc):
    """an oblique keyframe is refused inside its own feed call (.cpp:336-343) and a keyframe far outside the grid moves it at once (spreadMap,
    .cpp:561-604), while accepted keyframes before them still wait"""
    wl = workloads()
    import math
    poses = sortie(wl, 31, n=14)
    steep = list(poses[5]); steep[3:] = wl.quat_axis((1, 0, 0), math.radians(75.0))
    far = [poses[0][0] - 900.0, poses[0][1] - 700.0, poses[0][2], 0, 0, 0, 1]
    seq = [(p, True) for p in poses[:6]] + [(steep, True), (far, True)] + [(p, True) for p in poses[6:]] + [(far, True)] + [(p, True) for p in poses[2:5]]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=0, scale=2.0, lookahead=4)
    o = orc.OracleMap(force_float=0, scale=2.0)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    for k, (p, pixels) in enumerate(seq):
        img = frame(wl, 31, k) if pixels else None
        a, b = g.feed(img, p), o.feed(img, p)
        assert a == b, (k, a, b)
        assert g.grid()[0] == o.grid()[0], k                      # the grid advances inside the feed call (spreadMap)
    assert g.sync() and compare_maps(g, o) == []
    assert g.stats()["rejected"] == 1
    g.close()


def test_prepare_again_with_keyframes_waiting(pf, orc):
    wl = workloads()
    poses = sortie(wl, 41, n=12)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, scale=2.0, lookahead=8)
    o = orc.OracleMap(force_float=1, scale=2.0)
    for rnd in range(2):
        assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
        for k, p in enumerate(poses[: 7 + 5 * rnd]):
            img = frame(wl, 41 + rnd, k)
            assert g.feed(img, p) == o.feed(img, p)
    assert g.sync() and compare_maps(g, o) == []
    g.close()


@pytest.mark.parametrize("lookahead", [0, 4])
def test_threaded_map_with_lookahead(pf, orc, lookahead):
    """thread = 1: the render thread takes what the queue holds as company for the keyframe it renders and never waits for more"""
    wl = workloads()
    poses = sortie(wl, 51, n=16)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, True, force_float=0, scale=2.0, lookahead=lookahead)
    o = orc.OracleMap(force_float=0, scale=2.0)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    imgs = [frame(wl, 51, k) for k in range(len(poses))]
    for k, p in enumerate(poses):
        assert g.feed(imgs[k], p)
    assert g.sync()
    log = g.render_log()
    assert log == sorted(log) and len(log) == g.stats()["rendered"]
    for k in log:                                                   # the keyframes the queue did not drop, in order
        assert o.feed(imgs[k], poses[k])
    assert compare_maps(g, o) == []
    g.close()


def test_device_frames_wait_in_place(pf, orc):
    """pf_feed_device: the caller's buffers are read when the keyframe is rendered -- up to `lookahead` feeds later, at the latest in pf_sync"""
    import torch
    wl = workloads()
    poses = sortie(wl, 61, n=12)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, scale=2.0, lookahead=3)
    o = orc.OracleMap(force_float=1, scale=2.0)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    keep = []
    for k, p in enumerate(poses):
        img = frame(wl, 61, k)
        t = torch.from_numpy(img).cuda(); keep.append(t)
        torch.cuda.synchronize()
        assert g.feed_device(t.data_ptr(), 480, 640, p) == o.feed(img, p)
    assert g.sync() and compare_maps(g, o) == []
    g.close()


def test_seven_bands_weight_type_1_with_lookahead(pf, orc):
    wl = workloads()
    poses = wl.serpentine(CAM, 90.0, 16, per_row=4, fwd_overlap=0.85, side_overlap=0.7, seed=78, yaw_jitter_deg=20.0, tilt_jitter_deg=6.0, max_rows=4)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=0, scale=4.0, band_number=7, weight_type=1, lookahead=6)
    o = orc.OracleMap(force_float=0, scale=4.0, band_num=7, weight_type=1)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, poses[:8])
    for k, p in enumerate(poses + poses[:3]):
        img = frame(wl, 78, k)
        assert g.feed(img, p) == o.feed(img, p)
    assert g.sync() and compare_maps(g, o) == []
    assert g.culled_tiles() + g.culled_cells() > 0
    g.close()


def test_full_size_sortie_with_a_deep_window_against_the_oracle(pf, orc):
    """bench.py's path at its own size: thirty device-resident 4000 x 3000 keyframes over three flight lines, the default lookahead (48: every keyframe of a
    line waits until the next line's are in), a reader in the middle of the second line; every tile level, two blends and the save against the oracle."""
    torch = pytest.importorskip("torch")
    wl = workloads()
    cam = [4000, 3000, 3000, 3000, 2000, 1500]
    poses = wl.serpentine(cam, 100.0, 30, per_row=10)
    host = [wl.noise_frame(3000, 4000, 160 + k) if k else wl.smooth_frame(3000, 4000, 3) for k in range(3)]
    dev = [torch.from_numpy(f).cuda() for f in host]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1)
    o = orc.OracleMap(force_float=1)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        assert g.feed_device(dev[k % 3].data_ptr(), 3000, 4000, p) and o.feed(host[k % 3], p)
        if k == 14:
            assert g.stats()["rendered"] == 15 and sorted(g.tiles()) == sorted(o.tiles())
    assert g.sync() and g.grid() == o.grid()
    assert compare_maps(g, o) == []
    tiles = o.tiles()
    have = set(tiles)
    inner = [t for t in tiles if all((t[0] + dx, t[1] + dy) in have for dx in (-1, 0, 1) for dy in (-1, 0, 1))]
    assert len(tiles) > 700 and g.culled_tiles() > 2000
    for t in [inner[len(inner) // 4], inner[3 * len(inner) // 4]]:
        assert np.array_equal(g.blend_tile(*t), o.blend_tile(*t)), t
    assert np.array_equal(g.save_to_memory()[0], o.save()[0])
    g.close()


@pytest.mark.parametrize("force_float", [0, 1])
def test_equal_weights_the_newest_keyframe_wins(pf, orc, force_float):
    """every pose twice in a row, and the first three again at the end, each time with other pixels: the two keyframes' weights are EQUAL everywhere, `>=`
    (.cpp:521, :542) lets the newer one win, and no bound of one may cull the other (ub >= lb of the same geometry)"""
    wl = workloads()
    base = sortie(wl, 71, n=9, yaw=15.0, tilt=4.0)
    poses = [p for q in base for p in (q, list(q))] + [list(p) for p in base[:3]]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, scale=2.0, lookahead=7)
    o = orc.OracleMap(force_float=force_float, scale=2.0)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, base[:8]) == o.prepare(wl.IDENTITY_PLANE, CAM, base[:8])
    for k, p in enumerate(poses):
        img = frame(wl, 71, k)
        assert g.feed(img, p) == o.feed(img, p)
    assert g.sync() and compare_maps(g, o) == []
    g.close()
