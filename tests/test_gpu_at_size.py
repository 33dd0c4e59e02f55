"""BASELINE.json configs [2] and [4] exercised at size on one card (the 8-GPU node is the driver's to use):
  [2] 4000x3000 keyframes, tile grid sharded 8 ways -- eight shard maps (one per rank of a node) on this GPU, each fed
      every keyframe; their tiles partition the mosaic, and every tile equals the unsharded map's AND the oracle's;
  [4] 8000x6000 keyframes, 7-band blend, 65536-tile mosaic -- the tile store pre-sized for 256x256 tiles (16S: 57 GB of
      HBM on this one card), keyframes sampled over that area (seed 7), probed tiles bit-exact against the oracle."""
import numpy as np
import pytest

from helpers import compare_maps, map_digest, workloads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("force_float", [0, 1])
def test_cfg2_eight_way_sharding_at_full_frame_size(pf, orc, force_float):
    wl = workloads()
    cam = [4000, 3000, 3000, 3000, 2000, 1500]
    poses = wl.serpentine(cam, 100.0, 4, per_row=2)                     # 2 x 2 keyframes: forward and side overlap
    frames = [wl.noise_frame(3000, 4000, 40 + k) for k in range(2)]
    o = orc.OracleMap(force_float=force_float)
    ref = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float)
    shards = [pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, shard_rank=r, shard_count=8, shard_block=2) for r in range(8)]
    for m in [o, ref] + shards:
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        for m in [o, ref] + shards:
            assert m.feed(frames[k % 2], p)
    ref.sync()
    assert compare_maps(ref, o) == []
    dref, got = map_digest(ref), {}
    for r, m in enumerate(shards):
        m.sync()
        assert all(pf.tile_owner(m.opt, ix, iy) == r for ix, iy in m.tiles())
        d = map_digest(m)
        assert not (set(d) & set(got))
        got.update(d)
        rs = m.render_stats()
        assert rs["level0_px"] < 3.0 * rs["owned_px"]                   # a shard renders its cells + halo, not the whole canvas
    assert got == dref and len(ref.tiles()) >= 300
    assert sum(1 for m in shards if m.tiles()) == 8


@pytest.mark.parametrize("force_float", [1, 0])
def test_cfg2_timed_path_two_flight_lines_against_oracle(pf, orc, force_float):
    """The path bench.py times (device-resident 4000x3000 keyframes through pf_feed_device, fused = 1, cfg-A) against the ORACLE
    over several keyframes: three flight lines of the serpentine (forward AND side overlap, yaw / tilt jitter; from the second line on
    the cull of render_frame leaves out most of every canvas, and the level-0 blocks pick themselves by the tile table), so that the
    max-weight select runs on most pixels with real stored weights, level by level (MultiBandMap2DCPU.cpp:476-555); then
    Ele::blend on tiles whose 3x3 neighbourhood exists and the whole-mosaic save."""
    torch = pytest.importorskip("torch")
    wl = workloads()
    cam = [4000, 3000, 3000, 3000, 2000, 1500]
    poses = wl.serpentine(cam, 100.0, 12, per_row=4)                     # out along one line, back along the next, out along a third
    host = [wl.noise_frame(3000, 4000, 60 + k) for k in range(3)]
    dev = [torch.from_numpy(f).cuda() for f in host]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, fused=1)
    o = orc.OracleMap(force_float=force_float)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        assert g.feed_device(dev[k % 3].data_ptr(), 3000, 4000, p) and o.feed(host[k % 3], p)
    assert g.sync() and g.grid() == o.grid()
    assert compare_maps(g, o) == []
    tiles = o.tiles()
    have = set(tiles)
    inner = [t for t in tiles if all((t[0] + dx, t[1] + dy) in have for dx in (-1, 0, 1) for dy in (-1, 0, 1))]
    assert len(tiles) > 400 and len(inner) > 200 and g.culled_tiles() > 300 and g.culled_cells() > 1000
    for t in [inner[0], inner[len(inner) // 3], inner[2 * len(inner) // 3], inner[-1]]:
        assert np.array_equal(g.blend_tile(*t), o.blend_tile(*t)), t
    assert np.array_equal(g.save_to_memory()[0], o.save()[0])
    g.close()


def test_cfg4_65536_tile_store_and_7band_frames(pf, orc):
    torch = pytest.importorskip("torch")
    free, _ = torch.cuda.mem_get_info()
    if free < 70e9:
        pytest.skip("needs 57 GB of HBM for the tile store")
    wl = workloads()
    cam = [8000, 6000, 6000, 6000, 4000, 3000]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, band_number=7)
    o = orc.OracleMap(band_num=7)
    # mosaic of 256 x 256 tiles: resolution = 100 m / 6000 px, one tile = 256 px = 4.27 m -> 1092 m on a side
    side = 256 * 256 * (100.0 / 6000.0)
    corners = [[x, y, -100.0, 0, 0, 0, 1] for x in (0.0, side) for y in (0.0, side)]
    assert g.prepare(wl.IDENTITY_PLANE, cam, corners) and o.prepare(wl.IDENTITY_PLANE, cam, corners)
    assert g.num_levels == 8 and g.tile_bytes() > 850_000
    assert g.reserve_tiles(65536)                                        # 57 GB: the whole mosaic's tiles on one card
    assert torch.cuda.mem_get_info()[0] < free - 55e9
    rng = np.random.RandomState(7)
    frames = [wl.noise_frame(6000, 8000, 70 + k) for k in range(2)]
    n = 5
    for k in range(n):
        x, y = rng.uniform(80.0, side - 80.0, 2)
        q = wl.quat_axis((0, 0, 1), float(rng.uniform(-0.3, 0.3)))
        if k == n - 1:                                                   # the last one overlaps the first: real selects
            x, y = first[0] + 30.0, first[1] + 20.0
        pose = [float(x), float(y), -100.0] + q
        if k == 0:
            first = (float(x), float(y))
        assert g.feed(frames[k % 2], pose) and o.feed(frames[k % 2], pose)
    g.sync()
    assert g.grid() == o.grid()
    tiles = o.tiles()
    assert g.tiles() == tiles and len(tiles) > 3000
    spanx = max(t[0] for t in tiles) - min(t[0] for t in tiles); spany = max(t[1] for t in tiles) - min(t[1] for t in tiles)
    assert max(spanx, spany) > 100                                        # keyframes far apart in the 256 x 256 area
    probe = [tiles[i] for i in rng.choice(len(tiles), 48, replace=False)]
    for (ix, iy) in probe:
        for lv in range(8):
            gl, gw = g.tile_level(ix, iy, lv); ol, ow = o.tile_level(ix, iy, lv)
            assert np.array_equal(gw, ow) and np.array_equal(gl, ol), (ix, iy, lv)
    t = probe[0]
    assert np.array_equal(g.blend_tile(*t), o.blend_tile(*t))
