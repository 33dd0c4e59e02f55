"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle
on the same seeded inputs.  Bar: bit-exact for the int16 pyramids, the fp32 weights
and -- since the kernels keep the oracle's operation order -- the fp32 pyramids too
(the north star allows 1 ULP per band; the tests assert 0 and report ULPs on failure).
Reference path: Map2DFusion/MultiBandMap2DCPU.cpp:288-558 (feed/renderFrame),
:77-146 (Ele::blend), :779-847 (save)."""
import numpy as np
import pytest

from helpers import compare_maps, jitter_poses, workloads, map_digest, ulp_diff

pytestmark = pytest.mark.gpu


def run_pair(pf, orc, cam, poses, frames, n_prepare=None, **opt):
    wl = workloads()
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, **opt)
    o = orc.OracleMap(band_num=opt.get("band_number", 5), force_float=opt.get("force_float", 0),
                      weight_type=opt.get("weight_type", 0), high_quality=opt.get("high_quality_show", 1),
                      bg_color=opt.get("bg_color", 0), resolution=opt.get("resolution", 0.0), scale=opt.get("scale", 1.0))
    prep = poses[:n_prepare] if n_prepare else poses
    assert g.prepare(wl.IDENTITY_PLANE, cam, prep) == o.prepare(wl.IDENTITY_PLANE, cam, prep) == True
    for img, p in zip(frames, poses):
        assert g.feed(img, p) == o.feed(img, p)
    assert g.sync()
    assert g.grid() == o.grid()
    return g, o


@pytest.mark.parametrize("fused", [1, 2, 3, 0])
@pytest.mark.parametrize("force_float", [0, 1])
@pytest.mark.parametrize("content", ["noise", "smooth"])
def test_cfg1_plumbing(pf, orc, force_float, content, fused):
    """BASELINE.json configs[0]: 10 synthetic 640x480 frames, identity rotation."""
    wl = workloads()
    cam, poses = wl.cfg1()
    gen = wl.noise_frame if content == "noise" else wl.smooth_frame
    frames = [gen(480, 640, k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, force_float=force_float, fused=fused)
    assert compare_maps(g, o) == []


@pytest.mark.parametrize("fused", [1, 2, 3, 0])
@pytest.mark.parametrize("force_float", [0, 1])
@pytest.mark.parametrize("bands", [0, 1, 3, 5, 7, 8])
def test_perspective_and_spread(pf, orc, force_float, bands, fused):
    """Rotated / tilted frames; grid prepared from 2 poses so later frames hit spreadMap
    (.cpp:360-379, 561-604); band counts on both sides of the SSE-tail boundary."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(9, seed=7 + bands)
    frames = [wl.noise_frame(480, 640, 100 + k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, n_prepare=2, force_float=force_float, band_number=bands, fused=fused)
    assert compare_maps(g, o) == []


@pytest.mark.parametrize("force_float", [0, 1])
def test_camera_above_plane_weight_type_scale(pf, orc, force_float):
    """t.z > 0 branch of the down-look gate (.cpp:335-336), WeightType=1 (.cpp:415), Map2D.Scale=0.5."""
    wl = workloads()
    cam = [320, 240, 260, 250, 158.5, 121.25]
    poses = jitter_poses(6, seed=3, step=(11.0, 5.0), height=60.0, below=False)
    frames = [wl.smooth_frame(240, 320, k) ^ wl.noise_frame(240, 320, k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, force_float=force_float, weight_type=1, scale=0.5)
    assert compare_maps(g, o) == []


def test_rejections(pf, orc):
    """bool-return convention (.cpp:290, 319-323, 340-343)."""
    wl = workloads()
    cam, poses = wl.cfg1(3)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    img = wl.noise_frame(480, 640, 0)
    assert g.feed(img, poses[0]) is False                       # before prepare
    assert g.prepare(wl.IDENTITY_PLANE, cam, [[0, 0, -5, 0, 0, 0, 1], [0, 0, 5, 0, 0, 0, 1]]) is False   # z straddles 0
    assert g.prepare(wl.IDENTITY_PLANE, [640, 480, 0, 500, 320, 240], poses) is False
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses)
    assert g.feed(wl.noise_frame(240, 320, 0), poses[0]) is False  # wrong size
    assert g.feed(np.zeros((480, 640, 1), np.uint8), poses[0]) is False  # wrong type
    s = np.sin(np.radians(80) / 2); c = np.cos(np.radians(80) / 2)
    assert g.feed(img, [0, 0, -100, s, 0, 0, c]) is False       # oblique view
    assert g.feed(img, poses[0]) is True
    assert g.stats()["rendered"] == 1
    # a row step smaller than a row of pixels is refused at the boundary (cv::Mat could never say that)
    import ctypes as C
    bad = pf.Image(480, 640, pf.PF_8UC3, img.ctypes.data, 640 * 3 - 1)
    pose = (C.c_double * 7)(*poses[1])
    assert pf.lib().pf_feed(g._h, C.byref(bad), pose) == 0
    assert b"row step" in pf.lib().pf_last_error()
    assert g.stats()["rendered"] == 1
    assert pf.Map2D.create(pf.NoType) is None and pf.Map2D.create(pf.TypeRender) is None


@pytest.mark.parametrize("force_float", [0, 1])
def test_blend_and_save(pf, orc, force_float):
    """Ele::blend with and without the full 3x3 neighbourhood, updateTexture's 8U view, save()."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(8, seed=21, step=(25.0, 22.0), yaw_deg=40)
    frames = [wl.smooth_frame(480, 640, k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, force_float=force_float, bg_color=37)
    assert compare_maps(g, o) == []
    tiles = o.tiles()
    n_full = 0
    for (ix, iy) in tiles:
        full = all((ix + dx, iy + dy) in tiles for dx in (-1, 0, 1) for dy in (-1, 0, 1))
        n_full += full
        assert np.array_equal(g.blend_tile_raw(ix, iy), o.blend_tile_raw(ix, iy)), (ix, iy, full)
        assert np.array_equal(g.blend_tile(ix, iy), o.blend_tile(ix, iy)), (ix, iy, full)
    assert 0 < n_full < len(tiles)
    changed, imgs = g.blend_changed()
    assert sorted(changed) == sorted(tiles)
    for (ix, iy), im in zip(changed, imgs):
        assert np.array_equal(im, o.blend_tile(ix, iy))
    assert g.blend_changed()[0] == []                           # Ischanged cleared (.cpp:186)
    (gs, gorg), (os_, oorg) = g.save_to_memory(), o.save()
    assert gorg == oorg and np.array_equal(gs, os_)
    assert g.blend_tile(10 ** 6, 0) is None


def test_low_quality_show_blends_alone(pf, orc):
    wl = workloads()
    cam, poses = wl.cfg1(6, step=30.0)
    frames = [wl.noise_frame(480, 640, k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, high_quality_show=0)
    for (ix, iy) in o.tiles():
        assert np.array_equal(g.blend_tile_raw(ix, iy), o.blend_tile_raw(ix, iy))


def test_threaded_feed_equals_synchronous(pf, orc):
    """thread=true: queue (cap 20, drop-oldest, .cpp:298-304) + render thread; prepare frames are
    rendered first (Map2D.cpp:42).  With no overload the tiles equal the synchronous run."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(12, seed=5)
    frames = [wl.noise_frame(480, 640, 300 + k) for k in range(len(poses))]
    gt = pf.Map2D.create(pf.TypeMultiBandCPU, True)
    assert gt.prepare(wl.IDENTITY_PLANE, cam, poses[:3], images=frames[:3])
    for img, p in zip(frames[3:], poses[3:]):
        assert gt.feed(img, p)
    assert gt.sync() and gt.queueSize() == 0
    gs = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert gs.prepare(wl.IDENTITY_PLANE, cam, poses[:3])
    for img, p in zip(frames, poses):
        assert gs.feed(img, p)
    gs.sync()
    assert gt.stats()["dropped"] == 0 and gt.stats()["rendered"] == len(poses)
    assert map_digest(gt) == map_digest(gs)


def test_queue_overflow_drops_oldest(pf):
    wl = workloads()
    cam, poses = wl.cfg1(4)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, True, max_queue=2)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses)
    img = wl.noise_frame(480, 640, 0)
    for k in range(40):
        assert g.feed(img, poses[k % 4])
        assert g.queueSize() <= 2
    g.sync()
    st = g.stats()
    assert st["rendered"] + st["dropped"] == 40


def test_device_resident_feed_and_idempotence(pf):
    """pf_feed_device (frame already in HBM) equals pf_feed; feeding the same frame twice
    leaves every tile unchanged (ties resolve to the newest, identical, frame: .cpp:521)."""
    torch = pytest.importorskip("torch")
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(5, seed=11)
    frames = [wl.noise_frame(480, 640, 40 + k) for k in range(len(poses))]
    a = pf.Map2D.create(pf.TypeMultiBandCPU, False); b = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert a.prepare(wl.IDENTITY_PLANE, cam, poses) and b.prepare(wl.IDENTITY_PLANE, cam, poses)
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    for f, d, p in zip(frames, dev, poses):
        assert a.feed(f, p)
        assert b.feed_device(d.data_ptr(), 480, 640, p)
    a.sync(); b.sync()
    da = map_digest(a)
    assert da == map_digest(b)
    assert a.feed(frames[-1], poses[-1]); a.sync()
    assert map_digest(a) == da


@pytest.mark.parametrize("fused", [1, 2, 3, 0])
@pytest.mark.parametrize("force_float", [0, 1])
def test_full_size_frame_against_oracle(pf, orc, force_float, fused):
    """BASELINE.json configs[1] geometry: one 4000x3000 frame, 5 bands, vs the oracle (seconds on CPU)."""
    wl = workloads()
    cam, poses = wl.cfg2(3)
    frame = wl.noise_frame(3000, 4000, 9)
    g, o = run_pair(pf, orc, cam, poses[:1], [frame], force_float=force_float, fused=fused)
    assert len(o.tiles()) >= 200
    assert compare_maps(g, o) == []


def test_full_size_properties(pf):
    """Size-independent properties at BASELINE's full frame size, no oracle involved:
    a constant-colour frame has zero Laplacian bands below the top and the constant on top
    wherever the pyramid support lies inside the footprint; blend of it returns the constant."""
    wl = workloads()
    cam, poses = wl.cfg2(2)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses)
    frame = np.full((3000, 4000, 3), 77, np.uint8)
    assert g.feed(frame, poses[0]) and g.sync()
    tiles = g.tiles()
    xs = sorted({t[0] for t in tiles}); ys = sorted({t[1] for t in tiles})
    cx, cy = xs[len(xs) // 2], ys[len(ys) // 2]          # a tile in the middle of the footprint
    for lv in range(g.num_levels):
        lap, w = g.tile_level(cx, cy, lv)
        assert (w > 0).all()
        assert (lap == (77 if lv == g.num_levels - 1 else 0)).all()
    assert (g.blend_tile(cx, cy) == 77).all()


@pytest.mark.parametrize("force_float", [0, 1])
def test_stress_geometry_8000x6000_7band(pf, orc, force_float):
    """BASELINE.json configs[4] geometry: one 8000x6000 frame, 7-band blend (pyramid levels down to
    2x2 per tile, SSE-tail rule active on the upper levels), against the oracle."""
    wl = workloads()
    cam = [8000, 6000, 6000, 6000, 4000, 3000]
    poses = wl.serpentine(cam, 100.0, 2, seed=7)
    frame = wl.noise_frame(6000, 8000, 5)
    g, o = run_pair(pf, orc, cam, poses[:1], [frame], force_float=force_float, band_number=7)
    assert g.num_levels == 8 and len(o.tiles()) >= 800
    assert compare_maps(g, o) == []


def test_row_padded_frames(pf, orc):
    """cv::Mat::step > cols*channels (a ROI of a wider buffer) is honoured by the H2D copy: the frame lies in HBM
    byte for byte as the caller holds it (checked by reading the staging slot back after every feed, before any
    comparison of results), and the map equals both a map fed the packed pixels and the oracle."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(3, seed=4)
    a = pf.Map2D.create(pf.TypeMultiBandCPU, False); b = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    o = orc.OracleMap()
    assert a.prepare(wl.IDENTITY_PLANE, cam, poses) and b.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        wide = wl.noise_frame(480, 700, 70 + k)
        view = wide[:, 30:670]                         # 640 columns, row step 2100 bytes
        assert view.strides[0] == 2100 and not view.flags["C_CONTIGUOUS"]
        assert a.feed(view, p)
        got = a.read_last_frame()
        assert got is not None and got.size == 479 * 2100 + 1920
        flat = wide.reshape(-1)[90:90 + got.size]      # the caller's bytes from the view's first pixel on, padding included
        assert np.array_equal(got, flat), "strided frame %d: %d bytes differ in HBM" % (k, int((got != flat).sum()))
        packed = np.ascontiguousarray(view)
        assert b.feed(packed, p) and o.feed(packed, p)
        assert np.array_equal(b.read_last_frame(), packed.reshape(-1))
    a.sync(); b.sync()
    assert compare_maps(a, o) == []
    assert map_digest(a) == map_digest(b)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 64), (2, 48), (40, 1), (3, 3)])
def test_degenerate_frame_shapes(pf, orc, shape):
    """frames of one row / one column / a few pixels: every canvas pixel takes the border paths (the fast path's
    `row < rows-2` test must not wrap for a one-row frame; found by tests/cpp/warp_index_check.cpp)"""
    wl = workloads()
    rows, cols = shape
    cam = [cols, rows, 40.0, 40.0, cols / 2.0, rows / 2.0]
    poses = [[0.7 * k, 0.3 * k, -100.0] + wl.quat_axis((0, 0, 1), 0.2 * k) for k in range(3)]
    for ff in (0, 1):
        g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff); o = orc.OracleMap(force_float=ff)
        assert g.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
        for k, p in enumerate(poses):
            img = wl.noise_frame(rows, cols, 500 + k)
            assert g.feed(img, p) == o.feed(img, p)
        assert compare_maps(g, o) == []
    s = pf.Map2D.create(pf.TypeCPU, False); so = orc.OracleMap(single_band=True)
    assert s.prepare(wl.IDENTITY_PLANE, cam, poses) and so.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        img = wl.noise_frame(rows, cols, 500 + k)
        assert s.feed(img, p) == so.feed(img, p)
    assert s.tiles() == so.tiles()
    for t in so.tiles():
        assert np.array_equal(s.tile_bgra(*t), so.tile_bgra(*t))


@pytest.mark.parametrize("weight_type", [0, 1])
def test_frame_wider_than_8191_px(pf, orc, weight_type):
    """A frame with a side of 8192 px or more: the radial weight's squares (dx^2, |dx| up to 4128) no longer fit 24 bits, so the
    computed form of weightImage (one fused rounding) would differ from the reference's two roundings (MultiBandMap2DCPU.cpp:411)
    by an ulp now and then and flip max-weight selects.  Such frames gather the weight from the plane instead."""
    wl = workloads()
    w, h = 8256, 96
    cam = [w, h, 9000.0, 9000.0, w / 2, h / 2]
    poses = [[k * 7.0, k * 0.3, -100.0] + wl.quat_axis((0, 0, 1), 0.01 * k) for k in range(3)]
    frames = [wl.noise_frame(h, w, 300 + k) for k in range(3)]
    g, o = run_pair(pf, orc, cam, poses, frames, band_number=3, weight_type=weight_type)
    assert compare_maps(g, o) == []
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("force_float", [0, 1])
def test_general_coordinate_forms(force_float):
    """The warp's general forms (IEEE division, saturating conversions, short clamps, looped border reflection) are
    what every pixel takes for a degenerate homography or a frame wider than a short; a healthy frame only reaches
    them through PF_FORCE_GENERAL, which is read once per process -- hence the child process."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import load_package\n"
        "from helpers import workloads, jitter_poses, compare_maps\n"
        "import test_gpu_parity as T\n"
        "pf = load_package(); from oracle import orc\n"
        "wl = workloads(); cam = [640, 480, 500, 500, 320, 240]\n"
        "poses = jitter_poses(4, seed=11); frames = [wl.noise_frame(480, 640, 30 + k) for k in range(4)]\n"
        "for fused in (1, 0):\n"
        "    g, o = T.run_pair(pf, orc, cam, poses, frames, n_prepare=2, force_float=%d, fused=fused)\n"
        "    bad = compare_maps(g, o)\n"
        "    assert bad == [], bad[:5]\n"
        "print('general forms ok')\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), force_float)
    # the switch exists in the experiments build of the library only (csrc/env.hpp): the same sources plus the forms not adopted
    exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pi-slam-fusion_amd", "libpifusion_exp.so")
    assert os.path.exists(exp), "build the experiments library first (__graft_entry__.build())"
    env = dict(os.environ, PF_FORCE_GENERAL="1", PF_LIB=exp)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "general forms ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_reserve_tiles_is_only_an_allocator_hint(pf):
    """pf_reserve_tiles pre-sizes the tile store; tiles, their order of creation and their contents are unchanged."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(4, seed=21)
    frames = [wl.noise_frame(480, 640, 90 + k) for k in range(4)]
    maps = []
    for reserve in (0, 3, 500):                              # none, fewer than needed (falls back to slabs), plenty
        m = pf.Map2D.create(pf.TypeMultiBandCPU, False)
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
        if reserve:
            assert m.reserve_tiles(reserve)
        for f, p in zip(frames, poses):
            assert m.feed(f, p)
        assert m.sync()
        maps.append(map_digest(m))
    assert maps[0] == maps[1] == maps[2]
    assert not pf.Map2D.create(pf.TypeMultiBandCPU, False).reserve_tiles(0)


@pytest.mark.parametrize("thread", [False, True])
def test_prepare_again_resets_the_map(pf, thread):
    """prepare() may be called again at any time: frames in flight are finished or dropped, the
    mosaic starts over -- the result equals a fresh map given the second preparation only."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    first = jitter_poses(5, seed=31)
    second = jitter_poses(4, seed=32, step=(-15.0, 9.0))
    frames = [wl.noise_frame(480, 640, 120 + k) for k in range(5)]
    a = pf.Map2D.create(pf.TypeMultiBandCPU, thread)
    assert a.prepare(wl.IDENTITY_PLANE, cam, first[:2])
    for f, p in zip(frames, first):
        assert a.feed(f, p)
    assert a.prepare(wl.IDENTITY_PLANE, cam, second[:2])     # no sync in between: the pipeline is still full
    for f, p in zip(frames, second):
        assert a.feed(f, p)
    assert a.sync()
    b = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert b.prepare(wl.IDENTITY_PLANE, cam, second[:2])
    for f, p in zip(frames, second):
        assert b.feed(f, p)
    assert b.sync()
    assert a.grid() == b.grid()
    assert map_digest(a) == map_digest(b)


def test_section_timers_carry_the_reference_names(pf):
    """pi::timer's sections (PIL/src/base/time/Timer.h:43-85; MultiBandMap2DCPU.cpp:476,555,563,602,628-630,722,742) on the
    host side of the HIP path: one feed / renderFrame / Apply per accepted keyframe, spreadMap when the grid grows,
    updateTexture per blend batch, save once; min <= mean <= max."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(6, seed=5)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    for k, p in enumerate(poses):
        assert g.feed(wl.noise_frame(480, 640, k), p)
    assert g.sync()
    assert len(g.blend_changed()[0]) > 0
    assert g.save_to_memory() is not None
    t = g.timers()
    assert set(t) == {"Map2D::feed", "MultiBandMap2DCPU::renderFrame", "MultiBandMap2DCPU::Apply", "MultiBandMap2DCPU::spreadMap",
                      "MultiBandMap2DCPU::updateTexture", "MultiBandMap2DCPU::save"}
    assert t["Map2D::feed"]["calls"] == t["MultiBandMap2DCPU::renderFrame"]["calls"] == t["MultiBandMap2DCPU::Apply"]["calls"] == len(poses)
    assert t["MultiBandMap2DCPU::updateTexture"]["calls"] >= 1          # (spreadMap only when a frame leaves the prepared grid)
    assert t["MultiBandMap2DCPU::save"]["calls"] >= 1
    for v in t.values():
        if v["calls"]:
            assert 0 < v["min_s"] <= v["mean_s"] <= v["max_s"] < 5.0
    assert t["MultiBandMap2DCPU::Apply"]["mean_s"] <= t["MultiBandMap2DCPU::renderFrame"]["mean_s"] <= t["Map2D::feed"]["mean_s"]
    g.timer_reset()
    assert all(v["calls"] == 0 for v in g.timers().values())
