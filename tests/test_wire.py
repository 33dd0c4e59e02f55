"""Rows f1/f3 of SURVEY 8: the tracker->fusion queue, BGRA frames, the dataset wire format."""
import importlib
import os
import threading
import time

import numpy as np
import pytest

from helpers import compare_maps, jitter_poses, map_digest, workloads


def mods(pf):
    return (importlib.import_module("pi_slam_fusion_amd.datatrans"), importlib.import_module("pi_slam_fusion_amd.dataset"))


def test_datatrans_semantics(pf):
    dt, _ = mods(pf)
    q = dt.DataTrans()
    for k in range(35):
        q.product(k)
    assert q.size() == 30 and q.dropped == 5 and q.consumption() == 5          # oldest dropped (DataTrans.h:60-66)
    got = []
    t = threading.Thread(target=lambda: got.append(q2.consumption()))
    q2 = dt.DataTrans()
    t.start(); time.sleep(0.05); assert got == []                               # blocks while empty (:72-77)
    q2.product("x"); t.join(2); assert got == ["x"]


def write_dataset(tmp, cam, poses, frames, plane):
    os.makedirs(os.path.join(tmp, "rgb"))
    with open(os.path.join(tmp, "config.cfg"), "w") as f:
        f.write("// synthetic\nPlane = %s\nCamera.Paraments = [%s]\nGPS.Origin = 108.9 34.2 400\nMap2D.Type ?= 3\n" %
                (" ".join(repr(float(x)) for x in plane), " ".join(repr(float(x)) for x in cam)))
    with open(os.path.join(tmp, "trajectory.txt"), "w") as f:
        for k, p in enumerate(poses):
            f.write("%06d %s\n" % (k, " ".join(repr(float(x)) for x in p)))
            np.save(os.path.join(tmp, "rgb", "%06d.npy" % k), frames[k])


def test_dataset_format_roundtrip(pf, tmp_path):
    _, ds = mods(pf)
    wl = workloads()
    cam = [320, 240, 250, 250, 160, 120]
    poses = jitter_poses(4, seed=2)
    frames = [wl.noise_frame(240, 320, k) for k in range(4)]
    plane = [1.5, -2.0, 0.25, 0.0, 0.0, 0.0871557427, 0.9961946981]
    write_dataset(str(tmp_path), cam, poses, frames, plane)
    d = ds.DroneMapDataset(str(tmp_path))
    assert d.camera == cam and d.plane == plane and d.gps_origin == [108.9, 34.2, 400.0] and len(d) == 4
    img, pose = d.load(2)
    assert np.array_equal(img, frames[2]) and pose == [float(x) for x in poses[2]]
    # a JPEG frame decodes to BGR like cv::imread
    from PIL import Image
    rgb = np.zeros((8, 8, 3), np.uint8); rgb[..., 0] = 200
    Image.fromarray(rgb).save(os.path.join(str(tmp_path), "rgb", "000009.jpg"), quality=100)
    d.frames.append(("000009", poses[0]))
    bgr, _ = d.load(4)
    assert bgr.shape == (8, 8, 3) and bgr[..., 2].mean() > 180 and bgr[..., 0].mean() < 30


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [1, 0])
def test_bgra_frames_equal_bgr(pf, orc, fused):
    """The tracker hands BGRA (GImage 8UC4); dropping alpha in the gather == cvtColor BGRA2BGR first
    (TrackerOpt.cpp:376-380), i.e. the oracle fed the BGR pixels."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(5, seed=8)
    a = pf.Map2D.create(pf.TypeMultiBandCPU, False, fused=fused); b = pf.Map2D.create(pf.TypeMultiBandCPU, False, fused=fused)
    o = orc.OracleMap()
    assert a.prepare(wl.IDENTITY_PLANE, cam, poses) and b.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        bgr = wl.noise_frame(480, 640, 60 + k)
        bgra = np.concatenate([bgr, wl.noise_frame(480, 640, 900 + k)[:, :, :1]], axis=2)
        assert a.feed(bgr, p) and b.feed(bgra, p) and o.feed(bgr, p)
    a.sync(); b.sync()
    assert compare_maps(b, o) == []
    assert map_digest(a) == map_digest(b)


@pytest.mark.gpu
def test_live_wire_cfg4(pf, orc, tmp_path):
    """BASELINE cfg-4 boundary: a producer thread pushes (frame, pose) through DataTrans, the
    TestSystem loop feeds a thread=true map while queueSize() < 2; no drops, tiles equal the
    synchronous run."""
    dt, ds = mods(pf)
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(14, seed=41)
    frames = [wl.noise_frame(480, 640, 200 + k) for k in range(len(poses))]
    write_dataset(str(tmp_path), cam, poses, frames, wl.IDENTITY_PLANE)
    d = ds.DroneMapDataset(str(tmp_path))
    wire = dt.DataTrans()

    def producer():                       # the tracker thread: Trans.product({img, pose}) (TrackerOpt.cpp:382)
        for k in range(3, len(d)):
            wire.product(d.load(k)); time.sleep(0.004)
        wire.product(None)

    m = pf.Map2D.create(pf.TypeMultiBandCPU, True)
    first = [d.load(k) for k in range(3)]
    assert m.prepare(d.plane, d.camera, [p for _, p in first], images=[i for i, _ in first])
    th = threading.Thread(target=producer); th.start()
    fed = dt.feed_loop(m, lambda: wire.consumption(timeout=10), fps=0)
    th.join(); m.sync()
    assert fed == len(d) - 3 and wire.dropped == 0 and m.stats()["dropped"] == 0
    s = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert s.prepare(d.plane, d.camera, [p for _, p in first])
    for k in range(len(d)):
        img, p = d.load(k); assert s.feed(img, p)
    s.sync()
    assert map_digest(m) == map_digest(s)
    # ... and the oracle: the threaded map renders its prepare frames first (Map2D.cpp:42, .cpp:606-615), then the wire's
    o = orc.OracleMap()
    assert o.prepare(d.plane, d.camera, [p for _, p in first])
    for k in range(len(d)):
        img, p = d.load(k); assert o.feed(img, p)
    assert compare_maps(m, o) == []
    assert np.array_equal(m.save_to_memory()[0], o.save()[0])


def _gps_vectors():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gps_vectors.json")))


def test_overlay_message_equals_reference(pf):
    """Row f4: the Map2DUpdate message (MultiBandMap2DCPU.cpp:744-757) against strings printed by the reference's own
    operator<<(Point3d) / SE3 * Point3d / calcLngLatFromDistance (oracle/ref_gps.cpp -> tests/golden/gps_vectors.json):
    Python (overlay.py) and the C ABI (pf_format_map_update) reproduce every message byte for byte, and the GPS maths bit for bit."""
    ov = importlib.import_module("pi_slam_fusion_amd.overlay")
    g = _gps_vectors()
    assert len(g["lnglat"]) >= 16 and len(g["messages"]) >= 16
    for v in g["lnglat"]:
        assert ov.lnglat_from_distance(v["lng1"], v["lat1"], v["dx"], v["dy"]) == (v["lng2"], v["lat2"])
        assert pf.lnglat_from_distance(v["lng1"], v["lat1"], v["dx"], v["dy"]) == (v["lng2"], v["lat2"])
    for v in g["messages"]:
        want = v["cmd"]
        assert want.startswith("Map2DUpdate LastTexMat ") and len(want.split()) == 8
        assert all(len(f.split(".")[1]) == 6 for f in want.split()[2:])          # std::to_string: six decimals, not the call site's nine
        assert ov.format_map_update(pf, v["plane"], v["origin"], v["min"][0], v["min"][1], v["ele"], v["x"], v["y"]) == want
        assert pf.format_map_update(v["plane"], v["origin"], v["min"][0], v["min"][1], v["ele"], v["x"], v["y"]) == want


def test_overlay_gate(pf):
    """`updated && !inborder && Fuse2Google` (.cpp:744): rim tiles of the dense grid are never announced while HighQualityShow is on."""
    ov = importlib.import_module("pi_slam_fusion_amd.overlay")

    class FakeMap:
        def grid(self):
            return [10, 8, -2, -1], [-100.0, -50.0, 412.0, 359.6, 51.2, 0.2]
    plane, org = [0, 0, 0, 0, 0, 0, 1], [108.888931, 34.257287, 400.0]
    assert ov.tile_overlay_command(pf, FakeMap(), plane, org, -2, 3) is None              # x == 0: a neighbour outside the grid
    assert ov.tile_overlay_command(pf, FakeMap(), plane, org, 7, 6) is None               # x == w-1, y == h-1
    assert ov.tile_overlay_command(pf, FakeMap(), plane, org, 3, 3, fuse2google=False) is None
    s = ov.tile_overlay_command(pf, FakeMap(), plane, org, 0, 1)
    assert s == ov.format_map_update(pf, plane, org, -100.0, -50.0, 51.2, 2, 2)
    assert ov.tile_overlay_command(pf, FakeMap(), plane, org, -2, 3, high_quality_show=False) is not None


@pytest.mark.gpu
def test_overlay_from_map(pf):
    """pf_map_update_command on a real map: interior tiles with pyramids give the reference's text, rim tiles and
    tiles without pyramids give nothing."""
    ov = importlib.import_module("pi_slam_fusion_amd.overlay")
    wl = workloads()
    cam, poses = wl.cfg1(6, step=40.0)
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        assert m.feed(wl.noise_frame(480, 640, k), p)
    m.sync()
    org = [108.888931, 34.257287, 400.0]
    dims, geo = m.grid()
    told = 0
    for (ix, iy) in m.tiles():
        x, y = ix - dims[2], iy - dims[3]
        rim = x == 0 or y == 0 or x == dims[0] - 1 or y == dims[1] - 1
        got = m.map_update_command(ix, iy, org)
        want = None if rim else ov.format_map_update(pf, wl.IDENTITY_PLANE, org, geo[0], geo[1], geo[4], x, y)
        assert got == want, (ix, iy, got, want)
        told += got is not None
    assert told > 0
    assert m.map_update_command(10 ** 6, 10 ** 6, org) is None
    m.close()
