"""Row f3 of SURVEY 8: a DroneMap-style dataset (config.cfg / trajectory.txt / rgb/<name>.*) replayed through the HIP
path by both file drivers -- the C++ one (include/pifusion/TestSystem.h, the reference's backup/map2dfusion.cpp
testMap2D) and the Python one (tools/replay.py) -- against the oracle fed the same frames."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import jitter_poses, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_dataset(tmp, cam, poses, frames, plane, ppm=True, extra=""):
    os.makedirs(os.path.join(tmp, "rgb"))
    with open(os.path.join(tmp, "config.cfg"), "w") as f:
        f.write("// synthetic\nPlane = %s\nCamera.Paraments = [%s]\nGPS.Origin = 108.9 34.2 400\nMap2D.Type ?= 3\n%s" %
                (" ".join(repr(float(x)) for x in plane), " ".join(repr(float(x)) for x in cam), extra))
    with open(os.path.join(tmp, "trajectory.txt"), "w") as f:
        for k, p in enumerate(poses):
            f.write("%06d %s\n" % (k, " ".join(repr(float(x)) for x in p)))
            if ppm is None:
                continue                                   # the caller writes the frames itself
            if ppm:
                with open(os.path.join(tmp, "rgb", "%06d.ppm" % k), "wb") as g:
                    g.write(b"P6\n%d %d\n255\n" % (frames[k].shape[1], frames[k].shape[0]))
                    g.write(np.ascontiguousarray(frames[k][:, :, ::-1]).tobytes())
            else:
                np.save(os.path.join(tmp, "rgb", "%06d.npy" % k), frames[k])


def build_replay(tmp):
    exe = os.path.join(tmp, "pifusion_replay")
    lib = os.path.join(ROOT, "pi-slam-fusion_amd")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "cpp", "pifusion_replay.cpp"), "-o", exe,
                           "-L" + lib, "-l:libpifusion.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return exe


def test_cpp_driver_builds_and_reports_missing_dataset(pf, tmp_path):
    exe = build_replay(str(tmp_path))
    r = subprocess.run([exe, str(tmp_path / "nothing")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    assert r.returncode == 1 and b"Can't open file" in r.stdout
    # a dataset whose frames cannot be decoded loads nothing: testMap2D's -4
    wl = workloads()
    d = str(tmp_path / "d")
    write_dataset(d, [64, 48, 50, 50, 32, 24], jitter_poses(2, seed=1), [wl.noise_frame(48, 64, k) for k in range(2)], wl.IDENTITY_PLANE, ppm=False)
    r = subprocess.run([exe, d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    assert r.returncode == 1 and b"Loaded 0 frames" in r.stdout and b"-4" in r.stdout


def oracle_of(orc, cam, poses, frames, plane, n_prepare, rendered_from, ff):
    o = orc.OracleMap(force_float=ff)
    assert o.prepare(plane, cam, poses[:n_prepare])
    for k in range(rendered_from, len(poses)):
        assert o.feed(frames[k], poses[k])
    return o.save()[0]


@pytest.mark.gpu
@pytest.mark.parametrize("thread,ff", [(0, 0), (1, 1)])
def test_dataset_replay_equals_oracle(pf, orc, tmp_path, thread, ff):
    from PIL import Image
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(9, seed=21 + thread)
    frames = [wl.noise_frame(480, 640, 400 + k) for k in range(len(poses))]
    plane = [1.5, -2.0, 0.25, 0.0, 0.0, 0.0871557427, 0.9961946981]
    plane_poses = [list(pf.se3_mul(plane, p)) for p in poses]            # world poses whose plane-frame poses are `poses`
    d = str(tmp_path / "ds")
    write_dataset(d, cam, plane_poses, frames, plane, extra="PrepareFrameNum = 3\nVideo.fps = 0\n")
    # thread=1: the prepare frames are rendered first (Map2D.cpp:42); thread=0: they only size the grid
    ref = oracle_of(orc, cam, plane_poses, frames, plane, 3, 0 if thread else 3, ff)

    exe = build_replay(str(tmp_path))
    out = str(tmp_path / "cpp.png")
    r = subprocess.run([exe, d, "Map2D.Thread=%d" % thread, "MultiBandMap2DCPU.ForceFloat=%d" % ff, "Map.File2Save=" + out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert b"Loaded 3 frames" in r.stdout and b"Fed 6 frames" in r.stdout
    assert np.array_equal(np.asarray(Image.open(out).convert("RGB"))[:, :, ::-1], ref)

    out2 = str(tmp_path / "py.png")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "replay.py"), d, "--prepare", "3", "--thread", str(thread), "--out", out2]
    r = subprocess.run(cmd + (["--float"] if ff else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert np.array_equal(np.asarray(Image.open(out2).convert("RGB"))[:, :, ::-1], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("thread,extra_arg", [(0, None), (1, None), (0, "Map2D.DecodeOnHost=1")])
def test_jpeg_dataset_replay_equals_oracle(pf, orc, tmp_path, thread, extra_arg):
    """The dataset as the reference ships it: rgb/<name>.jpg.  Both file drivers decode the frames with the library
    (cv::imread's place, backup/map2dfusion.cpp:129-132); the oracle is fed what libjpeg-turbo (Pillow) decodes."""
    import io
    from PIL import Image
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(8, seed=77)
    plane = wl.IDENTITY_PLANE
    d = str(tmp_path / "ds")
    write_dataset(d, cam, poses, [None] * len(poses), plane, ppm=None, extra="PrepareFrameNum = 3\nVideo.fps = 0\n")
    frames = []
    for k in range(len(poses)):
        y, x = np.mgrid[0:480, 0:640]
        a = np.stack([(x * 3 + 40 * k) % 256, (y * 2 + x) % 256, ((x // 16 + y // 16) % 2) * 200 + 20], -1).astype(np.uint8)
        a = (a.astype(np.int32) + wl.noise_frame(480, 640, 900 + k) // 8).clip(0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", quality=88, subsampling=[2, 1, 0][k % 3], progressive=bool(k & 1))
        open(os.path.join(d, "rgb", "%06d.jpg" % k), "wb").write(b.getvalue())
        frames.append(np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))[:, :, ::-1]))
    ref = oracle_of(orc, cam, poses, frames, plane, 3, 0 if thread else 3, 0)

    exe = build_replay(str(tmp_path))
    out = str(tmp_path / "cpp.png")
    r = subprocess.run([exe, d, "Map2D.Thread=%d" % thread, "Map.File2Save=" + out] + ([extra_arg] if extra_arg else []),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert b"Loaded 3 frames" in r.stdout and b"Fed 5 frames" in r.stdout
    assert np.array_equal(np.asarray(Image.open(out).convert("RGB"))[:, :, ::-1], ref)

    out2 = str(tmp_path / "py.png")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay.py"), d, "--prepare", "3", "--thread", str(thread), "--out", out2],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert np.array_equal(np.asarray(Image.open(out2).convert("RGB"))[:, :, ::-1], ref)
