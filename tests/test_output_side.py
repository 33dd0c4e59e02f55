"""Row f4 of SURVEY 8: the output side of save() -- the file the reference gets from cv::imwrite
(Map2DFusion/MultiBandMap2DCPU.cpp:836-845).  The PNG writer is hand-rolled over zlib, so the CPU tests decode
the stream themselves (chunk framing, CRCs, inflate, filter bytes) as well as through PIL; the GPU tests compare the
saved file with the mosaic collapse in memory and with the oracle."""
import os
import struct
import zlib

import numpy as np
import pytest

from helpers import jitter_poses, workloads


def decode_png(path):
    """minimal PNG reader: 8-bit RGB, filter 0 rows only (what the writer emits); checks every chunk CRC"""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr, seen_end = 8, b"", None, False
    while pos < len(b):
        n, tag = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])
        assert crc == (zlib.crc32(tag + data) & 0xffffffff), "bad CRC in %s" % tag
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat += data
        elif tag == b"IEND":
            seen_end = True
        pos += 12 + n
    assert seen_end and pos == len(b) and hdr is not None
    w, h, depth, ctype, comp, filt, inter = hdr
    assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)
    raw = zlib.decompress(idat)
    assert len(raw) == h * (1 + 3 * w)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any()                      # filter type 0 on every scanline
    return rows[:, 1:].reshape(h, w, 3)              # RGB


def decode_ppm(path):
    b = open(path, "rb").read()
    parts = b.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = [int(x) for x in parts[1].split()]
    assert len(parts[3]) == w * h * 3
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (256, 256), (300, 1030)])
def test_write_image_png_and_ppm_decode_back(pf, tmp_path, shape):
    from PIL import Image
    rng = np.random.RandomState(shape[0] * 7 + shape[1])
    bgr = rng.randint(0, 256, shape + (3,)).astype(np.uint8)
    if shape == (256, 256):
        bgr[:] = 17                                   # highly compressible: one short IDAT
    png, ppm = str(tmp_path / "a.png"), str(tmp_path / "a.ppm")
    assert pf.write_image(png, bgr) and pf.write_image(ppm, bgr)
    rgb = bgr[:, :, ::-1]
    assert np.array_equal(decode_png(png), rgb)
    assert np.array_equal(np.asarray(Image.open(png).convert("RGB")), rgb)
    assert np.array_equal(decode_ppm(ppm), rgb)
    assert np.array_equal(np.asarray(Image.open(ppm).convert("RGB")), rgb)


def test_write_image_large_stream_splits_into_idat_chunks(pf, tmp_path):
    """incompressible 1.2 MP image: the deflate stream is longer than the writer's 1 MiB buffer"""
    bgr = np.random.RandomState(3).randint(0, 256, (700, 600, 3)).astype(np.uint8)
    png = str(tmp_path / "big.PNG")
    assert pf.write_image(png, bgr)
    assert open(png, "rb").read().count(b"IDAT") >= 2
    assert np.array_equal(decode_png(png), bgr[:, :, ::-1])
    # the zlib stream is made in bands of 256 rows on several threads (raw deflate streams end to end behind one header, Adler-32 combined):
    # one stream to every reader, whatever the band count -- exactly one band, one row more, several, compressible and not
    from PIL import Image
    for rows, cols, smooth in ((256, 40, False), (257, 33, True), (512, 17, False), (1100, 90, True), (1, 1, False)):
        a = np.random.RandomState(rows).randint(0, 256, (rows, cols, 3)).astype(np.uint8)
        if smooth:
            a = (np.add.outer(np.arange(rows), np.arange(cols))[:, :, None] // 3 + np.arange(3)).astype(np.uint8)
        p = str(tmp_path / ("b%d.png" % rows))
        assert pf.write_image(p, a)
        assert np.array_equal(decode_png(p), a[:, :, ::-1]) and np.array_equal(np.asarray(Image.open(p).convert("RGB")), a[:, :, ::-1]), rows


def test_write_image_fails_like_imwrite(pf, tmp_path):
    bgr = np.zeros((4, 4, 3), np.uint8)
    assert not pf.write_image(str(tmp_path / "no" / "such" / "dir.png"), bgr)      # bool false, no throw
    assert pf.lib().pf_write_image(None, bgr.ctypes.data, 4, 4) == 0
    assert pf.lib().pf_write_image(b"x.png", None, 4, 4) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("ff,bg", [(0, 0), (1, 255)])
def test_save_file_equals_memory_and_oracle(pf, orc, tmp_path, ff, bg):
    """pf_save -> .png / .ppm decode to save_to_memory's pixels, which equal the oracle's save()"""
    from PIL import Image
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(5, seed=12)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff, bg_color=bg)
    o = orc.OracleMap(force_float=ff, bg_color=bg)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        img = wl.noise_frame(480, 640, 300 + k)
        assert g.feed(img, p) and o.feed(img, p)
    mem, org = g.save_to_memory()
    ref, oorg = o.save()
    assert org == oorg and np.array_equal(mem, ref)
    for ext in (".png", ".ppm"):
        f = str(tmp_path / ("result" + ext))
        assert g.save(f)
        assert np.array_equal(np.asarray(Image.open(f).convert("RGB"))[:, :, ::-1], ref)
    assert np.array_equal(decode_png(str(tmp_path / "result.png"))[:, :, ::-1], ref)
    assert not g.save(str(tmp_path / "missing" / "x.png"))
