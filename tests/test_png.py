"""cv::imread's PNG leg (csrc/png_decode.cpp) for the file driver: the pixels are the file's (PNG is lossless), converted the way cv::imread's
default flag converts them -- three 8-bit channels, BGR.  Pinned against Pillow's reader; the library's own PNG writer (save(), image_io.cpp)
is read back too.  Host code: no GPU."""
import io
import struct
import zlib

import numpy as np
import pytest

Image = pytest.importorskip("PIL.Image")


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(x * 7 + y) % 256, (y * 5) % 256, (x ^ y) % 256], -1) + rng.integers(0, 30, (h, w, 3))
    return a.clip(0, 255).astype(np.uint8)


def read(pf, tmp_path, im, name="a.png", **kw):
    p = str(tmp_path / name)
    im.save(p, **kw)
    return pf.read_image(p)[:, :, ::-1], p


def test_png_colour_types_and_filters(pf, tmp_path):
    for (h, w) in [(1, 1), (3, 5), (37, 29), (64, 100)]:
        a = picture(h, w, h + w)
        for kw in ({}, {"optimize": True}, {"compress_level": 0}, {"compress_level": 9}):
            got, _ = read(pf, tmp_path, Image.fromarray(a), **kw)
            assert np.array_equal(got, a), (h, w, kw)
        rgba = np.dstack([a, (a[..., 0] // 2 + 60).astype(np.uint8)])
        got, _ = read(pf, tmp_path, Image.fromarray(rgba))
        assert np.array_equal(got, a)                                          # alpha dropped, not blended
        g = a[..., 1]
        got, _ = read(pf, tmp_path, Image.fromarray(g))
        assert np.array_equal(got, np.stack([g] * 3, -1))
        la = np.dstack([g, a[..., 0]])
        got, _ = read(pf, tmp_path, Image.fromarray(la))
        assert np.array_equal(got, np.stack([g] * 3, -1))
        pim = Image.fromarray(a).quantize(colors=200)                           # colour type 3
        got, _ = read(pf, tmp_path, pim)
        assert np.array_equal(got, np.asarray(pim.convert("RGB")))
        pim16 = Image.fromarray(a).quantize(colors=13)                          # 4-bit palette indices
        got, _ = read(pf, tmp_path, pim16, bits=4)
        assert np.array_equal(got, np.asarray(pim16.convert("RGB")))
        bw = Image.fromarray((g > 128).astype(np.uint8) * 255).convert("1")     # 1-bit grey
        got, _ = read(pf, tmp_path, bw)
        assert np.array_equal(got, np.stack([np.asarray(bw).astype(np.uint8) * 255] * 3, -1))
        g16 = (g.astype(np.uint16) << 8) | 0x5A                                 # 16-bit grey: the high byte
        got, _ = read(pf, tmp_path, Image.fromarray(g16))
        assert np.array_equal(got, np.stack([g] * 3, -1))


def test_own_writer_is_read_back(pf, tmp_path):
    a = picture(130, 257, 3)
    p = str(tmp_path / "w.png")
    assert pf.write_image(p, a[:, :, ::-1])
    assert np.array_equal(pf.read_image(p)[:, :, ::-1], a)
    assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), a)


def chunks(b):
    pos = 8
    while pos < len(b):
        n, tag = struct.unpack(">I4s", b[pos:pos + 8])
        yield pos, n, tag
        pos += 12 + n


def test_png_refusals(pf, tmp_path):
    a = picture(20, 20, 1)
    bio = io.BytesIO(); Image.fromarray(a).save(bio, "PNG"); b = bytearray(bio.getvalue())
    def write(bb, name):
        p = str(tmp_path / name); open(p, "wb").write(bytes(bb)); return p
    # a flipped data byte: the chunk's CRC says so
    bad = bytearray(b); pos, n, tag = [c for c in chunks(b) if c[2] == b"IDAT"][0]; bad[pos + 8 + n // 2] ^= 0x40
    with pytest.raises(RuntimeError, match="CRC"):
        pf.read_image(write(bad, "crc.png"))
    # interlaced: refused (IHDR rewritten with its CRC)
    il = bytearray(b); il[8 + 8 + 12] = 1
    il[8 + 8 + 13:8 + 8 + 17] = struct.pack(">I", zlib.crc32(bytes(il[12:8 + 8 + 13])) & 0xffffffff)
    with pytest.raises(RuntimeError, match="interlaced"):
        pf.read_image(write(il, "il.png"))
    with pytest.raises(RuntimeError, match="truncated"):
        pf.read_image(write(b[:len(b) // 2], "cut.png"))
    with pytest.raises(RuntimeError):
        pf.read_image(write(b"\x89PNG\r\n\x1a\n" + b"\0" * 40, "junk.png"))
