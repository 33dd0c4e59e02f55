"""CPU sanitizer builds of the host code that takes untrusted bytes or runs on several threads (SURVEY section 5's stance: sanitizer
options for the host tests; the reference's own known race is the unguarded deque read at MultiBandMap2DCPU.cpp:606-617).

  make -C tests/cpp asan   san_fuzz     AddressSanitizer + UndefinedBehaviorSanitizer: jpeg_decode.cpp (header walk, stuffing / RSTn stripping, the
                                        serial entropy pass, the full decode), the host simulation of the GPU's parallel Huffman pass
                                        (jpeg_huff_par.hpp: accepted implies equal to the serial pass), png_decode.cpp, image_io.cpp and its
                                        C entry points, TestSystem.h's readers (config.cfg, trajectory.txt, .ppm), DataTrans.h, plan_blend
  make -C tests/cpp tsan   san_threads  ThreadSanitizer: DataTrans producer / consumer, the decoders and file entry points on eight threads at
                                        once (the host side of pf_feed_jpeg_batch's workers), plan_blend on every rank at once

The corpus: every golden JPEG stream (tests/golden/jpeg_vectors.npz), encoder-made streams with restart intervals, PNG and PPM files, a
dataset's texts; san_fuzz runs each as it is and damaged (>= 10 000 damaged inputs in all).  Findings of the first run are kept as regression
files below (REGRESSIONS)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import jpeg_enc
from test_jpeg import picture, vectors

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def png_bytes(a, colour=2, filt=0):
    """a small PNG writer: colour type 2 (RGB8), 0 (grey8), 6 (RGBA8); one filter type for every row"""
    h, w = a.shape[:2]
    if colour == 0:
        rows = a[..., 1]
    elif colour == 6:
        rows = np.dstack([a, a[..., :1] // 2 + 50]).astype(np.uint8)
    else:
        rows = a
    rows = rows.reshape(h, -1).astype(np.uint8)
    bpp = {0: 1, 2: 3, 6: 4}[colour]
    raw = bytearray()
    for y in range(h):
        r = rows[y].astype(np.int32)
        if filt == 1:
            r = np.concatenate([r[:bpp], r[bpp:] - r[:-bpp]]) & 255
        elif filt == 2 and y:
            r = (r - rows[y - 1].astype(np.int32)) & 255
        raw.append(filt if not (filt == 2 and y == 0) else 0)
        raw += bytes(r.astype(np.uint8))
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def ppm_bytes(a, comment=False):
    h, w = a.shape[:2]
    return (b"P6\n" + (b"# made by the test\n" if comment else b"") + b"%d %d\n255\n" % (w, h)) + a.tobytes()


# inputs the first sanitizer runs tripped over, kept verbatim: (file name, bytes)
REGRESSIONS = [
    ("reg_ppm_digits.ppm", b"P6\n99999999999999999999 3\n255\n" + b"\0" * 27),                 # header number past INT_MAX (signed overflow in the digit loop)
    ("reg_ppm_area.ppm", b"P6\n70000 70000\n255\n" + b"\0" * 64),                               # w * h * 3 past INT_MAX, 14.7 GB asked for
    ("reg_ppm_eof.ppm", b"P6\n4 4\n25"),                                                         # header cut inside a number
]


def build_corpus(d):
    os.makedirs(d, exist_ok=True)
    n = 0
    for i, (case, stream, _) in enumerate(vectors()):
        open(os.path.join(d, "g%02d.jpg" % i), "wb").write(stream); n += 1
    shapes = [(40, 56, ((2, 2), (1, 1), (1, 1)), {"restart": 1}), (33, 47, ((1, 2), (1, 1), (1, 1)), {"restart": 2, "long_codes": True}),
              (130, 250, ((2, 1), (1, 1), (1, 1)), {"restart": 5}), (64, 64, ((1, 1), (1, 1), (1, 1)), {"restart": 3}), (200, 300, ((2, 2), (1, 1), (1, 1)), {"restart": 7}),
              (90, 120, ((2, 2), (1, 1), (1, 1)), {}), (9, 7, ((1, 1), (1, 1), (1, 1)), {"q16": True, "q": 3})]
    for j, (h, w, samp, kw) in enumerate(shapes):
        open(os.path.join(d, "e%02d.jpg" % j), "wb").write(jpeg_enc.encode(picture(h, w, 17 * j + 3), samp, **kw)); n += 1
    for j, (h, w, colour, filt) in enumerate([(1, 1, 2, 0), (13, 21, 2, 1), (37, 29, 0, 2), (30, 44, 6, 1), (64, 100, 2, 2)]):
        open(os.path.join(d, "p%02d.png" % j), "wb").write(png_bytes(picture(h, w, j + 1), colour, filt)); n += 1
    # headers that no longer describe their data (checksums right): sizes, bit depths and colour types changed under a valid zlib stream
    base = png_bytes(picture(20, 28, 3), 2, 1)
    def with_ihdr(b, **kw):
        w, h, depth, colour, comp, filt, lace = struct.unpack(">IIBBBBB", b[16:29])
        f = dict(w=w, h=h, depth=depth, colour=colour, comp=comp, filt=filt, lace=lace); f.update(kw)
        ih = struct.pack(">IIBBBBB", f["w"], f["h"], f["depth"], f["colour"], f["comp"], f["filt"], f["lace"])
        return b[:16] + ih + struct.pack(">I", zlib.crc32(b"IHDR" + ih) & 0xffffffff) + b[33:]
    for j, kw in enumerate([dict(w=29), dict(w=27), dict(h=21), dict(h=0x7fffffff), dict(w=0x40000000, h=4), dict(depth=16), dict(depth=1), dict(colour=6), dict(colour=3), dict(colour=4),
                            dict(colour=0), dict(lace=1), dict(depth=4, colour=3), dict(w=0), dict(comp=1), dict(filt=1)]):
        open(os.path.join(d, "q%02d.png" % j), "wb").write(with_ihdr(base, **kw)); n += 1
    for j, (h, w) in enumerate([(1, 1), (5, 3), (24, 40)]):
        open(os.path.join(d, "m%02d.ppm" % j), "wb").write(ppm_bytes(picture(h, w, j + 9), comment=j == 1)); n += 1
    # a dataset: config.cfg, trajectory.txt and the frames it names (frame0.jpg, frame1.png, frame2.ppm under rgb/)
    open(os.path.join(d, "config.cfg"), "w").write(
        "// phantom3-style dataset\nPlane = 0 0 0 0 0 0 1\nCamera.Paraments = [64 48 50 50 32 24]\nGPS.Origin = 108.9 34.2 400\nPrepareFrameNum ?= 2\nMap2D.Scale=0.5 # half\n")
    # (dirframe: san_fuzz makes rgb/dirframe.jpg a DIRECTORY -- it opens, and ftell() says LONG_MAX: found by the first runs, now refused)
    open(os.path.join(d, "trajectory.txt"), "w").write("".join("frame%d %g %g -100 0 0 0 1\n" % (k, 3.0 * k, 0.5 * k) for k in range(3)) + "dirframe 0 0 -100 0 0 0 1\nframe9 1 2 3\n")
    open(os.path.join(d, "frame0.jpg"), "wb").write(jpeg_enc.encode(picture(48, 64, 5), ((2, 2), (1, 1), (1, 1))))
    open(os.path.join(d, "frame1.png"), "wb").write(png_bytes(picture(48, 64, 6)))
    open(os.path.join(d, "frame2.ppm"), "wb").write(ppm_bytes(picture(48, 64, 7)))
    for name, b in REGRESSIONS:
        open(os.path.join(d, name), "wb").write(b)
    return n + 5 + len(REGRESSIONS)


@pytest.fixture(scope="module")
def san(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sanbuild"))
    r = subprocess.run(["make", "-C", os.path.join(HERE, "cpp"), "asan", "tsan", "OUT=" + out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    corpus = str(tmp_path_factory.mktemp("corpus"))
    nfiles = build_corpus(corpus)
    return out, corpus, nfiles


def test_address_and_undefined_behaviour_sanitizers(san, tmp_path):
    out, corpus, nfiles = san
    per_file = 10000 // nfiles + 1
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=2048", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    total = 0
    for seed in (1, 20261005):
        work = str(tmp_path / ("w%d" % seed)); os.makedirs(work)
        r = subprocess.run([os.path.join(out, "san_fuzz"), corpus, work, str(seed), str(per_file // 2 + 1)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        text = r.stdout.decode(errors="replace")
        assert r.returncode == 0 and "ERROR: AddressSanitizer" not in text and "runtime error" not in text and "VIOLATION" not in text, text[-4000:]
        line = [l for l in text.splitlines() if l.startswith("files ")][-1]
        f = dict(zip(["files", "mutants"], [int(line.split()[1]), int(line.split()[3])]))
        assert f["files"] == nfiles
        total += f["mutants"]
        # the run did real work on every reader: streams decoded AND refused, the parallel pass accepted streams and sent damaged ones to the serial pass
        num = lambda key: int(line.split(key)[1].split()[0])
        assert num("jpeg decoded ") > 300 and num(" refused ") > 100
        assert num("accepted ") > 100 and num("to-serial ") > 50 and num("unequal ") == 0
        assert num("png decoded ") >= 3 and num("png decoded ") + int(line.split("png decoded ")[1].split()[2]) > 300     # (a damaged zlib stream is refused by its checksum)
        assert num("ppm read ") > 20 and num("datasets opened ") > 100 and num("plans ") == 300
    assert total >= 10000


def test_thread_sanitizer(san, tmp_path):
    out, corpus, _ = san
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    r = subprocess.run([os.path.join(out, "san_threads"), corpus, "8", str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in text and "violations 0" in text, text[-4000:]
