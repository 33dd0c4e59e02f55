#!/usr/bin/env python3
"""Golden vectors for the file driver's JPEG leg (cv::imread(imgfile), backup/map2dfusion.cpp:129-132).

cv::imread decodes .jpg with libjpeg at its defaults; libjpeg is a third-party dependency that is not in /root/reference,
so the vectors come from libjpeg-turbo itself, through Pillow (`PIL.features.version("jpg")`, turbo build): every stream
below is decoded by it and the pixels are stored next to the stream.  Streams: Pillow's encoder for the shapes it writes
(4:4:4 / 4:2:2 / 4:2:0, optimised tables, progressive, restart markers, grey) and tests/jpeg_enc.py for the rest (other
sampling factors, non-interleaved scans, 16-bit codes, 16-bit quantisers, Adobe RGB).  Seeded, small, deterministic.
    python tests/golden/make_jpeg_vectors.py        -> tests/golden/jpeg_vectors.npz
"""
import io
import json
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import jpeg_enc  # noqa: E402


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([np.sin(x / 5.0) * 90 + 128, np.cos(y / 4.0) * 90 + 128, ((x + y) * 9) % 256], -1)
    a = (base + rng.normal(0, 18, (h, w, 3))) * (0.35 + 0.65 * ((x // 6 + y // 5) % 2))[..., None]
    a[: h // 3, : w // 4] = rng.integers(0, 2, (h // 3, w // 4, 1)) * 255            # hard black/white: the IDCT range limit
    return a.clip(0, 255).astype(np.uint8)


def main():
    streams, meta = [], []

    def pil(h, w, seed, mode="RGB", **kw):
        b = io.BytesIO()
        Image.fromarray(picture(h, w, seed)).convert(mode).save(b, "JPEG", **kw)
        streams.append(b.getvalue()); meta.append({"by": "pillow", "size": [h, w], "mode": mode, **{k: v for k, v in kw.items()}})

    def own(h, w, seed, grey=False, **kw):
        a = picture(h, w, seed)
        streams.append(jpeg_enc.encode(a[..., 1] if grey else a, **kw))
        meta.append({"by": "tests/jpeg_enc.py", "size": [h, w], "grey": grey, **{k: (list(map(list, v)) if k == "sampling" else v) for k, v in kw.items()}})

    pil(29, 37, 1, quality=75, subsampling=2)
    pil(29, 37, 2, quality=75, subsampling=1)
    pil(29, 37, 3, quality=75, subsampling=0)
    pil(40, 56, 4, quality=30, subsampling=2, optimize=True)
    pil(40, 56, 5, quality=100, subsampling=2)
    pil(33, 17, 6, quality=92, subsampling=2, progressive=True)
    pil(33, 17, 7, quality=60, subsampling=1, progressive=True, restart_marker_rows=1)
    pil(24, 50, 8, quality=85, subsampling=2, restart_marker_blocks=2)
    pil(31, 31, 9, mode="L", quality=80)
    pil(31, 31, 10, mode="L", quality=80, progressive=True)
    pil(1, 1, 11, quality=90, subsampling=2)
    pil(2, 2, 12, quality=90, subsampling=2)
    pil(5, 3, 13, quality=90, subsampling=2)
    pil(3, 5, 14, quality=90, subsampling=1)
    pil(17, 8, 15, quality=50, subsampling=0, progressive=True, optimize=True)
    own(27, 35, 20, sampling=((1, 2), (1, 1), (1, 1)))
    own(27, 35, 21, sampling=((4, 1), (1, 1), (1, 1)))
    own(27, 35, 22, sampling=((2, 2), (2, 1), (1, 2)))
    own(27, 35, 23, sampling=((1, 1), (2, 2), (1, 1)))
    own(27, 35, 24, sampling=((2, 2), (1, 1), (1, 1)), interleaved=False, restart=3, long_codes=True)
    own(27, 35, 25, sampling=((2, 1), (1, 1), (1, 1)), q16=True, q=3)
    own(27, 35, 26, sampling=((1, 1), (1, 1), (1, 1)), colour="rgb")
    own(27, 35, 27, sampling=((2, 2), (1, 1), (1, 1)), jfif=False)
    own(27, 35, 28, grey=True, sampling=((2, 2),), restart=1)
    own(9, 4, 29, sampling=((2, 2), (1, 1), (1, 1)))           # chroma two samples wide: libjpeg falls back to replication
    own(12, 70, 30, sampling=((1, 4), (1, 1), (1, 1)), long_codes=True)

    arrays = {}
    for i, s in enumerate(streams):
        im = Image.open(io.BytesIO(s))
        arrays["stream%02d" % i] = np.frombuffer(s, np.uint8)
        arrays["rgb%02d" % i] = np.asarray(im.convert("RGB"))
    arrays["meta"] = np.frombuffer(json.dumps({"decoder": "libjpeg-turbo via Pillow %s, jpeglib %s" % (Image.__version__, features.version("jpg")),
                                               "cases": meta}).encode(), np.uint8)
    out = os.path.join(HERE, "jpeg_vectors.npz")
    np.savez_compressed(out, **arrays)
    print("%d streams, %d bytes -> %s (%d bytes)" % (len(streams), sum(len(s) for s in streams), out, os.path.getsize(out)))


if __name__ == "__main__":
    main()
