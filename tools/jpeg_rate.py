#!/usr/bin/env python3
"""Rates of the file driver's JPEG leg on this box (one host thread): a 4000 x 3000 keyframe as a camera writes it (4:2:0, quality 90).
  host decode (csrc/jpeg_decode.cpp), libjpeg-turbo through Pillow (SIMD), and the split decode -- Huffman on the host, IDCT / upsampling /
  colour on the GPU (csrc/jpeg_device.hip) -- alone and as pf_feed_jpeg into a map (decode + render of the keyframe).
usage: python tools/jpeg_rate.py [--frames 12] [--md out.md]        (kernel times: run under rocprofv3 --kernel-trace --stats)"""
import argparse, io, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=12); ap.add_argument("--md", default=None)
ap.add_argument("--decode-only", action="store_true", help="only N split decodes of the 4:2:0 q90 frame (the run to put under rocprofv3)")
a = ap.parse_args()
pf = bench.load_package()
import gc
gc.disable()          # the interpreter's heap collection (~40 ms once torch is imported) stays out of the millisecond loops timed below
import importlib
wl = importlib.import_module("pi_slam_fusion_amd.workloads")
from PIL import Image

cam = bench.CAM
h, w = cam[1], cam[0]
rng = np.random.default_rng(0)
y, x = np.mgrid[0:h, 0:w]
pic = ((np.sin(x / 37.0) * 60 + np.cos(y / 23.0) * 60 + 128)[..., None] + rng.normal(0, 12, (h, w, 3))).clip(0, 255).astype(np.uint8)
if a.decode_only:
    b = io.BytesIO(); Image.fromarray(pic).save(b, "JPEG", quality=90, subsampling=2); s = b.getvalue()
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    for _ in range(a.frames):
        pf.decode_jpeg_device(s, out.data_ptr(), h, w)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), pf.decode_jpeg(s))
    print("decoded %d frames" % a.frames)
    sys.exit(0)
rows = []
for name, kw in [("4:2:0 q90", dict(quality=90, subsampling=2)), ("4:2:0 q75", dict(quality=75, subsampling=2)), ("4:4:4 q90", dict(quality=90, subsampling=0)),
                 ("4:2:0 q90, restart interval = one MCU row", dict(quality=90, subsampling=2, restart_marker_rows=1)),
                 ("4:2:0 q90 progressive", dict(quality=90, subsampling=2, progressive=True))]:
    b = io.BytesIO(); Image.fromarray(pic).save(b, "JPEG", **kw); s = b.getvalue()
    n = a.frames
    gc.collect()
    hc0 = pf.jpeg_huffman_counts()
    t = time.perf_counter(); ref = None
    for _ in range(3):
        ref = pf.decode_jpeg(s)
    t_host = (time.perf_counter() - t) / 3
    t = time.perf_counter()
    for _ in range(3):
        pil = np.asarray(Image.open(io.BytesIO(s)).convert("RGB"))
    t_pil = (time.perf_counter() - t) / 3
    assert np.array_equal(ref[:, :, ::-1], pil)
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    pf.decode_jpeg_device(s, out.data_ptr(), h, w); torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)
    t = time.perf_counter()
    for _ in range(n):
        pf.decode_jpeg_device(s, out.data_ptr(), h, w)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t) / n
    # the GPU part alone: events around the queued work of one frame (upload + two kernels)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); pf.decode_jpeg_device(s, out.data_ptr(), h, w); e1.record(); torch.cuda.synchronize()
    # (the default stream carries both; the event pair brackets host time too, so this is an upper bound of the GPU part)
    hc = pf.jpeg_huffman_counts()
    rows.append((name, len(s) / 1e6, t_host, t_pil, t_dev, hc[2] if hc[0] > hc0[0] else 0))
    print("%-24s %.2f MB: host decode %.1f ms, Pillow %.1f ms, device decode %.1f ms per frame (Huffman pass on the GPU: %s)" %
          (name, len(s) / 1e6, t_host * 1e3, t_pil * 1e3, t_dev * 1e3, ("%d launches of up to 8 sweeps" % hc[2]) if rows[-1][5] else "no, host"), flush=True)

# into a map: keyframes of the bench sortie as JPEG streams
poses = wl.serpentine(cam, 100.0, a.frames + 4)
b = io.BytesIO(); Image.fromarray(pic).save(b, "JPEG", quality=90, subsampling=2); s = b.getvalue()
res = {}
for mode in ("feed_jpeg", "decode_then_feed", "batch4", "batch8", "batch16"):
    gc.collect()
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:4])
    for k in range(2):
        m.feed(pf.decode_jpeg(s), poses[k]) if mode == "decode_then_feed" else m.feed_jpeg(s, poses[k])
    m.sync()
    t = time.perf_counter()
    if mode.startswith("batch"):
        nb = int(mode[5:]); done = 0
        frames_b = 4 * nb
        poses_b = wl.serpentine(cam, 100.0, frames_b + nb + 4)
        assert all(m.feed_jpeg_batch([s] * nb, poses_b[2:2 + nb], threads=nb))       # untimed: the batch's pinned buffers are allocated here
        m.sync()
        t = time.perf_counter()
        for k in range(2 + nb, 2 + nb + frames_b, nb):
            assert all(m.feed_jpeg_batch([s] * nb, poses_b[k:k + nb], threads=nb)); done += nb
        m.sync()
        res[mode] = done / (time.perf_counter() - t)
        m.close()
        continue
    for k in range(2, 2 + a.frames):
        m.feed_jpeg(s, poses[k]) if mode == "feed_jpeg" else m.feed(pf.decode_jpeg(s), poses[k])
    m.sync()
    res[mode] = a.frames / (time.perf_counter() - t)
    res[mode + "_img"] = m.save_to_memory()[0]
    m.close()
assert np.array_equal(res["feed_jpeg_img"], res["decode_then_feed_img"])
print("into a map: pf_feed_jpeg %.1f keyframes/s, host decode + pf_feed %.1f keyframes/s (one host thread; same mosaic)" % (res["feed_jpeg"], res["decode_then_feed"]), flush=True)
print("pf_feed_jpeg_batch: %s keyframes/s with 4 / 8 / 16 frames and threads per batch (%d host cores)" % (" / ".join("%.0f" % res["batch%d" % n] for n in (4, 8, 16)), os.cpu_count()), flush=True)
if a.md:
    with open(a.md, "w") as f:
        f.write("| stream (4000 x 3000) | size | host decode (`jpeg_decode.cpp`) | libjpeg-turbo via Pillow (SIMD) | device decode (`jpeg_device.hip`), host thread time per frame | Huffman pass |\n|---|---|---|---|---|---|\n")
        for (name, mb, th, tp, td, rounds) in rows:
            f.write("| %s | %.2f MB | %.1f ms | %.1f ms | **%.1f ms** | %s |\n" % (name, mb, th * 1e3, tp * 1e3, td * 1e3, ("GPU, %d launches of sweeps" % rounds) if rounds else "host (serial)"))
        f.write("\ninto a map (fp32 pyramids, one host thread, the same mosaic both ways): `pf_feed_jpeg` **%.1f keyframes/s**, host decode + `pf_feed` %.1f keyframes/s\n" % (res["feed_jpeg"], res["decode_then_feed"]))
        f.write("\n`pf_feed_jpeg_batch` (Huffman passes of a batch side by side, %d host cores): **%s keyframes/s** with 4 / 8 / 16 frames and threads per batch\n" % (os.cpu_count(), " / ".join("%.0f" % res["batch%d" % n] for n in (4, 8, 16))))
