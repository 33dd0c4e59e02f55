"""Steady state of the pipelined path (pf_options.fused = 1): runs long enough for every ring of the host
engine to wrap several times -- the per-frame tile tables (64 slots), the shard need-masks, the staging slots
of host frames (max_queue + 4) and the GW buffer parity -- compared with the CPU oracle tile by tile.
bench.py times exactly this regime (frames 20..220 of a sortie), so this is the parity evidence of the
headline number.  Reference semantics: Map2DFusion/MultiBandMap2DCPU.cpp:476-555 (Apply: ties go to the newest
frame, a fresh tile level is copied unconditionally), :288-309 (feed), :606-635 (render thread)."""
import os
import time

import numpy as np
import pytest

from helpers import compare_maps, map_digest, workloads

pytestmark = pytest.mark.gpu

CAM = [640, 480, 500, 500, 320, 240]
N_LONG = 168          # > 2 * 64 (table ring) + 5 levels in flight


def sortie(n, seed=11):
    """640x480 serpentine flown again and again over a 3-row area: bounded tile count, every frame a full render
    with its own yaw / tilt, heavy overlap (the max-weight select is exercised on every pixel)."""
    wl = workloads()
    return wl.serpentine(CAM, 100.0, n, per_row=9, fwd_overlap=0.7, side_overlap=0.5, seed=seed,
                         yaw_jitter_deg=12.0, tilt_jitter_deg=4.0, max_rows=3)


def frame(k):
    wl = workloads()
    return wl.noise_frame(480, 640, 7000 + k % 23) if k % 3 else wl.smooth_frame(480, 640, k) ^ wl.noise_frame(480, 640, k % 5)


def oracle_run(orc, poses, force_float, n_prepare=12, **opt):
    wl = workloads()
    o = orc.OracleMap(band_num=opt.get("band_number", 5), force_float=force_float)
    assert o.prepare(wl.IDENTITY_PLANE, CAM, poses[:n_prepare])
    for k, p in enumerate(poses):
        assert o.feed(frame(k), p)
    return o


def gpu_run(pf, poses, force_float, thread=False, n_prepare=12, **opt):
    wl = workloads()
    g = pf.Map2D.create(pf.TypeMultiBandCPU, thread, force_float=force_float, fused=1, **opt)
    # thread=True renders the prepare frames first when they carry images (Map2D.cpp:42); none are given here
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:n_prepare])
    for k, p in enumerate(poses):
        if thread:
            # the reference's producer loop (Map2DFusion.cpp:309-327) waits while the queue is long: nothing is dropped
            t0 = time.time()
            while g.queueSize() >= 8:
                time.sleep(0.0005)
                assert time.time() - t0 < 30
        assert g.feed(frame(k), p)
    assert g.sync()
    if thread:
        assert g.stats()["dropped"] == 0
    assert g.stats()["rendered"] == len(poses)
    return g


@pytest.mark.parametrize("thread", [False, True])
@pytest.mark.parametrize("force_float", [0, 1])
def test_long_run_equals_oracle(pf, orc, force_float, thread):
    """168 host frames through the pipelined path: table ring wraps twice, frame-slot ring seven times."""
    poses = sortie(N_LONG)
    o = oracle_run(orc, poses, force_float)
    g = gpu_run(pf, poses, force_float, thread=thread)
    assert g.grid() == o.grid()
    assert compare_maps(g, o) == []
    # blend of a few tiles after the long run (Ele::blend reads what the pipeline wrote last)
    for (ix, iy) in o.tiles()[:6]:
        assert np.array_equal(g.blend_tile(ix, iy), o.blend_tile(ix, iy))


@pytest.mark.parametrize("force_float", [0, 1])
def test_long_run_three_shards_equal_oracle(pf, orc, force_float):
    """Three shard maps on one card, 80 frames each: the per-frame need-mask ring (64 slots) wraps; the union of the
    shards' tiles is the oracle's map."""
    poses = sortie(80, seed=5)
    o = oracle_run(orc, poses, force_float)
    want = map_digest(o)
    got = {}
    for r in range(3):
        g = gpu_run(pf, poses, force_float, shard_rank=r, shard_count=3, shard_block=1)
        for (ix, iy) in g.tiles():
            assert pf.tile_owner(g.opt, ix, iy) == r
        d = map_digest(g)
        assert not (set(d) & set(got))
        got.update(d)
        assert g.grid() == o.grid()
        g.close()
    assert got == want


@pytest.mark.parametrize("force_float", [0, 1])
def test_table_copy_twin(pf, force_float):
    """The same long run with every tile table staged and copied to HBM in the stream (the round-1 form, what canvases of more than 256
    tiles take in the product): digests must equal the default table path's.  The switch that forces it (PF_TABLE_COPY=1) exists in the
    experiments build of the library only (csrc/env.hpp), so the twin runs in a child process on that build."""
    import subprocess, sys
    poses = sortie(N_LONG, seed=3)
    a = map_digest(gpu_run(pf, poses, force_float))
    here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
    exp = os.path.join(root, "pi-slam-fusion_amd", "libpifusion_exp.so")
    assert os.path.exists(exp), "build the experiments library first (__graft_entry__.build())"
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from conftest import load_package\nimport test_gpu_steady_state as T\nfrom helpers import map_digest\n"
            "pf = load_package(); import ctypes as C\n"
            "d = map_digest(T.gpu_run(pf, T.sortie(T.N_LONG, seed=3), %d))\n"
            "c = (C.c_longlong * 8)(); pf.lib().pf_debug_form_counts(c)\n"
            "print('TWIN', json.dumps({'d': d, 'args_launches': c[7]}))\n") % (here, root, force_float)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PF_TABLE_COPY="1", PF_LIB=exp), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import json
    twin = json.loads([l for l in r.stdout.splitlines() if l.startswith("TWIN ")][-1][5:])
    assert twin["args_launches"] == 0                                 # no table travelled in the kernel arguments there
    assert twin["d"] == {k: list(v) for k, v in a.items()}


@pytest.mark.parametrize("force_float", [1, 0])
def test_bench_sortie_pipelined_equals_per_op(pf, force_float):
    """bench.py's cfg-A sortie (4000x3000, 225 keyframes resident in HBM): the pipelined path (fused=1, the one
    the headline is timed on -- with the cull of render_frame, which the other path does not have) and the
    one-kernel-per-reference-op path (fused=0, oracle-checked at small sizes and on 4000x3000 frames in
    test_gpu_at_size.py) must build the same map, tile for tile, for both pyramid types."""
    torch = pytest.importorskip("torch")
    wl = workloads()
    cam = [4000, 3000, 3000, 3000, 2000, 1500]
    n = 225
    poses = wl.serpentine(cam, 100.0, n, max_rows=16)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    frames = [torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(4)]
    torch.cuda.synchronize()
    digests = []
    for fused in (1, 0):
        m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, fused=fused)
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
        for k, p in enumerate(poses):
            assert m.feed_device(frames[k % 4].data_ptr(), 3000, 4000, p)
        assert m.sync()
        assert m.stats()["rendered"] == n
        digests.append(map_digest(m))
        m.close()
        del m
    assert len(digests[0]) > 1000 * 6
    assert digests[0] == digests[1]


@pytest.mark.parametrize("force_float", [0, 1])
def test_cull_leaves_results_and_changed_flags_alone(pf, orc, force_float):
    """The cull (tiles in which a keyframe cannot win the max-weight select at any level are left out of the launch) against
    the oracle, which renders every tile of every canvas: same pyramids, same blends -- and the tiles draw() is told about
    are still ALL tiles of the keyframe's canvas (Apply sets Ischanged on each, MultiBandMap2DCPU.cpp:553), culled or not.
    Map2D.Scale = 2.5 makes a keyframe's canvas ~11 x 9 tiles, so that whole tiles lie on the losing side of a neighbour."""
    wl = workloads()
    poses = sortie(40, seed=5)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, fused=1, scale=2.5)
    o = orc.OracleMap(force_float=force_float, scale=2.5)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:12]) and o.prepare(wl.IDENTITY_PLANE, CAM, poses[:12])
    for k, p in enumerate(poses):
        assert g.feed(frame(k), p) and o.feed(frame(k), p)
    assert g.sync()
    culled = g.culled_tiles()
    assert culled > 40, culled                                    # the cull did take tiles out, many of them
    assert compare_maps(g, o) == []
    g.blend_changed(cap=8192)                                     # draw(): clears the flags
    # one more keyframe a little beside the last one: it loses wherever the last one is closer, yet its whole canvas is flagged
    extra = list(poses[-1]); extra[0] += 6.0
    assert g.feed(frame(61), extra) and o.feed(frame(61), extra) and g.sync()
    assert g.culled_tiles() > culled
    coords, px = g.blend_changed(cap=8192)
    xs = sorted(set(c[0] for c in coords)); ys = sorted(set(c[1] for c in coords))
    assert len(coords) == len(xs) * len(ys) >= 40 and xs == list(range(xs[0], xs[-1] + 1)) and ys == list(range(ys[0], ys[-1] + 1))
    assert compare_maps(g, o) == []
    for (ix, iy), img in list(zip(coords, px))[:: max(1, len(coords) // 8)]:
        assert np.array_equal(img, o.blend_tile(ix, iy)), (ix, iy)
    g.close()


@pytest.mark.parametrize("force_float", [0, 1])
def test_cull_with_seven_bands_single_pixel_cells(pf, orc, force_float):
    """Seven bands: the pyramid's reach is 382 px (the need rectangles instead of the in-kernel test), the cull's cells are dilated by
    256 px, and at level 6 a tile is 4 x 4 pixels -- one pixel per cell, so that a 2x2 quad of the select lies across cells and is
    gated per pixel; level 7 (2 x 2) goes through the top-level select.  Map2D.Scale = 4: a canvas of ~12 x 9 tiles."""
    wl = workloads()
    poses = sortie(24, seed=9)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, fused=1, scale=4.0, band_number=7)
    o = orc.OracleMap(force_float=force_float, scale=4.0, band_num=7)
    assert g.prepare(wl.IDENTITY_PLANE, CAM, poses[:12]) and o.prepare(wl.IDENTITY_PLANE, CAM, poses[:12])
    assert g.num_levels == 8
    for k, p in enumerate(poses):
        assert g.feed(frame(k), p) and o.feed(frame(k), p)
    assert g.sync()
    assert g.culled_cells() > 100 and g.culled_tiles() > 10, (g.culled_cells(), g.culled_tiles())
    assert compare_maps(g, o) == []
    g.close()


def test_run_bytes_follow_the_cull(pf):
    """Round 5: pf_profile_read_run hands out, next to SURVEY 8d's bytes of every canvas tile, the bytes of the part of
    the canvases the timed launches' blocks PROCESSED.  With the cull off the two are equal; with it on the run bytes are smaller, never
    below the frames read once, and the full-canvas bytes do not move (the denominator of bench.py's roofline.frac is the same time)."""
    wl = workloads()
    poses = sortie(60)
    res = {}
    for cull in (False, True):
        m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, fused=1, scale=2.5)
        m.set_cull(cull)
        assert m.prepare(wl.IDENTITY_PLANE, CAM, poses[:12])
        for k in range(20):
            assert m.feed(frame(k), poses[k])
        m.sync(); m.profile_reset(); m.profile_enable(1)
        for k in range(20, 60):
            assert m.feed(frame(k), poses[k])
        m.sync()
        p = m.profile_read()["level0_fused"]
        res[cull] = (p["alg_bytes"], p["alg_bytes_run"], p["launches"], m.culled_tiles() + m.culled_cells())
        m.close()
    (full0, run0, n0, c0), (full1, run1, n1, c1) = res[False], res[True]
    assert n0 == n1 == 40 and c0 == 0 and c1 > 0
    assert run0 == full0                                   # nothing left out: every block of every canvas ran
    assert full1 == full0                                  # SURVEY 8d's numerator ignores the cull
    frames_read = 40 * 640 * 480 * 3
    assert frames_read < run1 < 0.9 * full1, (run1, full1)
