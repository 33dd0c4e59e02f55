"""The fused output-side kernel (collapse_fused.hip) against the oracle: Ele::blend with and without neighbours, the 8U view and
save()'s whole-mosaic collapse for every band count (the pyramid regions a block stages in LDS change shape with it: one-pixel top
levels, borders of one pixel at the top of the padded square, a border as wide as a tile with eight bands), both pyramid types;
the list form (pf_blend_tiles) with page-locked and plain buffers; and the per-level kernels of rounds 1-5 (experiments library) as
a second opinion on a mosaic larger than the oracle finishes quickly.
Reference: Map2DFusion/MultiBandMap2DCPU.cpp:77-146 (Ele::blend), :149-188 (updateTexture), :806-840 (save)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import compare_maps, jitter_poses, workloads
from test_gpu_parity import run_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "pi-slam-fusion_amd", "libpifusion_exp.so")


@pytest.mark.parametrize("force_float", [0, 1])
@pytest.mark.parametrize("bands", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_blend_and_save_every_band_count(pf, orc, force_float, bands):
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(7, seed=40 + bands, step=(27.0, 21.0), yaw_deg=35)
    frames = [wl.smooth_frame(480, 640, k) ^ (wl.noise_frame(480, 640, k) & 31) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, force_float=force_float, band_number=bands, bg_color=201)
    assert compare_maps(g, o) == []
    tiles = sorted(o.tiles())
    full = [t for t in tiles if all((t[0] + dx, t[1] + dy) in tiles for dx in (-1, 0, 1) for dy in (-1, 0, 1))]
    assert full and len(full) < len(tiles)                       # both branches of Ele::blend (.cpp:90, :131)
    for (ix, iy) in tiles:
        assert np.array_equal(g.blend_tile_raw(ix, iy), o.blend_tile_raw(ix, iy)), (ix, iy, bands)
    got = g.blend_tiles(tiles)
    for k, (ix, iy) in enumerate(tiles):
        assert np.array_equal(got[k], o.blend_tile(ix, iy)), (ix, iy, bands)
    (gs, gorg), (os_, oorg) = g.save_to_memory(), o.save()
    assert gorg == oorg and np.array_equal(gs, os_)
    assert (gs == 201).any()                                      # the mosaic's bounding box has holes: background (.cpp:840)


def test_low_quality_show_every_tile_alone(pf, orc):
    """HighQualityShow = 0: every tile takes "blend by self" (.cpp:131-145) -- borders of the tile itself at every level."""
    wl = workloads()
    cam, poses = wl.cfg1(6, step=30.0)
    frames = [wl.noise_frame(480, 640, k) for k in range(len(poses))]
    for ff in (0, 1):
        g, o = run_pair(pf, orc, cam, poses, frames, high_quality_show=0, force_float=ff, band_number=6)
        tiles = sorted(o.tiles())
        got = g.blend_tiles(tiles)
        for k, (ix, iy) in enumerate(tiles):
            assert np.array_equal(got[k], o.blend_tile(ix, iy))
            assert np.array_equal(g.blend_tile_raw(ix, iy), o.blend_tile_raw(ix, iy))


def test_blend_tiles_list_form_and_buffers(pf, orc):
    """pf_blend_tiles: caller-chosen tiles in caller order, a tile without pyramid keeps the buffer's bytes, Ischanged untouched;
    a page-locked buffer (pf_host_alloc) receives the same bytes as a plain one."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(8, seed=21, step=(25.0, 22.0), yaw_deg=40)
    frames = [wl.smooth_frame(480, 640, k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames)
    tiles = sorted(o.tiles())
    ask = [tiles[3], (10 ** 6, 5), tiles[0], tiles[3], tiles[-1]]                    # a missing tile and a repeat in the list
    plain = np.full((len(ask), 256, 256, 3), 123, np.uint8)
    pinned = pf.host_array((len(ask), 256, 256, 3)); pinned[:] = 123
    assert g.blend_tiles(ask, out=plain) is not None and g.blend_tiles(ask, out=pinned) is not None
    for k, t in enumerate(ask):
        want = o.blend_tile(*t) if t in tiles else np.full((256, 256, 3), 123, np.uint8)
        assert np.array_equal(plain[k], want) and np.array_equal(pinned[k], want), (k, t)
    changed, imgs = g.blend_changed()                                                 # the list form left the flags alone
    assert sorted(changed) == tiles
    buf = pf.host_array((len(tiles) + 2, 256, 256, 3))
    for t in tiles[:3]:
        assert g.feed(frames[0], poses[0])                                            # sets flags again
    changed2, imgs2 = g.blend_changed(cap=len(tiles) + 2, out=buf)
    assert changed2 and all(np.array_equal(imgs2[k], g.blend_tile(*t)) for k, t in enumerate(changed2))
    (a, org_a), (b, org_b) = g.save_to_memory(), g.save_to_memory(alloc=pf.host_array)
    assert org_a == org_b and np.array_equal(a, b)
    assert pf.lib().pf_blend_tiles(g._h, None, 1, plain.ctypes.data) == 0


PER_LEVEL = """
import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from helpers import workloads, jitter_poses
from conftest import load_package
pf = load_package(); wl = workloads()
cam = [1600, 1200, 1250, 1250, 800, 600]
for ff, bands in ((0, 5), (1, 5), (0, 7), (1, 3)):
    poses = jitter_poses(14, seed=77, step=(190.0, 160.0), yaw_deg=25, height=260.0)
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff, band_number=bands, bg_color=9)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        assert g.feed(wl.smooth_frame(1200, 1600, k) ^ (wl.noise_frame(1200, 1600, k) & 15), p)
    g.sync()
    tiles = sorted(g.tiles())
    h = hashlib.sha256()
    h.update(g.blend_tiles(tiles).tobytes()); h.update(g.blend_tile_raw(*tiles[len(tiles) // 2]).tobytes()); h.update(g.save_to_memory()[0].tobytes())
    print("DIGEST", ff, bands, len(tiles), h.hexdigest())
"""


def test_fused_kernel_equals_per_level_kernels():
    """The same mosaics (up to ~150 tiles, 1600x1200 keyframes) blended and saved by the product library's fused kernel and by the
    per-level kernels of rounds 1-5 (experiments library, PF_BLEND_PER_LEVEL=1): identical bytes."""
    assert os.path.exists(EXP_LIB), "build the experiments library first (__graft_entry__.build())"
    probe = PER_LEVEL % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for env_add in ({}, {"PF_LIB": EXP_LIB, "PF_BLEND_PER_LEVEL": "1"}):
        env = dict(os.environ, **env_add)
        r = subprocess.run([sys.executable, "-c", probe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        outs.append([l for l in r.stdout.decode().splitlines() if l.startswith("DIGEST")])
    assert len(outs[0]) == 4 and outs[0] == outs[1], outs
    assert all(int(l.split()[3]) > 60 for l in outs[0]), outs[0]


def test_blend_tiles_more_than_one_launch(pf):
    """pf_blend_tiles cuts its list into launches of 4096 tiles (the result buffers in HBM are bounded) and its results into 32 MB pieces for the
    staging ring: a mosaic of more than 4096 tiles (640 x 480 keyframes at Map2D.Scale = 8: 300 tiles each, sixteen of them side by side)
    blended in one call equals the tiles blended one at a time -- across the launch boundary, at the list's ends, for a repeated and a missing tile."""
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    # footprint on the ground at 100 m: 128 x 96 m; keyframes 120 / 90 m apart: a little overlap, 4 x 4 of them
    poses = [[120.0 * (k % 4), 90.0 * (k // 4), -100.0, 0, 0, 0, 1] for k in range(16)]
    g = pf.Map2D.create(pf.TypeMultiBandCPU, False, scale=8.0, band_number=3)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        assert g.feed(wl.smooth_frame(480, 640, k) ^ (wl.noise_frame(480, 640, k) & 7), p)
    g.sync()
    tiles = sorted(g.tiles())
    assert len(tiles) > 4096 + 200, len(tiles)
    ask = tiles + [tiles[5], (10 ** 6, 3), tiles[-1]]
    out = np.full((len(ask), 256, 256, 3), 99, np.uint8)
    assert g.blend_tiles(ask, out=out) is not None
    rng = np.random.RandomState(5)
    probe = sorted(set([0, 1, 4094, 4095, 4096, 4097, len(tiles) - 1] + [int(i) for i in rng.randint(0, len(tiles), 30)]))
    for i in probe:
        assert np.array_equal(out[i], g.blend_tile(*tiles[i])), (i, tiles[i])
    assert np.array_equal(out[len(tiles)], out[5]) and (out[len(tiles) + 1] == 99).all() and np.array_equal(out[len(tiles) + 2], out[len(tiles) - 1])
    # the same through the draw() form with a cap below the tile count, twice: every tile exactly once
    xy1, im1 = g.blend_changed(cap=4200)
    xy2, im2 = g.blend_changed(cap=4200)
    assert len(xy1) == 4200 and len(xy1) + len(xy2) == len(tiles) and sorted(xy1 + xy2) == tiles
    for j in (0, 4095, 4096, 4199):
        assert np.array_equal(im1[j], out[tiles.index(xy1[j])])
    g.close()
