"""Row f2 of SURVEY 8: Map2DCPU semantics (single 8-bit band, alpha-byte weights) -- the
`Map2D::create(TypeCPU)` / `TypeGPU` object of the reference (Map2DFusion/Map2DCPU.cpp:150-334),
against the oracle's restatement of it.  Bit-exact byte work."""
import numpy as np
import pytest

from helpers import jitter_poses, workloads


# ---------------------------------------------------------------- oracle pins (CPU)
def test_weight_byte_image(orc):
    w = orc.weight_image_8uc4(480, 640, 0)
    assert (w[..., :3] == 0).all() and w[240, 320, 3] == 254 and w[0, 0, 3] == 2      # dis*254 truncated, floor 2
    xc, yc = np.float32(320), np.float32(240)
    i, j = np.mgrid[0:480, 0:640].astype(np.float32)
    dis = np.float32(1) - np.sqrt((i - yc) * (i - yc) + (j - xc) * (j - xc), dtype=np.float32) / np.sqrt(xc * xc + yc * yc, dtype=np.float32)
    a = np.maximum((dis.astype(np.float64) * 254.0).astype(np.uint8), 2)
    assert np.array_equal(w[..., 3], a)
    w1 = orc.weight_image_8uc4(9, 7, 1)
    assert w1[..., 3].min() == 2


def test_fixed_point_warp_known_answers(orc):
    rng = np.random.RandomState(1)
    src = rng.randint(0, 256, (12, 16, 4)).astype(np.uint8)
    assert np.array_equal(orc.warp_linear_const_8u(src, np.eye(3), 12, 16), src)
    T = np.array([[1, 0, 2], [0, 1, 1], [0, 0, 1.0]])
    out = orc.warp_linear_const_8u(src, T, 14, 20)
    assert np.array_equal(out[1:13, 2:18], src) and (out[0] == 0).all() and (out[:, :2] == 0).all()     # CONSTANT 0 outside
    H = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1.0]])                                                     # half pixel
    row = np.array([[[10], [20], [40], [41]]], np.uint8).repeat(3, axis=0)
    o = orc.warp_linear_const_8u(row, H, 3, 5)[1, :, 0]
    # taps 16384/16384, (a*16384 + b*16384 + 16384) >> 15; left of the image one tap is the constant 0
    assert list(o) == [(0 + 10 * 16384 + 16384) >> 15, (10 * 16384 + 20 * 16384 + 16384) >> 15,
                       (20 * 16384 + 40 * 16384 + 16384) >> 15, (40 * 16384 + 41 * 16384 + 16384) >> 15,
                       (41 * 16384 + 16384) >> 15]


# ---------------------------------------------------------------- GPU parity
def run_pair(pf, orc, cam, poses, frames, typ, n_prepare=None, **opt):
    wl = workloads()
    g = pf.Map2D.create(typ, False, **opt)
    o = orc.OracleMap(single_band=1, weight_type=opt.get("weight_type", 0), scale=opt.get("scale", 1.0))
    prep = poses[:n_prepare] if n_prepare else poses
    assert g.prepare(wl.IDENTITY_PLANE, cam, prep) and o.prepare(wl.IDENTITY_PLANE, cam, prep)
    for f, p in zip(frames, poses):
        assert g.feed(f, p) == o.feed(f[:, :, :3], p)
    g.sync()
    assert g.grid() == o.grid() and g.tiles() == o.tiles() and len(o.tiles()) > 0
    return g, o


@pytest.mark.gpu
@pytest.mark.parametrize("typ", ["TypeCPU", "TypeGPU"])
@pytest.mark.parametrize("weight_type", [0, 1])
def test_single_band_against_oracle(pf, orc, typ, weight_type):
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(9, seed=19 + weight_type)
    frames = [wl.noise_frame(480, 640, 40 + k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, getattr(pf, typ), n_prepare=2, weight_type=weight_type)
    for t in o.tiles():
        assert np.array_equal(g.tile_bgra(*t), o.tile_bgra(*t)), t
    assert g.num_levels == 1
    t = o.tiles()[len(o.tiles()) // 2]
    assert np.array_equal(g.blend_tile(*t), o.tile_bgra(*t)[:, :, :3])
    got = g.blend_tiles([t, (10 ** 6, 0), o.tiles()[0]])                 # the list form: a tile that does not exist keeps the buffer's bytes
    assert np.array_equal(got[0], o.tile_bgra(*t)[:, :, :3]) and (got[1] == 0).all() and np.array_equal(got[2], o.tile_bgra(*o.tiles()[0])[:, :, :3])
    changed, imgs = g.blend_changed()
    assert sorted(changed) == sorted(o.tiles()) and g.blend_changed()[0] == []
    img, org = g.save_to_memory()
    x0, y0 = org
    for (ix, iy) in o.tiles():
        assert np.array_equal(img[(iy - y0) * 256:(iy - y0 + 1) * 256, (ix - x0) * 256:(ix - x0 + 1) * 256], o.tile_bgra(ix, iy)[:, :, :3])


@pytest.mark.gpu
def test_single_band_bgra_input_above_plane_scale(pf, orc):
    wl = workloads()
    cam = [320, 240, 260, 250, 158.5, 121.25]
    poses = jitter_poses(6, seed=5, step=(11.0, 5.0), height=60.0, below=False)
    frames = [np.concatenate([wl.noise_frame(240, 320, k), wl.noise_frame(240, 320, 77 + k)[:, :, :1]], axis=2) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, pf.TypeCPU, scale=0.5)
    for t in o.tiles():
        assert np.array_equal(g.tile_bgra(*t), o.tile_bgra(*t)), t


@pytest.mark.gpu
def test_single_band_shards(pf):
    wl = workloads()
    cam = [640, 480, 500, 500, 320, 240]
    poses = jitter_poses(6, seed=23, step=(30.0, 25.0))
    frames = [wl.smooth_frame(480, 640, k) for k in range(len(poses))]
    maps = []
    for n, r in ((1, 0), (2, 0), (2, 1)):
        m = pf.Map2D.create(pf.TypeCPU, False, shard_count=n, shard_rank=r, shard_block=1)
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses)
        for f, p in zip(frames, poses):
            assert m.feed(f, p)
        m.sync(); maps.append(m)
    ref, a, b = maps
    got = {t: a.tile_bgra(*t) for t in a.tiles()}
    got.update({t: b.tile_bgra(*t) for t in b.tiles()})
    assert sorted(got) == sorted(ref.tiles()) and len(a.tiles()) and len(b.tiles())
    for t in ref.tiles():
        assert np.array_equal(got[t], ref.tile_bgra(*t))


@pytest.mark.gpu
@pytest.mark.parametrize("env", ["PF_SINGLE_OLD", "PF_FORCE_GENERAL"])
def test_single_band_other_kernel_forms(env):
    """The one-pixel-per-thread kernel and the general coordinate/tap forms of k_single2 give the same tiles; both
    are selected by environment switches that are read once per process, hence the child process."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from conftest import load_package\n"
        "from helpers import workloads, jitter_poses\n"
        "import test_single_band as T\n"
        "pf = load_package(); from oracle import orc\n"
        "wl = workloads(); cam = [640, 480, 500, 500, 320, 240]\n"
        "poses = jitter_poses(5, seed=23); frames = [wl.noise_frame(480, 640, 60 + k) for k in range(5)]\n"
        "g, o = T.run_pair(pf, orc, cam, poses, frames, pf.TypeCPU, n_prepare=2)\n"
        "for t in o.tiles():\n"
        "    assert np.array_equal(g.tile_bgra(*t), o.tile_bgra(*t)), t\n"
        "print('single band ok')\n"
    ) % (here, os.path.dirname(here))
    exp = os.path.join(os.path.dirname(here), "pi-slam-fusion_amd", "libpifusion_exp.so")      # the switches exist in the experiments build only (csrc/env.hpp)
    assert os.path.exists(exp), "build the experiments library first (__graft_entry__.build())"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{env: "1", "PF_LIB": exp}), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "single band ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
def test_single_band_cull_leaves_the_tiles_alone(pf, orc, seed):
    """Round 6: the cull of the multi-band path serves Map2DCPU semantics too -- a 64 x 64 cell in which a keyframe cannot raise a stored alpha
    (`if (ele.a < dst.a)`, Map2DCPU.cpp:326-327: 254 wmax <= 254 wlb - 3 with the radial-weight bounds of fusion_map.cpp's cell_out) is left
    out of the launch.  Overlapping sorties at several scales, yaw and tilt jitters and both weight types, the start flown over again: every
    tile byte for byte the oracle's (which renders everything), with cells really culled."""
    wl = workloads()
    rs = np.random.RandomState(7300 + seed)
    cam = [640, 480, 500, 500, 320, 240]
    wt = seed & 1; scale = float(rs.choice([1.0, 1.5, 2.0, 3.0]))
    yaw = float(rs.choice([3.0, 12.0, 45.0, 180.0])); tilt = float(rs.choice([0.5, 4.0, 10.0]))
    nfr = int(rs.randint(12, 22))
    poses = wl.serpentine(cam, float(rs.uniform(60, 140)), nfr, per_row=int(rs.randint(3, 7)), fwd_overlap=float(rs.uniform(0.5, 0.9)),
                          side_overlap=float(rs.uniform(0.3, 0.8)), seed=seed, yaw_jitter_deg=yaw, tilt_jitter_deg=tilt, max_rows=3)
    poses = poses + [list(p) for p in poses[:3]]
    frames = [wl.noise_frame(480, 640, 50 * seed + k) if k % 3 else wl.smooth_frame(480, 640, k) for k in range(len(poses))]
    g, o = run_pair(pf, orc, cam, poses, frames, pf.TypeCPU if seed & 2 else pf.TypeGPU, n_prepare=6, weight_type=wt, scale=scale)
    bad = [t for t in o.tiles() if not np.array_equal(g.tile_bgra(*t), o.tile_bgra(*t))]
    assert bad == [], (seed, bad[:4])
    assert g.culled_tiles() + g.culled_cells() > 0, "the sortie overlaps: something must have been culled"
    # ... and with the cull off the same tiles once more (pf_set_cull)
    if seed < 2:
        g2 = pf.Map2D.create(pf.TypeCPU, False, weight_type=wt, scale=scale)
        g2.set_cull(False)
        assert g2.prepare(wl.IDENTITY_PLANE, cam, poses[:6])
        for f, p in zip(frames, poses):
            g2.feed(f, p)
        g2.sync()
        assert g2.culled_tiles() + g2.culled_cells() == 0
        for t in o.tiles():
            assert np.array_equal(g2.tile_bgra(*t), o.tile_bgra(*t)), t
