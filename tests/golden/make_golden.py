"""Generates tests/golden/e2e_digests.json with the build's own oracle (the reference holds no
golden vectors for this path and cannot be built here: SURVEY 8c).  Each case: seeds -> per
tile / per level sha1 of the Laplacian and weight bytes, the save() mosaic sha1 and two blended
tiles.  Run from the repo root:  python tests/golden/make_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import jitter_poses, map_digest, sha, workloads  # noqa: E402
from oracle import orc  # noqa: E402

wl = workloads()


def cases():
    cam1, poses1 = wl.cfg1()
    yield "cfg1_noise", dict(cam=cam1, poses=poses1, frames=("noise", 480, 640, 0), n_prepare=None)
    yield "cfg1_smooth", dict(cam=cam1, poses=poses1, frames=("smooth", 480, 640, 0), n_prepare=None)
    cam = [640, 480, 500, 500, 320, 240]
    yield "perspective_spread", dict(cam=cam, poses=jitter_poses(9, seed=12), frames=("noise", 480, 640, 100), n_prepare=2)
    yield "overlap4", dict(cam=cam, poses=jitter_poses(4, seed=77, step=(6.0, 4.0)), frames=("noise", 480, 640, 700), n_prepare=None)


def frames_of(spec, n):
    kind, r, c, base = spec
    gen = wl.noise_frame if kind == "noise" else wl.smooth_frame
    return [gen(r, c, base + k) for k in range(n)]


def run_case(c, force_float):
    o = orc.OracleMap(force_float=force_float)
    prep = c["poses"][:c["n_prepare"]] if c["n_prepare"] else c["poses"]
    assert o.prepare(wl.IDENTITY_PLANE, c["cam"], prep)
    for f, p in zip(frames_of(c["frames"], len(c["poses"])), c["poses"]):
        assert o.feed(f, p)
    tiles = o.tiles()
    mid = tiles[len(tiles) // 2]
    img, org = o.save()
    return {"grid": o.grid(), "tiles": map_digest(o), "save": [sha(img), list(org), list(img.shape)],
            "blend": {"%d,%d" % t: sha(o.blend_tile_raw(*t)) for t in (tiles[0], mid)}}


if __name__ == "__main__":
    out = {}
    for name, c in cases():
        for ff in (0, 1):
            out["%s/%s" % (name, "f32" if ff else "int16")] = run_case(c, ff)
    path = os.path.join(HERE, "e2e_digests.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")
