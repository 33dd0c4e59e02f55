"""End-to-end golden digests (tests/golden/e2e_digests.json, made by make_golden.py with the
oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them with no
oracle in the process."""
import importlib.util
import json
import os

import pytest

from helpers import map_digest, sha, workloads

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "e2e_digests.json")))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)
CASES = dict(mk.cases())


@pytest.mark.parametrize("key", sorted(GOLD))
def test_oracle_reproduces_golden(key):
    name, dt = key.split("/")
    assert json.loads(json.dumps(mk.run_case(CASES[name], 1 if dt == "f32" else 0))) == GOLD[key]


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(GOLD))
def test_gpu_reproduces_golden(pf, key):
    wl = workloads()
    name, dt = key.split("/")
    c, g = CASES[name], GOLD[key]
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1 if dt == "f32" else 0)
    prep = c["poses"][:c["n_prepare"]] if c["n_prepare"] else c["poses"]
    assert m.prepare(wl.IDENTITY_PLANE, c["cam"], prep)
    for f, p in zip(mk.frames_of(c["frames"], len(c["poses"])), c["poses"]):
        assert m.feed(f, p)
    assert m.sync()
    dims, geo = m.grid()
    assert [dims, geo] == g["grid"]
    assert map_digest(m) == g["tiles"]
    img, org = m.save_to_memory()
    assert [sha(img), list(org), list(img.shape)] == g["save"]
    for k, h in g["blend"].items():
        ix, iy = map(int, k.split(","))
        assert sha(m.blend_tile_raw(ix, iy)) == h
