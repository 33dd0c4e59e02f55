import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from conftest import load_package
pf = load_package()
from oracle import orc
wl = __import__('importlib').import_module('pi_slam_fusion_amd.workloads')
for ff in (0, 1):
    cam, poses = wl.cfg1()
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff)
    o = orc.OracleMap(force_float=ff)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
    print(m.grid(), o.grid())
    for k, p in enumerate(poses):
        img = wl.noise_frame(480, 640, k)
        a = m.feed(img, p); b = o.feed(img, p)
        assert a and b, (a, b)
    m.sync()
    print(len(m.tiles()), len(o.tiles()), m.tiles() == o.tiles())
    bad = 0
    for (ix, iy) in o.tiles():
        for lv in range(o.num_levels):
            gl, gw = m.tile_level(ix, iy, lv); ol, ow = o.tile_level(ix, iy, lv)
            if not np.array_equal(gw, ow) or not np.array_equal(gl, ol):
                bad += 1
                if bad < 6:
                    d = (gl != ol).any(axis=2); dw = gw != ow
                    print("MISMATCH tile", ix, iy, "lvl", lv, "lap px", d.sum(), "w px", dw.sum(),
                          "maxdiff", np.abs(gl.astype(np.float64) - ol).max(), np.abs(gw - ow).max())
    print("force_float", ff, "mismatching tile-levels:", bad)
    t = o.tiles()[len(o.tiles()) // 2]
    print("blend eq", np.array_equal(m.blend_tile_raw(*t), o.blend_tile_raw(*t)), np.array_equal(m.blend_tile(*t), o.blend_tile(*t)))
    t0 = o.tiles()[0]
    print("blend(self) eq", np.array_equal(m.blend_tile_raw(*t0), o.blend_tile_raw(*t0)))
    sg, og = m.save_to_memory(), o.save()
    print("save eq", sg[1] == og[1], np.array_equal(sg[0], og[0]))
    print(m.stats())
