#!/usr/bin/env python3
"""Host-only model of the cull with a LOOKAHEAD window (no GPU, no library): how much of a keyframe's canvas still has to be rendered when the
weight bounds of the next D keyframes are known before the keyframe is rendered.

bench.py's cfg-A sortie, the cull's own rule per 64 x 64 cell of a tile dilated by 64 px (fusion_map.cpp, cell_out): a cell is out for keyframe k
when the largest radial weight k can have on the dilated cell (+ margins) is below wlb = max over the keyframes admitted so far of the smallest
weight each has there (0 unless the dilated cell maps wholly inside that keyframe).  D = 0 is the product of rounds 4-6.
   fresh = "all"    a keyframe renders every cell of a tile nobody rendered before (rounds 4-6)
   fresh = "tile"   ... unless all 16 cells are out: the tile then stays fresh for the next keyframe
   fresh = "free"   cells of fresh tiles are culled like any other (needs tile slots whose weights start at 0)
Prints rendered share of the canvases and the share within two cells (128 px >= the 94 px reach) of a rendered cell.
   python3 tools/lookahead_model.py [frames]"""
import importlib.util
import math
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
spec = importlib.util.spec_from_file_location("workloads", os.path.join(R, "pi-slam-fusion_amd", "workloads.py"))
wl = importlib.util.module_from_spec(spec); spec.loader.exec_module(wl)

CAM = [4000, 3000, 3000, 3000, 2000, 1500]
H = 100.0
LP = H / CAM[2]                # metres per canvas pixel (Map2D.Scale = 1)
CELL = 64 * LP
DMAX = math.hypot(CAM[0] // 2, CAM[1] // 2)
MPX, MW = 2.0, 1e-5


def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Frame:
    def __init__(self, pose):
        self.t = np.array(pose[:3]); self.R = rot(pose[3:])
        corners = np.array([[0, 0], [CAM[0], 0], [0, CAM[1]], [CAM[0], CAM[1]]], float)
        g = np.array([self.ground(u, v) for u, v in corners])
        self.x0, self.y0 = g.min(0); self.x1, self.y1 = g.max(0)
        self.pc = self.ground(CAM[4], CAM[5])

    def ground(self, u, v):
        ray = self.R @ np.array([(u - CAM[4]) / CAM[2], (v - CAM[5]) / CAM[3], 1.0])
        s = -self.t[2] / ray[2]
        return (self.t + s * ray)[:2]

    def image(self, gx, gy):
        p = np.stack([gx - self.t[0], gy - self.t[1], np.full_like(gx, -self.t[2])], -1) @ self.R      # R^T (g - t)
        return CAM[2] * p[..., 0] / p[..., 2] + CAM[4], CAM[3] * p[..., 1] / p[..., 2] + CAM[5]

    def bounds(self, ci0, cj0, ci1, cj1):
        """ub, lb per cell of the cell range [ci0, ci1) x [cj0, cj1)"""
        i = np.arange(ci0, ci1)[None, :]; j = np.arange(cj0, cj1)[:, None]
        xa, xb = (i - 1) * CELL, (i + 2) * CELL; ya, yb = (j - 1) * CELL, (j + 2) * CELL
        xa, xb, ya, yb = [np.broadcast_to(a, (cj1 - cj0, ci1 - ci0)).astype(float) for a in (xa, xb, ya, yb)]
        far = np.zeros_like(xa); inside = np.ones_like(xa, bool)
        for gx, gy in ((xa, ya), (xb, ya), (xa, yb), (xb, yb)):
            u, v = self.image(gx, gy)
            far = np.maximum(far, np.hypot(u - CAM[4], v - CAM[5]))
            inside &= (u >= 1) & (u <= CAM[0] - 2) & (v >= 1) & (v <= CAM[1] - 2)
        lb = np.where(inside, 1 - (far + MPX) / DMAX - MW, 0.0)
        lb = np.where(lb > 2e-5, lb, 0.0)
        nx, ny = np.clip(self.pc[0], xa, xb), np.clip(self.pc[1], ya, yb)
        u, v = self.image(nx, ny)
        near = np.hypot(u - CAM[4], v - CAM[5])
        ub = 1 - np.maximum(near - MPX, 0) / DMAX + MW
        return ub, lb


def dilate(m, r):
    o = m.copy()
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            s = np.zeros_like(m)
            ys = slice(max(dy, 0), m.shape[0] + min(dy, 0)); yd = slice(max(-dy, 0), m.shape[0] + min(-dy, 0))
            xs = slice(max(dx, 0), m.shape[1] + min(dx, 0)); xd = slice(max(-dx, 0), m.shape[1] + min(-dx, 0))
            s[yd, xd] = m[ys, xs]; o |= s
    return o


def run(n, D, fresh_mode, first=20):
    poses = wl.serpentine(CAM, H, n, max_rows=16)
    frames = [Frame(p) for p in poses]
    ox = min(f.x0 for f in frames) - 20; oy = min(f.y0 for f in frames) - 20
    NX = int((max(f.x1 for f in frames) - ox) / CELL) + 16; NY = int((max(f.y1 for f in frames) - oy) / CELL) + 16
    wlb = np.full((NY, NX), -1.0); fresh = np.ones((NY // 4 + 1, NX // 4 + 1), bool)
    rng, B = [], []
    for f in frames:
        ti0, tj0 = int(math.floor((f.x0 - ox) / (4 * CELL))), int(math.floor((f.y0 - oy) / (4 * CELL)))
        ti1, tj1 = int(math.ceil((f.x1 - ox) / (4 * CELL))), int(math.ceil((f.y1 - oy) / (4 * CELL)))
        rng.append((ti0, tj0, ti1, tj1))
        g = Frame.__new__(Frame); g.__dict__ = dict(f.__dict__); g.t = f.t - np.array([ox, oy, 0]); g.pc = f.pc - np.array([ox, oy])
        B.append(g.bounds(4 * ti0, 4 * tj0, 4 * ti1, 4 * tj1))
    admitted = 0
    tot = rend = run1 = run2 = 0
    for k in range(n):
        while admitted < min(n, k + D + 1):
            ti0, tj0, ti1, tj1 = rng[admitted]
            sl = (slice(4 * tj0, 4 * tj1), slice(4 * ti0, 4 * ti1))
            wlb[sl] = np.maximum(wlb[sl], B[admitted][1])
            admitted += 1
        ti0, tj0, ti1, tj1 = rng[k]
        sl = (slice(4 * tj0, 4 * tj1), slice(4 * ti0, 4 * ti1))
        ub, _ = B[k]
        out = (wlb[sl] > 2e-5) & (ub < wlb[sl])
        fr = np.kron(fresh[tj0:tj1, ti0:ti1], np.ones((4, 4), bool))
        if fresh_mode == "all":
            out &= ~fr
        elif fresh_mode == "tile":
            allout = out.reshape(tj1 - tj0, 4, ti1 - ti0, 4).all((1, 3))
            out &= ~(fr & ~np.kron(allout, np.ones((4, 4), bool)))
        r = ~out
        tile_rendered = r.reshape(tj1 - tj0, 4, ti1 - ti0, 4).any((1, 3))
        fresh[tj0:tj1, ti0:ti1] &= ~tile_rendered
        if k >= first:
            tot += r.size; rend += r.sum(); run1 += dilate(r, 1).sum(); run2 += dilate(r, 2).sum()
    return rend / tot, run1 / tot, run2 / tot


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
    print("| lookahead D | fresh tiles | rendered share | within 1 cell | within 2 cells |\n|---|---|---|---|---|")
    for mode in ("all", "tile", "free"):
        for D in (0, 1, 2, 4, 8, 20, 40):
            a, b, c = run(n, D, mode)
            print("| %d | %s | %.3f | %.3f | %.3f |" % (D, mode, a, b, c), flush=True)
