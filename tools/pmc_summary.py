#!/usr/bin/env python3
"""Summarise rocprofv3 output for profiles/.

  kernel stats : tools/pmc_summary.py stats <dir with *_kernel_stats.csv> > profiles/<name>.md
  HBM traffic  : tools/pmc_summary.py traffic <fetch dir> <write dir> <dtype> [profiles/pmc_traffic.json] [<SQ_INSTS_VALU dir>]
                 with PF_WINDOW="K W PRE [nocull]" in the environment: only the K TIMED launches of `bench.py --steps K --warmup W`
                 (PRE keyframes flown before the warm-up) are averaged, and the record goes under windows[bench.window_key(...)]
  SQ counters  : tools/pmc_summary.py counters <pmc dir> ...

The JSON records the build it was taken at (_meta.git_sha, _meta.kernels_sha = bench.kernels_sha()); bench.py
reports `roofline.pmc_build.current` = false when the device code has changed since.

Traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes; both are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced read, so it is doubled; WRITE_SIZE is taken as is.  Per launch = mean over
the launches of that kernel in the pass (same mix of pyramid levels as bench.py's
per-launch `achieved`)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

SHORT = [("k_warp", "warp"), ("k_pyrdown<float, float, 1>", "pyrdown_w"), ("k_pyrdown", "pyrdown_img"),
         ("k_lap_select", "lap_select"), ("k_level", "level"), ("k_blend_gather", "blend_gather"), ("k_collapse", "collapse"),
         ("k_blend_finish", "blend_finish"), ("k_mosaic_gather", "mosaic_gather"), ("k_save_finish", "save_finish"),
         ("k_single", "single_band")]


def pipelined(name):
    """the one-launch-per-keyframe kernels: the block form (k_levels) and the rolling-strip form (k_strips, r04)"""
    return "k_levels<" in name or "k_strips<" in name


def short(name):
    if pipelined(name):          # pipelined launch: level 0 of a frame + upper levels of earlier frames
        return "level0_fused"
    m = re.search(r"k_collapse_fused<(\w+), (\w+)>", name)
    if m:                        # the fused output-side kernel: Ele::blend (MOSAIC = false) / save (true)
        return "save_fused" if m.group(2) == "true" else "blend_fused"
    m = re.search(r"k_level3?<(\w+), (\w+)", name)
    if m:
        return "level0_fused" if m.group(2) == "true" else "level_fused"
    for k, s in SHORT:
        if k in name:
            return s
    return None


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not f:
        raise SystemExit("no %s under %s" % (pat, d))
    return f[0]


def stats(d):
    rows = list(csv.DictReader(open(find(d, "*_kernel_stats.csv"))))
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", ""))[:70]
        print("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s |" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
              float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    full_launches(d)


def full_launches(d):
    """k_levels only: mean duration of the full-size launches (one per keyframe) from the per-dispatch trace,
    i.e. without the short flush launches before a sync -- the launches bench.py puts its events around"""
    try:
        rows = [r for r in csv.DictReader(open(find(d, "*_kernel_trace.csv"))) if pipelined(r["Kernel_Name"])]
    except SystemExit:
        return
    if not rows:
        return
    size = lambda r: int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
    full = max(size(r) for r in rows)
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if size(r) * 2 > full]
    print()
    print("pipelined level kernel, full-size launches only (%d of %d dispatches): avg %.2f us, min %.2f us, max %.2f us" %
          (len(dur), len(rows), sum(dur) / len(dur), min(dur), max(dur)))


def window():
    """PF_WINDOW="K W PRE [nocull]": (K, W, PRE, nocull) or None"""
    w = os.environ.get("PF_WINDOW", "").split()
    if len(w) < 3:
        return None
    return int(w[0]), int(w[1]), int(w[2]), len(w) > 3 and w[3] == "nocull"


def counter_rows(d):
    """rows of a --pmc pass; the pipelined k_levels launches are kept only at full size (a keyframe's launch), not
    the short flush launches before a sync that carry upper levels only -- the same launches bench.py times.
    With PF_WINDOW: only the full-size launches PRE + W .. PRE + W + K - 1 of the process (bench.py --no-cpu feeds one map: PRE keyframes,
    W warm-up keyframes, K timed ones, one full-size launch each)."""
    rows = [r for r in csv.DictReader(open(find(d, "*_counter_collection.csv"))) if short(r["Kernel_Name"])]
    win = window()
    if win:
        # by position: PRE keyframe launches (+ L-1 = 4 short flush launches at the sync behind them), W launches + 4, K launches + 4
        K, W, PRE, _ = win
        nf = int(os.environ.get("PF_FLUSH_LAUNCHES", "4"))
        ids = sorted({int(r["Dispatch_Id"]) for r in rows if pipelined(r["Kernel_Name"])})
        if len(ids) != PRE + W + K + nf * (2 + (PRE > 0)):
            raise SystemExit("window %s: %d pipelined launches in the pass, %d expected" % (win, len(ids), PRE + W + K + nf * (2 + (PRE > 0))))
        start = PRE + (nf if PRE else 0) + W + nf
        keep = set(ids[start:start + K])
        return [r for r in rows if not pipelined(r["Kernel_Name"]) or int(r["Dispatch_Id"]) in keep]
    full = max([int(r["Grid_Size"]) for r in rows if pipelined(r["Kernel_Name"])] or [0])
    return [r for r in rows if not pipelined(r["Kernel_Name"]) or int(r["Grid_Size"]) * 2 > full]


def counter_means(d, counter):
    acc = defaultdict(list)
    for r in counter_rows(d):
        if r["Counter_Name"] != counter:
            continue
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def build_meta():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    try:
        import subprocess
        sha = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        sha = os.environ.get("PF_GIT_SHA")          # the GPU box has no .git: pass the sha in
    return {"git_sha": sha, "kernels_sha": bench.kernels_sha(),
            "how": "tools/profile_round.sh + tools/pmc_summary.py (separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)"}


def traffic(fd, wd, dtype, out, vd=None):
    f, nf = counter_means(fd, "FETCH_SIZE")
    w, _ = counter_means(wd, "WRITE_SIZE")
    v = counter_means(vd, "SQ_INSTS_VALU")[0] if vd else {}
    res = {}
    for k in sorted(set(f) | set(w)):
        res[k] = {"traffic": int(round(2 * f.get(k, 0.0) * 1024 + w.get(k, 0.0) * 1024))}
        if k in v:
            res[k]["valu_insts"] = int(round(v[k]))
        print("%-14s launches %5d  fetch(x2) %10.1f KiB  write %10.1f KiB  -> %d B/launch" %
              (k, nf.get(k, 0), 2 * f.get(k, 0.0), w.get(k, 0.0), res[k]["traffic"]), file=sys.stderr)
    if out:
        cur = json.load(open(out)) if os.path.exists(out) else {}
        meta = build_meta()
        if cur.get("_meta", {}).get("kernels_sha") != meta["kernels_sha"]:
            cur = {}                                  # numbers of another build do not mix
        cur["_meta"] = meta
        win = window()
        if win:
            import bench
            key = bench.window_key(*win)
            for k in res:
                res[k]["launches"] = nf.get(k, 0)
                cur.setdefault(dtype, {}).setdefault(k, {}).setdefault("windows", {})[key] = res[k]
        else:
            cur[dtype] = res
        json.dump(cur, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res))


def counters(dirs):
    """mean per launch of every counter found in the given --pmc output dirs, per kernel"""
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for r in counter_rows(d):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print("## %s" % k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("| %s | %d launches | %.4g per launch |" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "counters":
        counters(sys.argv[2:])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None,
                sys.argv[6] if len(sys.argv) > 6 else None)
