"""bench.py's host-side helpers with the driver's own command line (`--steps 20 --warmup 5`) and the
shortest run (`--steps 1 --warmup 0`): the CPU legs must never index past the sortie (round-1 crash:
VERDICT r01 item 1), and a failing leg must become a record, not an exception."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pose_indexing_wraps():
    for steps, warm in ((20, 5), (1, 0), (200, 20)):
        poses = list(range(steps + warm))
        for n in range(0, 130):
            assert bench.pose_at(poses, n) == n % (steps + warm)


def test_event_every_scales_with_steps():
    assert bench.event_every_for(20) == 4          # 5 timed launches under the driver's --steps 20
    assert bench.event_every_for(1) == 1
    assert bench.event_every_for(200) == 41         # not a multiple of the 20 keyframes of a flight line
    assert bench.event_every_for(200, 16) == 16 and bench.event_every_for(5, 0) == 0


def test_guarded_turns_failures_into_records():
    assert bench.guarded(lambda: 3) == 3
    r = bench.guarded(lambda: [][1])
    assert "IndexError" in r["error"]
    r = bench.guarded(lambda: (_ for _ in ()).throw(SystemExit("child died")))
    assert "SystemExit" in r["error"]


def test_cpu_legs_run_past_the_end_of_a_short_sortie(orc):
    """steps=1, warmup=0: one pose; the CPU baseline asks for three frames."""
    bench.load_package()
    wl = importlib.import_module("pi_slam_fusion_amd.workloads")
    old = bench.CAM
    bench.CAM = [320, 240, 240, 240, 160, 120]          # same geometry as cfg-A at 1/12.5 of the edge: seconds -> ms
    try:
        poses = wl.serpentine(bench.CAM, bench.HEIGHT, 1)
        frames = [wl.noise_frame(240, 320, k) for k in range(2)]
        r = bench.cpu_baseline(wl, poses, poses[:20], frames, 1, budget_s=30.0, max_frames=3)
        assert r["cores"] == 1 and r["kind"] == "port" and r["value"] > 0 and "first 3 frames" in r["sample"]
        poses = wl.serpentine(bench.CAM, bench.HEIGHT, 25)          # the driver's 20 + 5
        r = bench.cpu_baseline(wl, poses, poses[:20], frames, 0, budget_s=30.0, max_frames=40)
        assert "first 40 frames" in r["sample"]
    finally:
        bench.CAM = old


def test_pmc_record_carries_its_build():
    r = bench.pmc_record("f32", "level0_fused")
    assert r is None or ({"traffic", "git_sha", "kernels_sha", "current"} <= set(r) and isinstance(r["current"], bool))
    assert len(bench.kernels_sha()) == 16
