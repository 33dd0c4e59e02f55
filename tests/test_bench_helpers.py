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


def test_roofline_numerator_follows_the_canvas_share_that_ran():
    """roofline.frac counts the bytes of what the launches' blocks processed (alg_bytes_run), not SURVEY 8d's bytes of every canvas tile
    (VERDICT r04 item 1): with the frame read once (36 MB) and 377.8 MB of tile bytes, the numerator must scale with the run share."""
    S, T = 36e6, 377.8e6
    recs = []
    for share in (1.0, 0.687, 0.5):
        p = {"ms": 0.12009 * 5, "launches": 5, "alg_bytes": (S + T) * 5, "alg_bytes_run": (S + T * share) * 5}
        recs.append(bench.roofline_record("level0_fused", p, "f32", 4, pmc_ok=False, window=bench.window_key(20, 5, 15)))
    full, driver, half = recs
    assert full["frac"] == full["frac_full_canvas"]
    assert driver["frac_full_canvas"] == full["frac_full_canvas"]                # the old number does not move with the cull ...
    assert abs(driver["frac"] - (S + T * 0.687) / (S + T) * full["frac"]) < 2e-4  # ... the new one does
    assert abs(driver["frac"] - 0.3076) < 2e-3                                  # VERDICT r04's recomputation of the r04 driver window: 0.31
    assert half["alg_bytes_run_per_launch"] == round(S + T * 0.5) and half["frac"] < driver["frac"] < full["frac"]
    assert driver["window"] == "k20_w5_pre15" and driver["traffic"] is None and driver["frac_delivered"] is None


def test_pmc_records_are_keyed_by_window(tmp_path, monkeypatch):
    import json
    d = tmp_path / "profiles"; d.mkdir()
    (d / "pmc_traffic.json").write_text(json.dumps({
        "_meta": {"git_sha": "x", "kernels_sha": "y"},
        "f32": {"level0_fused": {"windows": {"k200_w20_pre0": {"traffic": 350, "valu_insts": 7, "launches": 200},
                                             "k20_w5_pre15": {"traffic": 400, "launches": 20}}}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernels_sha", lambda: "y")
    assert bench.pmc_record("f32", "level0_fused", "k20_w5_pre15")["traffic"] == 400
    assert bench.pmc_record("f32", "level0_fused", "k200_w20_pre0")["current"] is True
    assert bench.pmc_record("f32", "level0_fused", "k20_w5_pre15_nocull") is None      # no pass of its own: no number
    p = {"ms": 0.1 * 20, "launches": 20, "alg_bytes": 4e8 * 20, "alg_bytes_run": 3e8 * 20}
    r = bench.roofline_record("level0_fused", p, "f32", 4, True, "k20_w5_pre15")
    assert r["traffic"] == 400 and r["frac_delivered"] is not None and r["pmc_build"]["launches_averaged"] == 20
    assert bench.roofline_record("level0_fused", p, "f32", 4, True, "k20_w5_pre15_nocull")["traffic"] is None


def test_delivered_over_alg_is_in_the_roofline_record(tmp_path, monkeypatch):
    """The wasted-traffic ratio (PMC bytes moved per algorithmic byte processed) is part of the line, not only derivable from it
    (VERDICT r05 item 5): 331.9 / 254.6 MB of round 5's driver window = 1.30."""
    import json
    d = tmp_path / "profiles"; d.mkdir()
    (d / "pmc_traffic.json").write_text(json.dumps({
        "_meta": {"git_sha": "x", "kernels_sha": "y"},
        "f32": {"level0_fused": {"windows": {"k20_w5_pre15": {"traffic": 331900000, "launches": 20}}}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernels_sha", lambda: "y")
    p = {"ms": 0.1076 * 20, "launches": 20, "alg_bytes": 4.2e8 * 20, "alg_bytes_run": 254.59e6 * 20}
    r = bench.roofline_record("level0_fused", p, "f32", 4, True, "k20_w5_pre15")
    assert abs(r["delivered_over_alg"] - 331.9 / 254.59) < 2e-3
    assert bench.roofline_record("level0_fused", p, "f32", 4, False, "k20_w5_pre15")["delivered_over_alg"] is None


def test_kernels_sha_covers_the_host_engine_and_the_collapse_kernel():
    """The cull, the need bitmaps and the launch arguments are built in fusion_map.cpp: a change there changes the bytes a launch
    touches, so the PMC numbers' build id must move with it."""
    assert {"fusion_map.cpp", "collapse_fused.hip", "kernels.hip"} <= set(bench.KERNEL_SOURCES)


def test_output_side_record_fields(monkeypatch):
    """bench.output_side_rate against a stand-in package: the record's arithmetic (tiles/s, GB/s, roofline fraction of the launch)."""
    import numpy as np
    import types

    class FakeMap:
        def __init__(self): self.prof = {}
        def prepare(self, *a): return True
        def feed_device(self, *a): return True
        def sync(self): return True
        def tiles(self): return [(x, y) for y in range(4) for x in range(5)]
        def profile_reset(self): self.prof = {}
        def profile_enable(self, v): pass
        def close(self): pass
        def blend_tiles(self, tiles, out=None):
            out[:] = 7
            r = self.prof.setdefault("blend_fused", {"ms": 0.0, "launches": 0, "alg_bytes": 0.0})
            r["ms"] += 0.01; r["launches"] += 1; r["alg_bytes"] += len(tiles) * 1.5e6
            return out
        def save_to_memory(self, alloc=None):
            a = alloc((1024, 1280, 3))
            r = self.prof.setdefault("save_fused", {"ms": 0.0, "launches": 0, "alg_bytes": 0.0})
            r["ms"] += 0.02; r["launches"] += 1; r["alg_bytes"] += 4e7
            return a, (0, 0)
        def profile_read(self): return self.prof

    fake = types.SimpleNamespace(TypeMultiBandCPU=3, host_array=lambda shape: np.zeros(shape, np.uint8),
                                 Map2D=types.SimpleNamespace(create=lambda *a, **k: FakeMap()))
    wl = types.SimpleNamespace(IDENTITY_PLANE=[0, 0, 0, 0, 0, 0, 1])
    dev = [types.SimpleNamespace(data_ptr=lambda: 0)]
    r = bench.output_side_rate(fake, wl, [[0, 0, -100, 0, 0, 0, 1]] * 4, None, dev, 1, frames=3, reps=2)
    assert r["tiles"] == 20 and r["kernel"] == "blend_fused" and r["buffers_equal"] is True
    assert r["kernel_ms"] == 0.01 and r["alg_bytes_per_launch"] == 30000000
    assert abs(r["kernel_alg_GBps"] - 3000.0) < 1 and abs(r["frac"] - 3000.0 / bench.HBM_PEAK_GBS) < 1e-3
    assert r["kernel_tiles_per_s"] == 2000000 and r["tiles_per_s"] > 0 and r["d2h_GBps"] > 0
    assert r["save"]["kernel"] == "save_fused" and r["save"]["mosaic"] == [1024, 1280] and abs(r["save"]["kernel_alg_GBps"] - 2000.0) < 1
