"""The warp's source addressing (pi-slam-fusion_amd/csrc/warp_index.hpp -- the very code the kernel compiles) checked
exhaustively on the host: every coordinate in a wide ring around frames of many shapes, every path; no load outside the
bytes the caller handed over, taps == cv::borderInterpolate(BORDER_REFLECT).  See tests/cpp/warp_index_check.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_warp_addressing_in_bounds_and_right_taps(tmp_path):
    exe = str(tmp_path / "warp_index_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", os.path.join(ROOT, "tests", "cpp", "warp_index_check.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert b"in bounds with the right taps" in r.stdout
