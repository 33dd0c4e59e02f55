"""Synthetic workloads of BASELINE.md section 4 / SURVEY.md 8d (no datasets in the
container).  Pure host-side pose/frame generators shared by tests and bench.py."""
import math

import numpy as np

IDENTITY_PLANE = [0, 0, 0, 0, 0, 0, 1]


def splitmix64_bytes(seed, n):
    """n pseudo-random bytes from splitmix64(seed) (cfg-1 pixel content)."""
    cnt = (n + 7) // 8
    idx = np.arange(1, cnt + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8)[:n].copy()


def noise_frame(rows, cols, k):
    return splitmix64_bytes(1000 + k, rows * cols * 3).reshape(rows, cols, 3)


def smooth_frame(rows, cols, k):
    y, x, c = np.meshgrid(np.arange(rows), np.arange(cols), np.arange(3), indexing="ij")
    return ((x * 3 + y * 5 + k * 17 + c * 40) & 255).astype(np.uint8)


def quat_mul(a, b):
    x, y, z, w = a
    return [w * b[0] + x * b[3] + y * b[2] - z * b[1],
            w * b[1] + y * b[3] + z * b[0] - x * b[2],
            w * b[2] + z * b[3] + x * b[1] - y * b[0],
            w * b[3] - x * b[0] - y * b[1] - z * b[2]]


def quat_axis(axis, angle):
    s = math.sin(angle / 2)
    return [axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle / 2)]


def cfg1(n=10, step=10.0):
    """cfg-1 (plumbing): 640x480, identity rotation, t=(k*step, 0, -100)."""
    cam = [640, 480, 500, 500, 320, 240]
    poses = [[k * step, 0.0, -100.0, 0, 0, 0, 1] for k in range(n)]
    return cam, poses


def serpentine(cam, height, n_frames, per_row=20, fwd_overlap=0.8, side_overlap=0.6, seed=42,
               yaw_jitter_deg=5.0, tilt_jitter_deg=2.0, origin=(0.0, 0.0), max_rows=None):
    """cfg-2 trajectory: camera below the plane (z=-H) looking along +z, image y is
    the flight direction; 80 % forward / 60 % side overlap, yaw +-5 deg, roll/pitch +-2 deg.
    max_rows: after that many rows the sortie is flown again from its first row (bounded area
    and tile memory for long runs; every frame is still a full render)."""
    w, h, fx, fy = cam[0], cam[1], cam[2], cam[3]
    foot_x, foot_y = w * height / fx, h * height / fy
    dy, dx = foot_y * (1 - fwd_overlap), foot_x * (1 - side_overlap)
    rng = np.random.RandomState(seed)      # mt19937
    poses = []
    for k in range(n_frames):
        row, col = divmod(k, per_row)
        if max_rows:
            row %= max_rows
        if row & 1:
            col = per_row - 1 - col
        yaw = math.radians(rng.uniform(-yaw_jitter_deg, yaw_jitter_deg))
        roll = math.radians(rng.uniform(-tilt_jitter_deg, tilt_jitter_deg))
        pitch = math.radians(rng.uniform(-tilt_jitter_deg, tilt_jitter_deg))
        q = quat_mul(quat_axis((0, 0, 1), yaw), quat_mul(quat_axis((0, 1, 0), pitch), quat_axis((1, 0, 0), roll)))
        poses.append([origin[0] + row * dx, origin[1] + col * dy, -height] + q)
    return poses


def cfg2(n_frames=220, scale=1.0):
    cam = [4000, 3000, 3000, 3000, 2000, 1500]
    return cam, serpentine(cam, 100.0, n_frames)
