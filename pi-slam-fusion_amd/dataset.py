"""DroneMap / NPU dataset wire format, as read by the reference's file driver
(backup/map2dfusion.cpp obtainFrame / testMap2D, gui/IO/DatasetNPUDroneMap.cpp:178-213):

    <dir>/config.cfg        svar lines: `Plane = x y z qx qy qz qw`,
                            `Camera.Paraments = [w h fx fy cx cy]`, `GPS.Origin = lon lat alt`
    <dir>/trajectory.txt    one keyframe per line: `name x y z qx qy qz qw`
    <dir>/rgb/<name>.jpg    the frames (BGR after decode, as cv::imread gives them)
"""
import os
import re

import numpy as np


def parse_config(path):
    """Subset of the svar grammar the dataset files use: `key = value`, `key ?= value` (assign only if unset), comments
    (GSLAM/core/Svar.h; pinned by tests/golden/svar_vectors.json, made with the reference's own parser)."""
    out = {}
    for line in open(path):
        line = line.split("//")[0].split("#")[0].strip()
        m = re.match(r"^([\w.]+)\s*(\??=)\s*(.*)$", line)
        if not m:
            continue
        key, weak, val = m.group(1), m.group(2) == "?=", m.group(3).strip()
        if weak and key in out:
            continue
        nums = re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?", val)
        out[key] = [float(x) for x in nums] if nums and re.fullmatch(r"[\[\]\s,\d.eE+-]+", val) else val
    return out


def read_trajectory(path):
    frames = []
    for line in open(path):
        p = line.split()
        if len(p) >= 8:
            frames.append((p[0], [float(x) for x in p[1:8]]))
    return frames


class DroneMapDataset:
    def __init__(self, datapath):
        self.path = datapath
        self.cfg = parse_config(os.path.join(datapath, "config.cfg"))
        self.plane = self.cfg.get("Plane", [0, 0, 0, 0, 0, 0, 1])        # pi::SE3d() default
        self.camera = self.cfg.get("Camera.Paraments")
        if not self.camera or len(self.camera) != 6:
            raise ValueError("Invalid camera parameters!")              # backup/map2dfusion.cpp testMap2D
        self.gps_origin = self.cfg.get("GPS.Origin")
        self.frames = read_trajectory(os.path.join(datapath, "trajectory.txt"))

    def __len__(self):
        return len(self.frames)

    def load(self, k, encoded=False):
        """(BGR uint8 image, pose7) of keyframe k.  encoded=True: a .jpg frame comes back as the file's bytes, for Map2D.feed to decode
        on the GPU (pf_feed_jpeg)."""
        name, pose = self.frames[k]
        base = os.path.join(self.path, "rgb", name)
        if encoded and os.path.exists(base + ".jpg"):
            return open(base + ".jpg", "rb").read(), pose
        for ext in (".jpg", ".png", ".ppm", ".npy"):
            if os.path.exists(base + ext):
                if ext == ".npy":
                    return np.load(base + ext), pose
                from . import read_image                                   # the library's own readers (csrc/jpeg_decode.cpp, png_decode.cpp)
                return read_image(base + ext), pose
        raise FileNotFoundError(base + ".jpg")
