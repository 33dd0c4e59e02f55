"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/oracle.h).  The product never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ELE = 256


class Options(C.Structure):
    _fields_ = [("band_num", C.c_int), ("force_float", C.c_int), ("weight_type", C.c_int),
                ("high_quality", C.c_int), ("bg_color", C.c_int),
                ("resolution", C.c_double), ("scale", C.c_double), ("single_band", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def use_openmp():
    """Switch this process to liboracle_omp.so (same code, row/tile loops in parallel): the "generous"
    all-core CPU baseline of BASELINE.md (B2).  Call before the first lib()."""
    global _LIB, _NAME
    assert _LIB is None, "oracle library already loaded"
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_omp.so"])
    _NAME = "liboracle_omp.so"


_NAME = "liboracle.so"


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, _NAME)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    dp, fp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
    L.orc_map_create.restype = vp
    L.orc_map_create.argtypes = [C.POINTER(Options)]
    for name in ("orc_map_destroy",):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = None
    L.orc_map_prepare.argtypes = [vp, dp, dp, C.c_int, dp]
    L.orc_map_feed.argtypes = [vp, vp, C.c_int, C.c_int, dp]
    L.orc_map_footprint.argtypes = [vp, dp, dp]
    L.orc_map_grid.argtypes = [vp, ip, dp]
    L.orc_map_grid.restype = None
    L.orc_map_num_levels.argtypes = [vp]
    L.orc_map_tile_count.argtypes = [vp]
    L.orc_map_tile_coords.argtypes = [vp, ip, C.c_int]
    L.orc_map_get_tile_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.orc_map_get_tile_bgra.argtypes = [vp, C.c_int, C.c_int, vp]
    L.orc_map_blend_tile_raw.argtypes = [vp, C.c_int, C.c_int, vp]
    L.orc_map_blend_tile.argtypes = [vp, C.c_int, C.c_int, vp]
    L.orc_map_save_size.argtypes = [vp, ip, ip, ip, ip]
    L.orc_map_save.argtypes = [vp, vp]
    L.orc_map_last_canvas.argtypes = [vp, ip, dp]
    L.orc_map_keep_last.argtypes = [vp, C.c_int]
    L.orc_map_keep_last.restype = None
    L.orc_map_last_level.argtypes = [vp, C.c_int, vp, vp]
    _LIB = L
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


# ------------------------------------------------------------------ raw ops
def se3_inverse(a):
    a, pa = _d(a); o = np.zeros(7); lib().orc_se3_inverse(pa, o.ctypes.data_as(C.POINTER(C.c_double))); return o


def se3_mul(a, b):
    a, pa = _d(a); b, pb = _d(b); o = np.zeros(7)
    lib().orc_se3_mul(pa, pb, o.ctypes.data_as(C.POINTER(C.c_double))); return o


def so3_rotate(q, p):
    q, pq = _d(q); p, pp = _d(p); o = np.zeros(3)
    lib().orc_so3_rotate(pq, pp, o.ctypes.data_as(C.POINTER(C.c_double))); return o


def get_perspective_transform(src, dst):
    s = np.ascontiguousarray(src, dtype=np.float32).reshape(8)
    d = np.ascontiguousarray(dst, dtype=np.float32).reshape(8)
    M = np.zeros(9)
    lib().orc_get_perspective_transform(s.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                        M.ctypes.data_as(C.c_void_p))
    return M.reshape(3, 3)


def invert3x3(M):
    M, pm = _d(np.asarray(M).reshape(9)); o = np.zeros(9)
    ok = lib().orc_invert3x3(pm, o.ctypes.data_as(C.POINTER(C.c_double)))
    return o.reshape(3, 3) if ok else None


def weight_image(rows, cols, weight_type=0):
    w = np.empty((rows, cols), np.float32)
    lib().orc_weight_image(w.ctypes.data_as(C.c_void_p), rows, cols, weight_type)
    return w


def _img(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    if a.ndim == 2:
        a = a[:, :, None]
    return a


def _sfx(dt):
    return "16s" if dt == np.int16 else "32f"


def warp_linear_reflect(src, M0, drows, dcols):
    dt = np.int16 if src.dtype == np.int16 else np.float32
    s = _img(src, dt); cn = s.shape[2]
    d = np.empty((drows, dcols, cn), dt)
    M0, pm = _d(np.asarray(M0).reshape(9))
    getattr(lib(), "orc_warp_linear_reflect_" + _sfx(dt))(
        s.ctypes.data_as(C.c_void_p), s.shape[0], s.shape[1], cn, d.ctypes.data_as(C.c_void_p), drows, dcols, pm)
    return d


def warp_nearest_const(src, M0, drows, dcols):
    s = _img(src, np.float32); cn = s.shape[2]
    d = np.empty((drows, dcols, cn), np.float32)
    M0, pm = _d(np.asarray(M0).reshape(9))
    lib().orc_warp_nearest_const_32f(s.ctypes.data_as(C.c_void_p), s.shape[0], s.shape[1], cn,
                                     d.ctypes.data_as(C.c_void_p), drows, dcols, pm)
    return d


def warp_linear_const_8u(src, M0, drows, dcols):
    s = _img(src, np.uint8); cn = s.shape[2]
    d = np.empty((drows, dcols, cn), np.uint8)
    M0, pm = _d(np.asarray(M0).reshape(9))
    lib().orc_warp_linear_const_8u(s.ctypes.data_as(C.c_void_p), s.shape[0], s.shape[1], cn, d.ctypes.data_as(C.c_void_p), drows, dcols, pm)
    return d


def weight_image_8uc4(rows, cols, weight_type=0):
    w = np.empty((rows, cols, 4), np.uint8)
    lib().orc_weight_image_8uc4(w.ctypes.data_as(C.c_void_p), rows, cols, weight_type)
    return w


def pyr_down(src):
    dt = np.int16 if src.dtype == np.int16 else np.float32
    s = _img(src, dt); cn = s.shape[2]
    d = np.empty(((s.shape[0] + 1) // 2, (s.shape[1] + 1) // 2, cn), dt)
    getattr(lib(), "orc_pyr_down_" + _sfx(dt))(s.ctypes.data_as(C.c_void_p), s.shape[0], s.shape[1], cn,
                                                d.ctypes.data_as(C.c_void_p))
    return d


def pyr_up(src):
    dt = np.int16 if src.dtype == np.int16 else np.float32
    s = _img(src, dt); cn = s.shape[2]
    d = np.empty((s.shape[0] * 2, s.shape[1] * 2, cn), dt)
    getattr(lib(), "orc_pyr_up_" + _sfx(dt))(s.ctypes.data_as(C.c_void_p), s.shape[0], s.shape[1], cn,
                                              d.ctypes.data_as(C.c_void_p), d.shape[0], d.shape[1])
    return d


def _levels(img, n):
    dt = np.int16 if img.dtype == np.int16 else np.float32
    lv = [_img(img, dt).copy()]
    for _ in range(n):
        r, c, cn = lv[-1].shape
        lv.append(np.zeros(((r + 1) // 2, (c + 1) // 2, cn), dt))
    return lv, dt


def _ptr_array(lv):
    arr = (C.c_void_p * len(lv))()
    for i, a in enumerate(lv):
        arr[i] = a.ctypes.data
    return arr


def create_laplace_pyr(img, n):
    lv, dt = _levels(img, n)
    getattr(lib(), "orc_create_laplace_pyr_" + _sfx(dt))(_ptr_array(lv), lv[0].shape[0], lv[0].shape[1], lv[0].shape[2], n)
    return lv


def restore_from_laplace_pyr(levels):
    dt = np.int16 if levels[0].dtype == np.int16 else np.float32
    lv = [_img(a, dt).copy() for a in levels]
    getattr(lib(), "orc_restore_from_laplace_pyr_" + _sfx(dt))(_ptr_array(lv), lv[0].shape[0], lv[0].shape[1],
                                                                lv[0].shape[2], len(lv) - 1)
    return lv[0]


# ------------------------------------------------------------------ the map
class OracleMap:
    """MultiBandMap2DCPU restated (thread=false)."""

    def __init__(self, band_num=5, force_float=0, weight_type=0, high_quality=1, bg_color=0,
                 resolution=0.0, scale=1.0, single_band=0):
        self.opt = Options(band_num, force_float, weight_type, high_quality, bg_color, resolution, scale, single_band)
        self.h = lib().orc_map_create(C.byref(self.opt))
        self.force_float = force_float
        self.dtype = np.float32 if force_float else np.int16

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_map_destroy(self.h); self.h = None

    def prepare(self, plane, cam, poses):
        plane, pp = _d(plane); cam, pc = _d(cam)
        poses, ps = _d(np.asarray(poses).reshape(-1, 7))
        return bool(lib().orc_map_prepare(self.h, pp, pc, poses.shape[0], ps))

    def feed(self, bgr, pose):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        pose, pq = _d(pose)
        return bool(lib().orc_map_feed(self.h, bgr.ctypes.data_as(C.c_void_p), bgr.shape[0], bgr.shape[1], pq))

    def footprint(self, pose):
        pose, pq = _d(pose); o = np.zeros(8)
        ok = lib().orc_map_footprint(self.h, pq, o.ctypes.data_as(C.POINTER(C.c_double)))
        return o.reshape(4, 2) if ok else None

    def grid(self):
        dims = (C.c_int * 4)(); geo = (C.c_double * 6)()
        lib().orc_map_grid(self.h, dims, geo)
        return list(dims), list(geo)

    @property
    def num_levels(self):
        return lib().orc_map_num_levels(self.h)

    def tiles(self):
        n = lib().orc_map_tile_count(self.h)
        xy = (C.c_int * (2 * max(n, 1)))()
        lib().orc_map_tile_coords(self.h, xy, n)
        return [(xy[2 * i], xy[2 * i + 1]) for i in range(n)]

    def tile_level(self, ix, iy, level):
        s = ELE >> level
        lap = np.empty((s, s, 3), self.dtype); w = np.empty((s, s), np.float32)
        ok = lib().orc_map_get_tile_level(self.h, ix, iy, level, lap.ctypes.data_as(C.c_void_p),
                                          w.ctypes.data_as(C.c_void_p))
        return (lap, w) if ok else None

    def tile_bgra(self, ix, iy):
        out = np.empty((ELE, ELE, 4), np.uint8)
        return out if lib().orc_map_get_tile_bgra(self.h, ix, iy, out.ctypes.data_as(C.c_void_p)) else None

    def blend_tile_raw(self, ix, iy):
        out = np.empty((ELE, ELE, 3), self.dtype)
        ok = lib().orc_map_blend_tile_raw(self.h, ix, iy, out.ctypes.data_as(C.c_void_p))
        return out if ok else None

    def blend_tile(self, ix, iy):
        out = np.empty((ELE, ELE, 3), np.uint8)
        ok = lib().orc_map_blend_tile(self.h, ix, iy, out.ctypes.data_as(C.c_void_p))
        return out if ok else None

    def save(self):
        r, c, x0, y0 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if not lib().orc_map_save_size(self.h, C.byref(r), C.byref(c), C.byref(x0), C.byref(y0)):
            return None
        out = np.empty((r.value, c.value, 3), np.uint8)
        lib().orc_map_save(self.h, out.ctypes.data_as(C.c_void_p))
        return out, (x0.value, y0.value)

    def keep_last(self, on=True):
        lib().orc_map_keep_last(self.h, 1 if on else 0)

    def last_canvas(self):
        dims = (C.c_int * 4)(); M = (C.c_double * 9)()
        if not lib().orc_map_last_canvas(self.h, dims, M):
            return None
        return list(dims), np.array(list(M)).reshape(3, 3)

    def last_level(self, level):
        dims, _ = self.last_canvas()
        r, c = (dims[3] * ELE) >> level, (dims[2] * ELE) >> level
        lap = np.empty((r, c, 3), self.dtype); w = np.empty((r, c), np.float32)
        ok = lib().orc_map_last_level(self.h, level, lap.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
        return (lap, w) if ok else None
