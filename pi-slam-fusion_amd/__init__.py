"""pi-slam-fusion_amd -- ctypes host binding of libpifusion.so.

Mirrors the reference's Map2D interface (Map2DFusion/Map2D.h:79-98:
create / prepare / feed / save / queueSize) and the MultiBandMap2DCPU::Ele tile
surface (MultiBandMap2DCPU.h:32-51).  The arithmetic lives in the HIP library;
this module only marshals pointers.  There is no CPU fallback: creating a map
without a HIP device (or without the built library) raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PF_LIB") or os.path.join(_HERE, "libpifusion.so")      # PF_LIB: another build of the same library (A/B runs)
ELE_PIXELS = 256

# Map2D::Map2DType (Map2D.h:83)
NoType, TypeCPU, TypeGPU, TypeMultiBandCPU, TypeRender = 0, 1, 2, 3, 4
PF_8UC3, PF_8UC4, PF_16SC3, PF_32FC3 = 16, 24, 19, 21


def host_array(shape, dtype=np.uint8):
    """numpy array over page-locked host memory (pf_host_alloc); freed with the array."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib().pf_host_alloc(n)
    if not p:
        raise MemoryError("pf_host_alloc(%d)" % n)
    buf = (C.c_uint8 * n).from_address(p)
    a = np.frombuffer(buf, dtype=dtype).reshape(shape)
    import weakref
    weakref.finalize(buf, lib().pf_host_free, p)
    return a


class Options(C.Structure):
    _fields_ = [("band_number", C.c_int), ("force_float", C.c_int), ("high_quality_show", C.c_int),
                ("weight_type", C.c_int), ("bg_color", C.c_int), ("resolution", C.c_double),
                ("scale", C.c_double), ("device", C.c_int), ("shard_rank", C.c_int),
                ("shard_count", C.c_int), ("shard_block", C.c_int), ("max_queue", C.c_int), ("fused", C.c_int),
                ("lookahead", C.c_int)]


class Image(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("type", C.c_int), ("data", C.c_void_p),
                ("step", C.c_size_t)]


class DistStats(C.Structure):
    _fields_ = [("bytes_sent", C.c_ulonglong), ("bytes_received", C.c_ulonglong), ("strips_sent", C.c_ulonglong),
                ("strips_received", C.c_ulonglong), ("tiles", C.c_ulonglong), ("peers", C.c_ulonglong),
                ("plan_ms", C.c_double), ("pack_ms", C.c_double), ("exchange_ms", C.c_double), ("compute_ms", C.c_double),
                ("verified", C.c_ulonglong)]


# pf_exchange_fn: all-to-all-v over host buffers (include/pifusion.h)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_size_t), C.c_int)

_lib = None


def build(verbose=False):
    """Compile libpifusion.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libpifusion.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME, different file name).
    # Loading torch first makes the dynamic linker hand that one runtime to this library
    # too; the other order ends with two HIP runtimes in the process and torch blind.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.pf_default_options.argtypes = [C.POINTER(Options)]; L.pf_default_options.restype = None
    L.pf_options_set.argtypes = [C.POINTER(Options), C.c_char_p, C.c_char_p]
    L.pf_create.argtypes = [C.c_int, C.c_int, C.POINTER(Options)]; L.pf_create.restype = vp
    L.pf_destroy.argtypes = [vp]; L.pf_destroy.restype = None
    L.pf_last_error.restype = C.c_char_p
    L.pf_prepare.argtypes = [vp, dp, dp, C.c_int, C.POINTER(Image), dp]
    L.pf_feed.argtypes = [vp, C.POINTER(Image), dp]
    L.pf_feed_device.argtypes = [vp, C.POINTER(Image), dp]
    L.pf_queue_size.argtypes = [vp]; L.pf_queue_size.restype = C.c_uint
    L.pf_debug_read_last_frame.argtypes = [vp, vp, C.c_size_t]; L.pf_debug_read_last_frame.restype = C.c_long
    L.pf_sync.argtypes = [vp]
    L.pf_save.argtypes = [vp, C.c_char_p]
    L.pf_save_to_memory.argtypes = [vp, vp, ip, ip, ip, ip]
    L.pf_write_image.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
    L.pf_image_info.argtypes = [C.c_char_p, ip, ip]
    L.pf_read_image.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
    L.pf_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, ip, ip, ip]
    L.pf_jpeg_decode_bgr.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_int, C.c_int]
    L.pf_jpeg_decode_device.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_int, C.c_int, vp]
    L.pf_feed_jpeg.argtypes = [vp, C.c_char_p, C.c_size_t, dp]
    L.pf_debug_jpeg_huffman.argtypes = [vp, C.POINTER(C.c_longlong)]; L.pf_debug_jpeg_huffman.restype = None
    L.pf_feed_jpeg_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), dp, C.c_int, ip]
    L.pf_num_levels.argtypes = [vp]
    L.pf_pyramid_type.argtypes = [vp]
    L.pf_grid.argtypes = [vp, ip, dp]
    L.pf_tile_count.argtypes = [vp]
    L.pf_tile_coords.argtypes = [vp, ip, C.c_int]
    L.pf_get_tile_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.pf_get_tile_bgra.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pf_blend_tile_raw.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pf_blend_tile.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pf_blend_changed.argtypes = [vp, ip, vp, C.c_int]
    if hasattr(L, "pf_blend_tiles") or not os.environ.get("PF_LIB"):
        L.pf_blend_tiles.argtypes = [vp, ip, C.c_int, vp]
        L.pf_host_alloc.argtypes = [C.c_size_t]; L.pf_host_alloc.restype = vp
        L.pf_host_free.argtypes = [vp]; L.pf_host_free.restype = None
    L.pf_format_map_update.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.pf_map_update_command.argtypes = [vp, C.c_int, C.c_int, dp, C.c_char_p, C.c_int]
    L.pf_lnglat_from_distance.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]; L.pf_lnglat_from_distance.restype = None
    L.pf_normalize_using_weight_map.argtypes = [vp, vp, C.c_size_t]
    L.pf_mul_weight_map.argtypes = [vp, vp, C.c_size_t]
    L.pf_tile_owner.argtypes = [C.POINTER(Options), C.c_int, C.c_int]
    L.pf_halo_bytes.argtypes = [vp, C.c_int, C.c_int]; L.pf_halo_bytes.restype = C.c_size_t
    L.pf_halo_pack.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.pf_blend_tile_halo.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), vp, vp]
    L.pf_tile_bytes.argtypes = [vp]; L.pf_tile_bytes.restype = C.c_size_t
    L.pf_tile_export.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pf_tile_import.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pf_dist_unique_id.argtypes = [vp]
    L.pf_dist_init_rccl.argtypes = [vp, vp, C.c_int, C.c_int]; L.pf_dist_init_rccl.restype = vp
    L.pf_dist_init_host.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp]; L.pf_dist_init_host.restype = vp
    L.pf_dist_destroy.argtypes = [vp]; L.pf_dist_destroy.restype = None
    L.pf_dist_blend_changed.argtypes = [vp, ip, vp, C.c_int]
    L.pf_dist_feed.argtypes = [vp, C.POINTER(Image), dp, C.c_int]
    L.pf_dist_feed_jpeg.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_int, C.c_int, dp, C.c_int]
    L.pf_dist_save.argtypes = [vp, C.c_char_p]
    L.pf_dist_save_to_memory.argtypes = [vp, vp, ip, ip, ip, ip]
    L.pf_dist_last_stats.argtypes = [vp, C.POINTER(DistStats)]
    L.pf_dist_info.argtypes = [vp, ip, ip, C.POINTER(C.c_char_p)]
    L.pf_dist_set_verify.argtypes = [vp, C.c_int]
    L.pf_dist_plan_blend.argtypes = [C.c_int, C.c_int, ip, ip, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_size_t), vp, C.c_int, ip,
                                     vp, C.c_int, ip, ip, C.c_int, ip]
    if hasattr(L, "pf_debug_compact_launches") or not os.environ.get("PF_LIB"):
        L.pf_debug_compact_launches.argtypes = []; L.pf_debug_compact_launches.restype = C.c_longlong
    L.pf_profile_enable.argtypes = [vp, C.c_int]
    L.pf_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), dp, C.POINTER(C.c_longlong), dp]
    # (PF_LIB may name an older build of the library for an A/B round, tools/ab.sh: it lacks this round's entry points)
    if hasattr(L, "pf_profile_read_run") or not os.environ.get("PF_LIB"):
        L.pf_profile_read_run.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), dp, C.POINTER(C.c_longlong), dp, dp]
    L.pf_profile_reset.argtypes = [vp]
    L.pf_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.pf_reserve_tiles.argtypes = [vp, C.c_longlong]
    L.pf_debug_culled_tiles.argtypes = [vp]; L.pf_debug_culled_tiles.restype = C.c_longlong
    L.pf_set_cull.argtypes = [vp, C.c_int]; L.pf_set_cull.restype = None
    if hasattr(L, "pf_debug_render_log") or not os.environ.get("PF_LIB"):
        L.pf_debug_render_log.argtypes = [vp, C.POINTER(C.c_longlong), C.c_int]
    L.pf_debug_culled_cells.argtypes = [vp]; L.pf_debug_culled_cells.restype = C.c_longlong
    L.pf_debug_level0_exact_px.argtypes = [vp]; L.pf_debug_level0_exact_px.restype = C.c_double
    L.pf_render_stats.argtypes = [vp, dp]
    L.pf_timer_read.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), dp, dp, dp]
    L.pf_timer_reset.argtypes = [vp]
    _lib = L
    return L


POSE7 = C.c_double * 7


def _pose(p):
    if isinstance(p, POSE7):          # a caller that feeds thousands of poses a second converts them once (bench.py)
        return p, p
    a = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def default_options(**kw):
    o = Options()
    lib().pf_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def se3_inverse(a):
    a, pa = _pose(a); o = np.zeros(7); lib().pf_se3_inverse(pa, o.ctypes.data_as(C.POINTER(C.c_double))); return o


def format_map_update(plane, gps_origin, min_x, min_y, ele_size, x, y):
    """pf_format_map_update: the reference's Map2DUpdate text for dense tile index (x, y) (C implementation)."""
    _, pp = _pose(plane)
    buf = C.create_string_buffer(256)
    g = (C.c_double * 3)(*[float(v) for v in gps_origin])
    n = lib().pf_format_map_update(pp, g, float(min_x), float(min_y), float(ele_size), int(x), int(y), buf, 256)
    return buf.value.decode() if n > 0 else None


def lnglat_from_distance(lng1, lat1, dx, dy):
    a, b = C.c_double(), C.c_double()
    lib().pf_lnglat_from_distance(float(lng1), float(lat1), float(dx), float(dy), C.byref(a), C.byref(b))
    return a.value, b.value


def se3_mul(a, b):
    a, pa = _pose(a); b, pb = _pose(b); o = np.zeros(7)
    lib().pf_se3_mul(pa, pb, o.ctypes.data_as(C.POINTER(C.c_double))); return o


def so3_rotate(q, p):
    q, pq = _pose(q); p, pp = _pose(p); o = np.zeros(3)
    lib().pf_so3_rotate(pq, pp, o.ctypes.data_as(C.POINTER(C.c_double))); return o


def footprint(cam, pose_plane):
    c, pc = _pose(cam); p, pp = _pose(pose_plane); o = np.zeros(8)
    ok = lib().pf_footprint(pc, pp, o.ctypes.data_as(C.POINTER(C.c_double)))
    return o.reshape(4, 2) if ok else None


def perspective_transform(src, dst):
    s = np.ascontiguousarray(src, dtype=np.float32).reshape(8); d = np.ascontiguousarray(dst, dtype=np.float32).reshape(8)
    M = np.zeros(9)
    lib().pf_perspective_transform(s.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p))
    return M.reshape(3, 3)


def write_image(filename, bgr):
    """cv::imwrite leg of save() (MultiBandMap2DCPU.cpp:841): HxWx3 BGR uint8 -> .png / .ppm"""
    a = np.ascontiguousarray(bgr, dtype=np.uint8)
    return bool(lib().pf_write_image(filename.encode(), a.ctypes.data, a.shape[0], a.shape[1]))


def read_image(filename):
    """cv::imread(filename) of the file driver (backup/map2dfusion.cpp:129-132): JPEG or binary PPM -> HxWx3 BGR uint8.
    Raises on a file the library cannot read (there is no other decoder behind it)."""
    L = lib(); r = C.c_int(); c = C.c_int()
    if not L.pf_image_info(filename.encode(), C.byref(r), C.byref(c)):
        raise RuntimeError("read_image: %s" % L.pf_last_error().decode())
    out = np.empty((r.value, c.value, 3), np.uint8)
    if not L.pf_read_image(filename.encode(), out.ctypes.data, r.value, c.value):
        raise RuntimeError("read_image: %s" % L.pf_last_error().decode())
    return out


def decode_jpeg(data):
    """cv::imdecode-style entry: the bytes of a JPEG stream -> HxWx3 BGR uint8 (libjpeg's default decode, byte for byte)."""
    L = lib(); b = bytes(data); r = C.c_int(); c = C.c_int(); k = C.c_int()
    if not L.pf_jpeg_info(b, len(b), C.byref(r), C.byref(c), C.byref(k)):
        raise ValueError("decode_jpeg: %s" % L.pf_last_error().decode())
    out = np.empty((r.value, c.value, 3), np.uint8)
    if not L.pf_jpeg_decode_bgr(b, len(b), out.ctypes.data, r.value, c.value):
        raise ValueError("decode_jpeg: %s" % L.pf_last_error().decode())
    return out


def jpeg_info(data):
    """(rows, cols, components) of a JPEG stream"""
    L = lib(); b = bytes(data); r = C.c_int(); c = C.c_int(); k = C.c_int()
    if not L.pf_jpeg_info(b, len(b), C.byref(r), C.byref(c), C.byref(k)):
        raise ValueError("jpeg_info: %s" % L.pf_last_error().decode())
    return r.value, c.value, k.value


def decode_jpeg_device(data, dev_ptr, rows, cols, stream=None):
    """The same decode with its back end on the GPU (csrc/jpeg_device.hip): BGR8, rows*cols*3 bytes at the device address `dev_ptr`,
    complete in the order of `stream` (a hipStream_t handle; None = the default stream).  Returns once the work is queued."""
    L = lib(); b = bytes(data)
    if not L.pf_jpeg_decode_device(b, len(b), dev_ptr, rows, cols, stream):
        raise ValueError("decode_jpeg_device: %s" % L.pf_last_error().decode())


def jpeg_huffman_counts(map2d=None):
    """(frames whose Huffman pass ran on the GPU, frames that fell back to the host after trying, rounds of the most recent GPU pass) of a map's
    decoder, or of decode_jpeg_device's when map2d is None"""
    out = (C.c_longlong * 3)()
    lib().pf_debug_jpeg_huffman(map2d._h if map2d is not None else None, out)
    return tuple(out)


def tile_owner(opt, ix, iy):
    return lib().pf_tile_owner(C.byref(opt), ix, iy)


class Map2D:
    """Map2D::create(type, thread) -> object with prepare/feed/save/queueSize."""

    def __init__(self, handle, opt):
        self._h = handle
        self.opt = opt

    @staticmethod
    def create(type=TypeMultiBandCPU, thread=False, options=None, **kw):
        opt = options if options is not None else default_options(**kw)
        h = lib().pf_create(type, 1 if thread else 0, C.byref(opt))
        if not h:
            if type in (NoType, TypeRender):
                return None
            raise RuntimeError("pf_create failed: %s" % lib().pf_last_error().decode())
        return Map2D(h, opt)

    def close(self):
        if self._h:
            lib().pf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Map2D virtuals
    def prepare(self, plane, camera, poses, images=None):
        """plane: 7 doubles; camera: [w h fx fy cx cy]; poses: n x 7 camera-to-world."""
        pl, ppl = _pose(plane); cam, pc = _pose(camera)
        ps = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
        imgs = None
        keep = []
        if images is not None:
            imgs = (Image * len(images))()
            for i, im in enumerate(images):
                im = np.ascontiguousarray(im, dtype=np.uint8); keep.append(im)
                imgs[i] = Image(im.shape[0], im.shape[1], PF_8UC3, im.ctypes.data, 0)
        return bool(lib().pf_prepare(self._h, ppl, pc, ps.shape[0], imgs, ps.ctypes.data_as(C.POINTER(C.c_double))))

    def feed(self, img, pose):
        """img: HxWx3 (BGR) or HxWx4 (BGRA) uint8 numpy array (host), None for a geometry-only frame, or the bytes of a .jpg
        file (decoded by the map: feed_jpeg)."""
        if isinstance(img, (bytes, bytearray, memoryview)):
            return self.feed_jpeg(img, pose)
        p, pp = _pose(pose)
        if img is None:
            return bool(lib().pf_feed(self._h, None, pp))
        img = np.asarray(img)
        step = 0
        if img.ndim == 3 and img.dtype == np.uint8 and img.strides[2] == 1 and img.strides[1] == img.shape[2] and img.strides[0] >= img.shape[1] * img.shape[2]:
            step = img.strides[0]                       # row-padded view (cv::Mat::step): handed over as is
        else:
            img = np.ascontiguousarray(img)
        typ = -1
        if img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] in (3, 4):
            typ = PF_8UC3 if img.shape[2] == 3 else PF_8UC4      # BGRA as the tracker holds it (TrackerOpt.cpp:376)
        im = Image(img.shape[0], img.shape[1], typ, img.ctypes.data, step)
        return bool(lib().pf_feed(self._h, C.byref(im), pp))

    def feed_device(self, data_ptr, rows, cols, pose, step=0):
        """Frame already resident in HBM (e.g. torch tensor .data_ptr())."""
        p, pp = _pose(pose)
        im = Image(rows, cols, PF_8UC3, data_ptr, step)
        return bool(lib().pf_feed_device(self._h, C.byref(im), pp))

    def feed_jpeg(self, data, pose):
        """feed(cv::imread(file), pose) in one call: the JPEG stream is decoded into HBM on the map's stream (Huffman on this thread,
        IDCT / upsampling / colour on the GPU) and rendered from there."""
        p, pp = _pose(pose)
        b = bytes(data)
        return bool(lib().pf_feed_jpeg(self._h, b, len(b), pp))

    def feed_jpeg_batch(self, streams, poses, threads=0):
        """n .jpg keyframes at once: Huffman passes on `threads` host threads (0: one per frame), the rest on the GPU, fed in order.
        Returns the list of per-frame results (what feed_jpeg returns)."""
        n = len(streams)
        keep = [bytes(s) for s in streams]
        data = (C.c_char_p * n)(*keep); lens = (C.c_size_t * n)(*[len(s) for s in keep])
        pp = (C.c_double * (7 * n))(*[float(v) for p in poses for v in p])
        res = (C.c_int * n)()
        lib().pf_feed_jpeg_batch(self._h, n, data, lens, pp, threads, res)
        return [bool(r) for r in res]

    def read_last_frame(self):
        """test hook: bytes of the most recently uploaded host frame as they lie in HBM"""
        n = lib().pf_debug_read_last_frame(self._h, None, 0)
        if n < 0:
            return None
        out = np.empty(n, np.uint8)
        return out if lib().pf_debug_read_last_frame(self._h, out.ctypes.data, n) == n else None

    def queueSize(self):
        return int(lib().pf_queue_size(self._h))

    def sync(self):
        return bool(lib().pf_sync(self._h))

    def save(self, filename):
        return bool(lib().pf_save(self._h, filename.encode()))

    def save_to_memory(self, alloc=None):
        """(mosaic BGR8, (tile x0, tile y0)); alloc(shape) -> uint8 array supplies the buffer (e.g. host_array)."""
        r, c, x0, y0 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if not lib().pf_save_to_memory(self._h, None, C.byref(r), C.byref(c), C.byref(x0), C.byref(y0)):
            return None
        out = alloc((r.value, c.value, 3)) if alloc else np.empty((r.value, c.value, 3), np.uint8)
        if not lib().pf_save_to_memory(self._h, out.ctypes.data, C.byref(r), C.byref(c), C.byref(x0), C.byref(y0)):
            return None
        return out, (x0.value, y0.value)

    # ---- Ele surface
    @property
    def num_levels(self):
        return lib().pf_num_levels(self._h)

    @property
    def dtype(self):
        return np.float32 if lib().pf_pyramid_type(self._h) == PF_32FC3 else np.int16

    def grid(self):
        dims = (C.c_int * 4)(); geo = (C.c_double * 6)()
        if not lib().pf_grid(self._h, dims, geo):
            return None
        return list(dims), list(geo)

    def tiles(self):
        n = lib().pf_tile_count(self._h)
        xy = (C.c_int * (2 * max(n, 1)))()
        lib().pf_tile_coords(self._h, xy, n)
        return [(xy[2 * i], xy[2 * i + 1]) for i in range(n)]

    def tile_level(self, ix, iy, level):
        s = ELE_PIXELS >> level
        lap = np.empty((s, s, 3), self.dtype); w = np.empty((s, s), np.float32)
        ok = lib().pf_get_tile_level(self._h, ix, iy, level, lap.ctypes.data, w.ctypes.data)
        return (lap, w) if ok else None

    def tile_bgra(self, ix, iy):
        out = np.empty((ELE_PIXELS, ELE_PIXELS, 4), np.uint8)
        return out if lib().pf_get_tile_bgra(self._h, ix, iy, out.ctypes.data) else None

    def blend_tile_raw(self, ix, iy):
        out = np.empty((ELE_PIXELS, ELE_PIXELS, 3), self.dtype)
        return out if lib().pf_blend_tile_raw(self._h, ix, iy, out.ctypes.data) else None

    def blend_tile(self, ix, iy):
        out = np.empty((ELE_PIXELS, ELE_PIXELS, 3), np.uint8)
        return out if lib().pf_blend_tile(self._h, ix, iy, out.ctypes.data) else None

    def map_update_command(self, ix, iy, gps_origin):
        """draw()'s "Map2DUpdate LastTexMat ..." text for tile (ix, iy), or None under the reference's gate
        `updated && !inborder` (MultiBandMap2DCPU.cpp:744); Fuse2Google is the caller's flag."""
        buf = C.create_string_buffer(256)
        g = (C.c_double * 3)(*[float(v) for v in gps_origin])
        n = lib().pf_map_update_command(self._h, ix, iy, g, buf, 256)
        return buf.value.decode() if n > 0 else None

    def blend_changed(self, cap=4096, out=None):
        """draw()'s texture refresh: ([(ix, iy)...], n x 256 x 256 x 3 uint8).  out: a caller-owned buffer of at least cap tiles
        (e.g. host_array(): page-locked, filled straight from HBM)."""
        xy = (C.c_int * (2 * cap))()
        if out is None:
            out = np.empty((cap, ELE_PIXELS, ELE_PIXELS, 3), np.uint8)
        n = lib().pf_blend_changed(self._h, xy, out.ctypes.data, cap)
        return [(xy[2 * i], xy[2 * i + 1]) for i in range(n)], out[:n]

    def blend_tiles(self, tiles, out=None):
        """Ele::blend + 8U view of the listed tiles (Ischanged untouched); tiles without pyramid keep out's bytes."""
        n = len(tiles)
        xy = (C.c_int * (2 * n))(*[v for t in tiles for v in t])
        if out is None:
            out = np.zeros((n, ELE_PIXELS, ELE_PIXELS, 3), np.uint8)
        return out[:n] if lib().pf_blend_tiles(self._h, xy, n, out.ctypes.data) else None

    # ---- multi-GPU seam exchange
    def halo_bytes(self, dx, dy):
        return int(lib().pf_halo_bytes(self._h, dx, dy))

    def halo_pack(self, ix, iy, dx, dy, dev_ptr):
        return bool(lib().pf_halo_pack(self._h, ix, iy, dx, dy, dev_ptr))

    def blend_tile_halo(self, ix, iy, halo_ptrs, raw=False):
        arr = (C.c_void_p * 9)(*[(p if p else None) for p in halo_ptrs])
        if raw:
            out = np.empty((ELE_PIXELS, ELE_PIXELS, 3), self.dtype)
            ok = lib().pf_blend_tile_halo(self._h, ix, iy, arr, None, out.ctypes.data)
        else:
            out = np.empty((ELE_PIXELS, ELE_PIXELS, 3), np.uint8)
            ok = lib().pf_blend_tile_halo(self._h, ix, iy, arr, out.ctypes.data, None)
        return out if ok else None

    def tile_bytes(self):
        return int(lib().pf_tile_bytes(self._h))

    def tile_export(self, ix, iy, dev_ptr):
        return bool(lib().pf_tile_export(self._h, ix, iy, dev_ptr))

    def tile_import(self, ix, iy, dev_ptr):
        return bool(lib().pf_tile_import(self._h, ix, iy, dev_ptr))

    # ---- measurement
    def profile_enable(self, mode=1):
        lib().pf_profile_enable(self._h, mode)

    def profile_reset(self):
        lib().pf_profile_reset(self._h)

    def profile_read(self):
        cap = 32
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); n = (C.c_longlong * cap)(); by = (C.c_double * cap)(); br = (C.c_double * cap)()
        if not hasattr(lib(), "pf_profile_read_run"):          # PF_LIB = an older build (A/B rounds)
            k = lib().pf_profile_read(self._h, cap, names, ms, n, by)
            return {names[i].decode(): {"ms": ms[i], "launches": n[i], "alg_bytes": by[i], "alg_bytes_run": by[i]} for i in range(k)}
        k = lib().pf_profile_read_run(self._h, cap, names, ms, n, by, br)
        # alg_bytes: SURVEY 8d's bytes for every tile of every canvas; alg_bytes_run: for the part of the canvases the launches' blocks processed
        return {names[i].decode(): {"ms": ms[i], "launches": n[i], "alg_bytes": by[i], "alg_bytes_run": br[i]} for i in range(k)}

    def reserve_tiles(self, n_tiles):
        """Allocator hint: HBM for n_tiles more tiles now (see pf_reserve_tiles)."""
        return bool(lib().pf_reserve_tiles(self._h, int(n_tiles)))

    def render_stats(self):
        o = (C.c_double * 4)()
        lib().pf_render_stats(self._h, o)
        return {"frames_with_pixels": int(o[0]), "level0_px": o[1], "owned_px": o[2], "tiles_held": int(o[3])}

    def timers(self):
        """host section timers under the reference's section names (pi::timer): {name: {calls, mean_s, min_s, max_s}}"""
        cap = 16
        names = (C.c_char_p * cap)(); calls = (C.c_longlong * cap)(); mean = (C.c_double * cap)(); mn = (C.c_double * cap)(); mx = (C.c_double * cap)()
        k = lib().pf_timer_read(self._h, cap, names, calls, mean, mn, mx)
        return {names[i].decode(): {"calls": calls[i], "mean_s": mean[i], "min_s": mn[i], "max_s": mx[i]} for i in range(k)}

    def timer_reset(self):
        lib().pf_timer_reset(self._h)

    def set_cull(self, on):
        """the cull of render_frame on / off (off: every tile of every canvas rendered, as the reference does; same mosaic)"""
        lib().pf_set_cull(self._h, 1 if on else 0)

    def render_log(self):
        """test hook: indices of the feed() calls whose keyframes were rendered, in render order (pf_debug_render_log)"""
        n = lib().pf_debug_render_log(self._h, None, 0)
        out = (C.c_longlong * max(n, 1))()
        k = lib().pf_debug_render_log(self._h, out, n)
        return [int(out[i]) for i in range(min(n, k))]

    def culled_tiles(self):
        """tiles left out of launches because the keyframe could not win the select anywhere in them (diagnostics)"""
        return int(lib().pf_debug_culled_tiles(self._h))

    def culled_cells(self):
        """64 x 64 cells switched off inside tiles that were rendered (diagnostics)"""
        return int(lib().pf_debug_culled_cells(self._h))

    def stats(self):
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        lib().pf_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"rendered": a.value, "rejected": b.value, "dropped": c.value}
