"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/pifusion.h declares; the stateless host geometry entry points agree with the
vectors generated from the reference's own SE3 headers (tests/golden/se3_vectors.json,
made by tests/golden/make_se3_vectors.py from oracle/_ref/se3_ref)."""
import ctypes
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pifusion.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pf):
    lib = ctypes.CDLL(pf.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "missing export: " + n


def test_options_keys_follow_svar_names(pf):
    o = pf.default_options()
    assert (o.band_number, o.force_float, o.high_quality_show, o.weight_type, o.scale, o.max_queue) == (5, 0, 1, 0, 1.0, 20)
    L = pf.lib()
    assert L.pf_options_set(ctypes.byref(o), b"MultiBandMap2DCPU.BandNumber", b"7") == 1 and o.band_number == 7
    assert L.pf_options_set(ctypes.byref(o), b"Map2D.Scale", b"0.5") == 1 and o.scale == 0.5
    assert L.pf_options_set(ctypes.byref(o), b"MultiBandMap2DCPU.ForceFloat", b"1") == 1 and o.force_float == 1
    assert L.pf_options_set(ctypes.byref(o), b"No.Such.Key", b"1") == 0
    # the last field of pf_options (round 6): the binding's struct and the library's agree on its place and default
    assert o.lookahead == 48 and o.fused == 1 and o.max_queue == 20
    assert L.pf_options_set(ctypes.byref(o), b"Lookahead", b"0") == 1 and o.lookahead == 0 and o.fused == 1


def test_no_device_fails_loudly(pf):
    """The product path has no CPU fallback: on a box without a GPU create() raises."""
    import pytest
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    with pytest.raises(RuntimeError):
        pf.Map2D.create(pf.TypeMultiBandCPU, False)


def test_tile_owner_spatial_hash(pf):
    o = pf.default_options(shard_count=8, shard_block=4)
    owners = {(x, y): pf.tile_owner(o, x, y) for x in range(-16, 16) for y in range(-16, 16)}
    assert set(owners.values()) == set(range(8))
    for (x, y), r in owners.items():        # constant inside a 4x4 cell, negative coords use floor division
        assert r == owners[((x // 4) * 4, (y // 4) * 4)]
    assert pf.tile_owner(pf.default_options(), 5, -3) == 0


def test_host_geometry_matches_reference_headers(pf, orc):
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "se3_vectors.json")))
    cam = vec["cam"]
    n_ok = 0
    for c in vec["cases"]:
        for impl in (pf, orc):
            assert np.array_equal(impl.se3_inverse(c["plane"]), np.array(c["plane_inv"]))
            local = impl.se3_mul(impl.se3_inverse(c["plane"]), c["world"])
            assert np.array_equal(local, np.array(c["local"]))
            assert np.array_equal(impl.so3_rotate(c["local"][3:], c["probe"]), np.array(c["rot"]))
        fp = pf.footprint(cam, c["local"])
        assert (fp is not None) == bool(c["ok"])
        if c["ok"]:
            n_ok += 1
            assert np.array_equal(fp.reshape(-1), np.array(c["pts"]))
    assert 0 < n_ok < len(vec["cases"])


def test_image_type_codes_are_the_reference_headers(pf):
    """pf_image.type is cv::Mat::type() / GImage::type(): the codes in include/pifusion.h equal the reference's own
    GSLAM::GImageType<element, channels>::Type (GImage.h:97-101), printed by oracle/_ref/se3_ref into the golden file."""
    types = json.load(open(os.path.join(ROOT, "tests", "golden", "se3_vectors.json")))["gimage_types"]
    src = open(os.path.join(ROOT, "include", "pifusion.h")).read()
    declared = {k: int(v) for k, v in re.findall(r"PF_(\d+[USF]C\d)\s*=\s*(\d+)", src)}
    assert declared == types
    assert (pf.PF_8UC3, pf.PF_8UC4, pf.PF_16SC3, pf.PF_32FC3) == (types["8UC3"], types["8UC4"], types["16SC3"], types["32FC3"])


def test_perspective_transform_same_in_product_and_oracle(pf, orc):
    rng = np.random.RandomState(0)
    for _ in range(50):
        src = np.array([0, 0, 4000, 0, 0, 3000, 4000, 3000], np.float32)
        dst = (src.reshape(4, 2) * rng.uniform(0.8, 1.2) + rng.uniform(-300, 300, (4, 2))).astype(np.float32).reshape(-1)
        Mp, Mo = pf.perspective_transform(src, dst), orc.get_perspective_transform(src, dst)
        assert np.array_equal(Mp, Mo)
        for i in range(4):      # it is a homography through the four correspondences
            v = Mp @ np.array([src[2 * i], src[2 * i + 1], 1.0])
            assert np.allclose(v[:2] / v[2], dst[2 * i:2 * i + 2], atol=1e-6)


def test_weight_map_helpers(pf):
    """MultiBandMap2DCPUEle::normalizeUsingWeightMap / mulWeightMap (MultiBandMap2DCPU.cpp:57-75):
    no caller in the reference; host loops kept for API completeness (row a17)."""
    rng = np.random.RandomState(1)
    w = rng.uniform(0, 1, 1000).astype(np.float32)
    src = rng.uniform(0, 1, (1000, 3)).astype(np.float32)
    a = src.copy()
    assert pf.lib().pf_mul_weight_map(w.ctypes.data, a.ctypes.data, 1000) == 1
    assert np.array_equal(a, src * w[:, None])
    b = src.copy()
    assert pf.lib().pf_normalize_using_weight_map(w.ctypes.data, b.ctypes.data, 1000) == 1
    d = (w.astype(np.float64) + 1e-5).astype(np.float32)                 # *weightP + 1e-5 narrowed to the float divisor
    inv = (1.0 / d.astype(np.float64)).astype(np.float32)                 # Point3_ operator/ multiplies by (1./b)
    assert np.array_equal(b, inv[:, None] * src)
    assert pf.lib().pf_mul_weight_map(None, a.ctypes.data, 10) == 0


def test_product_library_carries_the_product_kernels_only():
    """the forms of the level kernel that were measured and not adopted (rolling strips, LDS patch, 64x28 / 64x64 blocks,
    the other row loops, stamped instantiations) live in libpifusion_exp.so (-DPF_EXPERIMENTS=1); libpifusion.so carries four instantiations
    of k_levels -- {fp32 deferred stage A, int16 two rows per step} x {weight computed, weight plane gathered}."""
    import re
    here = os.path.join(ROOT, "pi-slam-fusion_amd")
    prod = open(os.path.join(here, "libpifusion.so"), "rb").read()
    names = set(m.decode() for m in re.findall(rb"_ZN2pf8k_levelsIL[0-9A-Za-z_]+?EEEvNS_10LevelBatchE", prod))
    assert len(names) == 4, names
    assert b"_ZN2pf8k_stripsI" not in prod
    exp_path = os.path.join(here, "libpifusion_exp.so")
    if os.path.exists(exp_path):
        exp = open(exp_path, "rb").read()
        assert b"_ZN2pf8k_stripsI" in exp
        assert len(set(re.findall(rb"_ZN2pf8k_levelsIL[0-9A-Za-z_]+?EEEvNS_10LevelBatchE", exp))) > 10


def test_product_library_reads_only_its_documented_environment():
    """csrc/env.hpp: the product library reads PF_CULL, PF_ROCTX, PF_DIST_VERIFY and PF_COPY_THREADS; A/B switches and timing-only
    diagnostics (some of which produce wrong tiles) exist in the experiments build alone -- their names are not even among the
    product's strings (`strings libpifusion.so | grep PF_`)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = open(os.path.join(root, "pi-slam-fusion_amd", "libpifusion.so"), "rb").read()
    names = {m.decode() for m in re.findall(rb"(?<![A-Za-z0-9_])(PF_[A-Z][A-Z0-9_]+)\x00", blob)}
    assert names == {"PF_CULL", "PF_ROCTX", "PF_DIST_VERIFY", "PF_COPY_THREADS"}, sorted(names)
    exp = os.path.join(root, "pi-slam-fusion_amd", "libpifusion_exp.so")
    if os.path.exists(exp):
        eblob = open(exp, "rb").read()
        for n in (b"PF_NO_UPPER", b"PF_TABLE_COPY", b"PF_BLEND_PER_LEVEL", b"PF_FORCE_GENERAL", b"PF_CULL_EXACT_STAT"):
            assert n + b"\x00" in eblob, n
