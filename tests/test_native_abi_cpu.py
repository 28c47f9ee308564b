"""CPU tests of the product side that need no GPU: the C-ABI library loads and exports every symbol the header
declares, its pure host functions agree with the oracle / the reference table, and the host object filter is
bit-identical to the oracle's independent restatement."""
import os
import re

import numpy as np
import pytest
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def nv():
    import __graft_entry__ as g
    g.build_native()
    from video_unscreen_amd import _native
    _native.load()
    return _native


def test_library_exports_every_declared_symbol(nv):
    header = open(os.path.join(ROOT, "include", "unscreen_hip.h")).read()
    declared = set(re.findall(r"\b(usc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = nv.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/unscreen_hip.h but not exported"
    assert declared == set(nv.EXPORTS), declared ^ set(nv.EXPORTS)
    assert lib.usc_abi_version() == 1


def test_host_geometry_matches_reference_table(nv, oracle_lib):
    table = np.load(os.path.join(GOLD, "target_size.npz"))["table"]
    for h, w, L, d, th, tw in table:
        assert nv.get_target_size(h, w, L, d) == (th, tw)
    assert nv.get_target_size(1080, 1920, 960, 32) == (544, 960)
    assert nv.pad_resize_geometry(1080, 1920, 544, 960) == (540, 960) == oracle_lib.pad_resize_geometry(1080, 1920, 544, 960)
    assert nv.inv_pad_geometry(544, 960, 1080, 1920) == (540, 960)
    assert nv.pad_resize_geometry(256, 256, 960, 960) == (960, 960)
    assert nv.pad_resize_geometry(1920, 1080, 960, 544) == (960, 540)


def test_segagent_host_geometry_matches_reference(nv, oracle_lib):
    z = np.load(os.path.join(GOLD, "deeplab_97x129.npz"))
    for h, w, L, th, tw in z["target_sizes"]:
        assert nv.segagent_target_size(h, w, L) == (th, tw)
        assert np.array_equal(nv.grid_crop_locations(th, tw), z[f"loc_{h}x{w}_{L}"])
    assert len(nv.grid_crop_locations(541, 962)) == 12 and len(nv.grid_crop_locations(962, 962)) == 18      # 12 / 18 crops
    assert len(nv.grid_crop_locations(541, 962, flip=False)) == 6
    with pytest.raises(nv.NativeError):
        nv.grid_crop_locations(500, 962)            # smaller than the crop


def test_device_entry_points_reject_host_arrays(nv):
    import torch
    with pytest.raises(nv.NativeError):
        nv.trimap(torch.zeros(8, 8, dtype=torch.uint8))          # CPU tensor: no silent fallback


def test_argument_errors_are_reported(nv):
    lib = nv.load()
    rc = lib.usc_trimap_u8(None, 0, 0, 960, 3, 5, None, None, 0, None)
    assert rc == -1 and b"usc_trimap_u8" in lib.usc_last_error()
    rc = lib.usc_morph_u8(1, 2, 8, 8, 99, 1, 0, None, 0, None)
    assert rc == -2
    # argument checks of the later entry points come before any launch, so they can be exercised without a GPU
    assert lib.usc_resize_u8(None, 4, 4, 3, None, 8, 8, 1, None) == -1 and b"usc_resize_u8" in lib.usc_last_error()
    assert lib.usc_resize_u8(1, 4, 4, 2, 2, 8, 8, 1, None) == -1                      # 2 channels
    assert lib.usc_cvt_color_u8(1, 16, 99, 2, None, None) == -1 and b"conversion code" in lib.usc_last_error()
    assert lib.usc_cvt_color_u8(1, 16, 3, 2, None, None) == -1 and b"lab_tables" in lib.usc_last_error().lower()
    assert lib.usc_gmm_alpha_u8(None, 16, None, None, None, None) == -1
    assert lib.usc_color_correct_u8(1, 2, 8, 8, 3, 960, 0.95, 4, 5, None, 6, 16, None) == -3 and b"workspace" in lib.usc_last_error()
    assert lib.usc_regionfill_u8(1, 2, 8, 8, 3, 1e-7, 0, 100, None, 4, 16, None) == -3
    assert lib.usc_regionfill_u8(1, 2, 8, 8, 3, 0.0, 0, 100, None, 4, 1 << 30, None) == -1                      # rtol must be > 0
    assert lib.usc_segagent_post(1, 65, 129, 129, 32, 2, 513, 513, 541, 962, 1080, 1920, 3, None, 4, 1 << 30, None) == -2      # > 64 crops
    assert lib.usc_upsample_nc_nhwc_f32(1, 4, 4, 6, 2, 8, 8, 8, 0, None) == -1                               # C % 4
    assert lib.usc_pack_kn_f32(1, 8, 8, 1, 8, 40, 32, 2, None) == -1                                         # Kpad % 32
    assert lib.usc_box_filter_u8(1, 8, 8, 1, 17, 2, None) == -1                                              # 16-bit column sums
    assert lib.usc_stm_post(1, 4, 4, 3, 16, 16, 16, 16, 2, 3, None) == -1                                    # odd channel pitch
    import ctypes as C
    from video_unscreen_amd.vmatting.net_hip import ConvDesc
    assert lib.usc_conv_desc_bytes() == C.sizeof(ConvDesc) == 232                                              # the ctypes mirror and the C struct move together
    blob = (C.c_uint8 * lib.usc_lab_tables_bytes())()
    lib.usc_lab_tables_host(blob)                                                                             # host function: runs here
    gamma = np.frombuffer(blob, np.uint16, 256)
    assert gamma[0] == 0 and gamma[255] == 2040 and np.all(np.diff(gamma.astype(int)) >= 0)
    lut = np.empty(256, np.float32)
    dp = C.POINTER(C.c_double)
    one = (C.c_double * 1)
    assert lib.usc_gmm_lut_host(one(100.0), one(10.0), one(1.0), 1, lut.ctypes.data) == 0
    assert abs(float(lut[100]) - 1 / (10 * np.sqrt(2 * np.pi))) < 1e-8 and abs(float(lut.sum()) - 1) < 1e-4


_HOST_FILTER = None


def remove_invalid_objects(cfg, alpha, segmask=None):
    """The border-following host stage of the library (usc_remove_invalid_objects_host) - it runs without a GPU; the NumPy util
    of the package goes through the device filter, which tests/test_gpu_objects.py checks against the same oracle."""
    global _HOST_FILTER
    from video_unscreen_amd import _native
    if _HOST_FILTER is None:
        _HOST_FILTER = _native.ObjectFilter()
    h, w = alpha.shape
    centers = cfg['objectremoval']['score_map_center']
    return _HOST_FILTER(alpha, segmask, centers['landscape'] if w > h else centers['portrait'],
                        cfg['objectremoval']['saliency_thr'], cfg['objectremoval']['consensus_thr'])


def _random_mask(H, W, seed, q, sigma):
    rng = np.random.default_rng(seed)
    x = ndi.gaussian_filter(rng.random((H, W)), sigma)
    return (x > np.quantile(x, q)).astype(np.uint8) * 255


@pytest.mark.parametrize("seed", range(12))
def test_host_object_filter_equals_oracle(nv, oracle_lib, cfg, seed):
    H, W = [(64, 96), (120, 80), (200, 300), (33, 47)][seed % 4]
    m = _random_mask(H, W, seed, [0.5, 0.7, 0.3, 0.9][seed % 4], [1, 2, 3, 5][(seed // 4) % 4])
    if seed % 3 == 0:                                             # grey alpha values (anything non-zero is object)
        m = ((m > 0) * np.random.default_rng(seed).integers(1, 256, m.shape)).astype(np.uint8)
    seg = None if seed % 2 == 0 else _random_mask(H, W, seed + 1000, 0.5, 3)
    want = oracle_lib.remove_invalid_objects(cfg, m.copy(), seg)
    got = remove_invalid_objects(cfg, m.copy(), seg)
    assert np.array_equal(got, want)


def test_host_object_filter_edge_cases(nv, oracle_lib, cfg):
    for m in (np.zeros((40, 50), np.uint8), np.full((40, 50), 255, np.uint8)):
        assert np.array_equal(remove_invalid_objects(cfg, m.copy()), oracle_lib.remove_invalid_objects(cfg, m.copy()))
    m = np.zeros((60, 60), np.uint8)
    m[0:30, 40:60] = 255            # touches top and right image edges
    m[59, 0] = 255                  # single pixel in a corner
    m[20:50, 5:35] = 255
    m[30:40, 15:25] = 0             # hole
    m[33:37, 18:22] = 255           # island inside the hole
    assert np.array_equal(remove_invalid_objects(cfg, m.copy()), oracle_lib.remove_invalid_objects(cfg, m.copy()))
    a = m.copy()
    assert remove_invalid_objects(cfg, a) is a                   # in place, like the reference
