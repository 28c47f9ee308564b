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


def _npy_groups(tmp_path, n, H=8, W=12):
    groups = []
    for i in range(n):
        p = str(tmp_path / f"{i:06d}.npy")
        np.save(p, np.full((H, W, 3), i, np.uint8))
        groups.append((p,))
    return groups


def test_reader_out_of_slots_waits_for_a_release_instead_of_hanging(tmp_path):
    """A consumer that runs ahead (every slot delivered, none released, no decode in flight) used to block for ever in done.get():
    with a wait_release callback the reader asks for a slot back; without one it raises."""
    import pytest
    groups = _npy_groups(tmp_path, 7)
    rd = FrameReader(groups, [(8, 12, 3)], workers=2, depth=2, pin=False)
    next(rd), next(rd)
    with pytest.raises(RuntimeError, match="none was released"):
        next(rd)
    rd.close()
    held, calls = [], []
    rd = FrameReader(groups, [(8, 12, 3)], workers=2, depth=2, pin=False)

    def wait_release():
        calls.append(len(held))
        rd.release(held.pop(0))                               # e.g. after synchronising on the oldest upload

    rd.wait_release = wait_release
    seen = []
    for slot, (t,) in rd:
        seen.append(int(t[0, 0, 0]))
        held.append(slot)
    rd.close()
    assert seen == list(range(7)) and len(calls) == 5 and all(c == 2 for c in calls)


def test_reader_reports_a_dead_decoder(tmp_path):
    import pytest
    groups = _npy_groups(tmp_path, 4) + [(str(tmp_path / "missing.npy"),)]
    rd = FrameReader(groups[:4], [(8, 12, 3)], workers=1, depth=1, pin=False)
    slot, _ = next(rd)
    rd.procs[0].terminate()
    rd.procs[0].join()
    rd.release(slot)                                          # a task goes to a worker that no longer exists
    with pytest.raises(RuntimeError, match="died"):
        next(rd)


def test_packed_host_layouts():
    """hostio's packed layouts: a host group / output pair laid out as the device staging slot / output-ring entry is recognised as ONE
    contiguous span (one copy per frame and direction); anything else falls back to per-array copies.  Odd sizes keep 16-byte starts."""
    import torch
    from video_unscreen_amd.hostio import _host_span, in_layout, out_layout, pack_host_group, packed_outputs
    for H, W in ((33, 47), (270, 480)):
        n = H * W
        f = torch.randint(0, 255, (H, W, 3), dtype=torch.uint8)
        b = torch.randint(0, 255, (H, W, 3), dtype=torch.uint8)
        m = torch.randint(0, 255, (H, W), dtype=torch.uint8)
        bg_off, seg_off, total = in_layout(H, W)
        assert bg_off % 16 == 0 and seg_off % 16 == 0 and total % 16 == 0 and bg_off >= 3 * n and seg_off >= bg_off + 3 * n
        F, S, B = pack_host_group(f, b, m, pin=False)
        assert torch.equal(F, f) and torch.equal(B, b) and torch.equal(S, m)
        span = _host_span([F, B, S], [0, bg_off, seg_off])
        assert span is not None and span.numel() == seg_off + n and torch.equal(span[bg_off:bg_off + 3 * n].view(H, W, 3), b)
        F2, S2, B2 = pack_host_group(f, b, None, pin=False)
        assert S2 is None and _host_span([F2, B2], [0, bg_off, seg_off]).numel() == bg_off + 3 * n
        assert _host_span([f, b], [0, bg_off, seg_off]) is None                      # separate allocations: two copies
        assert _host_span([F, S, B], [0, bg_off, seg_off]) is None                   # wrong order
        a, g = packed_outputs(3, H, W, pin=False)
        fg_off, tot = out_layout(H, W)
        assert a.shape == (3, H, W) and g.shape == (3, H, W, 3) and a[1].is_contiguous() and g[1].is_contiguous() and fg_off % 16 == 0
        assert _host_span([a[1], g[1]], [0, fg_off]).numel() == fg_off + 3 * n
        assert _host_span([a[1], g[2]], [0, fg_off]) is None
        g[1].fill_(7)
        a[1].fill_(9)
        assert int(_host_span([a[1], g[1]], [0, fg_off])[:n].sum()) == 9 * n and not bool((a[0] == 9).all() and (g[2] == 7).all())


def test_ring_slots_share_the_staging_layout(tmp_path):
    """A reader slot (frame, bg, mask) and a writer slot (alpha, fg) are laid out like hostio.in_layout / out_layout: whole slots move
    with one copy."""
    from video_unscreen_amd._io_workers import part_offsets
    from video_unscreen_amd.hostio import in_layout, out_layout
    for H, W in ((33, 47), (1080, 1920)):
        offs, total = part_offsets([(H, W, 3), (H, W, 3), (H, W)])
        assert (offs[1], offs[2], total) == in_layout(H, W)
        offs, total = part_offsets([(H, W), (H, W, 3)])
        assert (offs[1], total) == out_layout(H, W)


class _Ev:
    def __init__(self, fail=False, gate=None):
        self.fail, self.gate = fail, gate

    def synchronize(self):
        if self.gate is not None:
            self.gate.wait()
        if self.fail:
            raise RuntimeError("hipErrorLaunchFailure")


def test_failed_hand_over_raises_in_the_loop_instead_of_hanging(tmp_path):
    """after a failure in DeferredSubmit's helper thread the later slots were neither written nor returned, the loop
    then blocked for ever in FrameWriter.acquire() (close(), the only place the error surfaced, was never reached).  Now the slots come
    back, and the next submit_after() / acquire() on the loop's thread raises the helper's error."""
    import threading
    import pytest
    from video_unscreen_amd.utils.frameio import DeferredSubmit
    H, W, depth = 6, 8, 3
    wr = FrameWriter([(H, W)], workers=1, depth=depth, pin=False)
    d = DeferredSubmit(wr)
    gate = threading.Event()
    # the loop runs ahead of the "GPU": every slot acquired, all of them waiting in the helper thread, nothing with the encoders yet
    for i in range(depth):
        slot, (a,) = wr.acquire()
        a.zero_()
        d.submit_after(_Ev(fail=(i == 0), gate=gate), slot, (str(tmp_path / f"a_{i}.npy"),))
    threading.Timer(0.3, gate.set).start()
    with pytest.raises(RuntimeError, match="handing a finished frame to the writer failed") as ei:
        wr.acquire()                                   # used to block for ever
    assert "hipErrorLaunchFailure" in str(ei.value.__cause__)
    with pytest.raises(RuntimeError, match="handing a finished frame"):
        d.submit_after(_Ev(), 0, ("x",))
    with pytest.raises(RuntimeError, match="handing a finished frame"):
        d.close()
    assert not list(tmp_path.iterdir())                # nothing is written after the failure
    wr.close()                                         # every slot came back: close() does not wait for anything
    assert sorted(wr.free) == list(range(depth))


def test_writer_acquire_is_bounded(tmp_path):
    """Slots that were acquired and never handed on, or encoders that died, make acquire() raise instead of waiting for ever."""
    import pytest
    wr = FrameWriter([(4, 4)], workers=1, depth=2, pin=False)
    wr.acquire(), wr.acquire()
    with pytest.raises(RuntimeError, match="none is being written"):
        wr.acquire()
    wr.give_back(0)
    slot, _ = wr.acquire()
    assert slot == 0
    wr.submit(slot, (str(tmp_path / "nonexistent_dir" / "a.npy"),))       # the encoder reports the failure ...
    with pytest.raises(RuntimeError, match="FrameWriter"):
        for _ in range(3):
            wr.acquire()                                                   # ... and the next acquire raises it
    wr.errors.clear()
    slot, _ = wr.acquire()                                                 # (the failed slot came back; slot 1 is still held)
    wr.procs[0].terminate()
    wr.procs[0].join()
    wr.submit(slot, (str(tmp_path / "b.npy"),))
    with pytest.raises(RuntimeError, match="died"):
        wr.acquire()
    wr.inflight = 0
    wr.errors.clear()
    wr.close()


def test_async_saver_save_raises_after_a_failed_copy():
    """AsyncSaver.save() checks the helper thread's error before it asks for a slot (CPU stand-in: the class is built by hand around a
    CPU writer; the real save() copies on the current HIP stream)."""
    import pytest
    from video_unscreen_amd.utils.frameio import AsyncSaver, DeferredSubmit
    sv = AsyncSaver.__new__(AsyncSaver)
    sv.writer = FrameWriter([(4, 4)], workers=1, depth=2, pin=False)
    sv.deferred = DeferredSubmit(sv.writer)
    slot, _ = sv.writer.acquire()
    sv.deferred.submit_after(_Ev(fail=True), slot, ("a",))
    sv.deferred.th.join(timeout=0.5)
    import time
    t0 = time.time()
    while sv.deferred.error is None and time.time() - t0 < 5:
        time.sleep(0.01)
    with pytest.raises(RuntimeError, match="handing a finished frame"):
        sv.save([None], [None])
    with pytest.raises(RuntimeError):
        sv.close()
