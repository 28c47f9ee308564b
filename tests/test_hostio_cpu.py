"""Host-side logic of the frame loop's boundary that needs no GPU: the stand-in for a copy event that a helper thread records later
(hostio.PendingCopy), and the CPU-share cap of torch's thread pool (video_unscreen_amd.usable_cpus)."""
import os
import threading
import time

import pytest
import torch


class _FakeEvent:
    def __init__(self):
        self.done = False
        self.waited_by = []

    def query(self):
        return self.done

    def synchronize(self):
        self.done = True


class _FakeStream:
    def __init__(self):
        self.waits = []

    def wait_event(self, ev):
        self.waits.append(ev)


def test_pending_copy_stands_in_for_an_event_recorded_later():
    from video_unscreen_amd.hostio import PendingCopy
    p = PendingCopy()
    assert p.query() is False                       # not enqueued yet: not finished
    ev = _FakeEvent()
    st = _FakeStream()
    t = threading.Timer(0.05, lambda: p._set(ev))
    t0 = time.perf_counter()
    t.start()
    p.wait_on(st)                                   # blocks until the helper thread has enqueued the copy, then orders the stream
    assert time.perf_counter() - t0 >= 0.04 and st.waits == [ev]
    assert PendingCopy.waited > 0
    assert p.query() is False
    p.synchronize()
    assert p.query() is True


def test_pending_copy_hands_the_helper_threads_error_to_whoever_waits():
    from video_unscreen_amd.hostio import PendingCopy
    p = PendingCopy()
    p._set(error=ValueError("copy failed"))
    for call in (p.synchronize, p.query, lambda: p.wait_on(_FakeStream())):
        with pytest.raises(RuntimeError, match="download thread") as ei:
            call()
        assert isinstance(ei.value.__cause__, ValueError)


def test_torch_thread_pool_respects_the_cpu_share():
    import subprocess
    import sys
    import video_unscreen_amd as pkg
    n = pkg.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    pkg.respect_cpu_share()                          # (what _native.load() does before anything is built)
    assert torch.get_num_threads() <= max(n, 1) or os.environ.get("USC_KEEP_TORCH_THREADS")
    # importing the package must not pull torch in: the frame I/O worker processes import it too
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import sys; import video_unscreen_amd, video_unscreen_amd._io_workers; print('torch' in sys.modules)"],
                       capture_output=True, text=True, cwd=root)
    assert r.returncode == 0 and r.stdout.strip() == "False", r.stdout + r.stderr


def test_layouts_keep_every_part_16_byte_aligned():
    from video_unscreen_amd.hostio import in_layout, out_layout
    for H, W in ((1080, 1920), (135, 241), (1, 3), (2160, 3840), (33, 47)):
        bg, seg, total = in_layout(H, W)
        fg, tot = out_layout(H, W)
        assert bg % 16 == 0 and seg % 16 == 0 and total % 16 == 0 and fg % 16 == 0 and tot % 16 == 0
        assert bg >= 3 * H * W and seg >= bg + 3 * H * W and total >= seg + H * W and fg >= H * W and tot >= fg + 3 * H * W
