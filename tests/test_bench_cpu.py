"""CPU tests of bench.py's host logic for the N-rank launch (no GPU, no process group): the frame ranges and the segmentation batch of the
strong-scaling form for every world size the driver uses, the NUMA binding helper, and the error hand-over of the frame writer's helper
thread."""
import os
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def bench():
    import bench as b
    return b


@pytest.mark.parametrize("frames", [300, 31, 97, 8])
def test_strong_form_ranges_and_segmentation_groups(bench, frames):
    """BASELINE.json configs[3] as the driver launches it (--clip-frames F at 1 ... 8 ranks): every rank gets a non-empty contiguous
    range, the ranges tile the clip, every rank plans the same segmentation batch, no group exceeds 12 frames (the largest batch the
    parsing net is planned for) and the batch is the even split of the largest range, not 12 + 12 + ... + a short tail."""
    from video_unscreen_amd.dist import shard_range
    for world in range(1, 9):
        if frames < world:
            continue
        sb = bench.strong_seg_batch(frames, world)
        assert 1 <= sb <= 12
        edges = [shard_range(frames, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == frames
        for (lo, hi), nxt in zip(edges, edges[1:] + [(frames, frames)]):
            assert hi > lo and hi == nxt[0]
            k = hi - lo
            groups = [min(sb, k - g) for g in range(0, k, sb)]
            assert sum(groups) == k and max(groups) <= 12 and min(groups) >= 1
        kmax = max(hi - lo for lo, hi in edges)
        ngroups = -(-kmax // 12)
        assert sb == -(-kmax // ngroups)                      # fewest groups, evenly filled
    assert bench.strong_seg_batch(300, 8) == 10 and bench.strong_seg_batch(300, 1) == 12 and bench.strong_seg_batch(38, 1) == 10


def test_bind_near_gpu_without_numa_information_leaves_affinity_alone(bench, tmp_path, monkeypatch):
    before = os.sched_getaffinity(0)
    # (a) no GPU / no sysfs entry for it: nothing to bind to
    what, sharers = bench.bind_near_gpu(0, local_world=4, sysfs=str(tmp_path))
    assert what == "no NUMA node reported" and sharers == 4 and os.sched_getaffinity(0) == before
    # (b) the device reports a node but the node's cpulist is missing (a container without /sys/devices/system/node)
    monkeypatch.setattr(bench, "gpu_numa_node", lambda index, sysfs="/sys": 1)
    what, sharers = bench.bind_near_gpu(0, local_world=2, sysfs=str(tmp_path))
    assert what.startswith("unchanged (") and sharers == 2 and os.sched_getaffinity(0) == before
    # (c) a node whose CPUs are all outside this process's cpuset
    nd = tmp_path / "devices" / "system" / "node" / "node1"
    nd.mkdir(parents=True)
    (nd / "cpulist").write_text("%d-%d\n" % (max(before) + 1000, max(before) + 1003))
    what, _ = bench.bind_near_gpu(0, local_world=2, sysfs=str(tmp_path))
    assert "outside this process's cpuset" in what and os.sched_getaffinity(0) == before


def test_bind_near_gpu_binds_every_thread_and_counts_the_sharers(bench, tmp_path, monkeypatch):
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip("needs two CPUs in the cpuset")
    keep = sorted(before)[:max(1, len(before) // 2)]
    nd = tmp_path / "devices" / "system" / "node" / "node0"
    nd.mkdir(parents=True)
    (nd / "cpulist").write_text(",".join(str(c) for c in keep) + "\n")
    monkeypatch.setattr(bench, "gpu_numa_node", lambda index, sysfs="/sys": 0 if index in (0, 2, 3) else 1)    # 4 local GPUs, three on node 0
    stop = threading.Event()
    seen = {}

    def other():                                  # a thread that exists BEFORE the bind (torch's intra-op pool in the real run)
        stop.wait()
        seen["aff"] = os.sched_getaffinity(0)
    th = threading.Thread(target=other)
    th.start()
    try:
        what, sharers = bench.bind_near_gpu(0, local_world=4, sysfs=str(tmp_path))
        assert what == "node 0: %d cpus" % len(keep) and sharers == 3
        assert os.sched_getaffinity(0) == set(keep)
    finally:
        stop.set()
        th.join()
        for tid in [int(t) for t in os.listdir("/proc/self/task")]:
            os.sched_setaffinity(tid, before)
    assert seen["aff"] == set(keep)                # the pre-existing thread was re-bound too


def test_deferred_submit_hands_its_error_to_close():
    from video_unscreen_amd.utils.frameio import DeferredSubmit

    class Ev:
        def synchronize(self):
            pass

    class Writer:
        def __init__(self):
            self.got = []

        def submit(self, slot, paths):
            if slot == 1:
                raise OSError("disk full")
            self.got.append(slot)
    w = Writer()
    d = DeferredSubmit(w)
    for slot in range(4):
        d.submit_after(Ev(), slot, ("a", "f"))
    with pytest.raises(RuntimeError, match="handing a finished frame to the writer failed") as ei:
        d.close()
    assert isinstance(ei.value.__cause__, OSError) and w.got == [0]          # nothing is written after the failure, close() does not hang
    ok = DeferredSubmit(Writer())
    ok.submit_after(Ev(), 0, ("a", "f"))
    ok.close()
